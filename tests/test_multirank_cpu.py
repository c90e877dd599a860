"""CPU, gloo, world_size=2: the batch-sharding logic used on N GPUs.

The sharded run (scatter -> per-rank forward -> gather) must equal the single-process run bit for
bit because no op mixes samples.  The per-rank forward here is the CPU oracle (checker only): the
property under test is the sharding/collective plumbing, which is what runs over RCCL on GPUs."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
    import fdn_oracle as O
    from common import fixture, fixture_weights
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    fx = fixture("mar_full")
    P = {"net_a." + k: v for k, v in fixture_weights("mar_full", fx["shapes"]).items()}
    full_x = torch.cat([fx["x"], fx["x"].flip(0)])            # global batch 4 -> 2 per rank
    full_r = torch.cat([fx["ratio"], fx["ratio"].flip(0)])
    per = full_x.shape[0] // world
    x, r = torch.empty(per, *full_x.shape[1:]), torch.empty(per, 1)
    dist.scatter(x, list(full_x.chunk(world)) if rank == 0 else None, src=0)
    dist.scatter(r, list(full_r.chunk(world)) if rank == 0 else None, src=0)
    with torch.no_grad():
        y = O.mar(x, r.view(-1, 1, 1, 1), P, "net_a")[2].contiguous()
    outs = [torch.empty_like(y) for _ in range(world)] if rank == 0 else None
    dist.gather(y, outs, dst=0)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                   # the bench's max-over-ranks timing reduction
    if rank == 0:
        with torch.no_grad():
            ref = O.mar(full_x, full_r.view(-1, 1, 1, 1), P, "net_a")[2]
        q.put((torch.equal(torch.cat(outs), ref), float(t.item())))
    dist.destroy_process_group()


def test_sharded_equals_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    same, tmax = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
    assert same and tmax == 2.0
