"""CPU, gloo, world_size=2: the batch-sharding code that runs over RCCL on N GPUs (fdn_hip/sharding.py
is the module bench.py drives), and bench.py's own launcher (`--gpus 2` without torchrun) in its CPU rehearsal mode.

The sharded run (scatter -> per-rank forward -> gather) must equal the single-process run bit for bit because no op
mixes samples.  The per-rank forward here is the CPU oracle (checker only): the property under test is the
sharding / collective plumbing."""
import json
import os
import subprocess
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
    import fdn_oracle as O
    from common import fixture, fixture_weights
    from fdn_hip import sharding                                   # host-side module: importable without a GPU
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    fx = fixture("mar_full")
    P = {"net_a." + k: v for k, v in fixture_weights("mar_full", fx["shapes"]).items()}
    full_x = torch.cat([fx["x"], fx["x"].flip(0)])            # global batch 4 -> 2 per rank
    full_r = torch.cat([fx["ratio"], fx["ratio"].flip(0)])
    per = full_x.shape[0] // world
    r = sharding.scatter_batch(dist, torch.empty(per, 1), sharding.split_batch(full_r, world) if rank == 0 else None)

    def forward(x):
        with torch.no_grad():
            return O.mar(x, r.view(-1, 1, 1, 1), P, "net_a")[2].contiguous()

    outs = [torch.empty(per, *fx["x"].shape[1:]) for _ in range(world)] if rank == 0 else None
    sharding.sharded_step(dist, forward, torch.empty(per, *full_x.shape[1:]),
                          sharding.split_batch(full_x, world) if rank == 0 else None, outs)
    tmax = sharding.max_over_ranks(dist, float(rank + 1), torch.device("cpu"))   # the bench's max-over-ranks reduction
    if rank == 0:
        with torch.no_grad():
            ref = O.mar(full_x, full_r.view(-1, 1, 1, 1), P, "net_a")[2]
        q.put((torch.equal(torch.cat(outs), ref), tmax))
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


def test_shard_bounds():
    sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
    from fdn_hip import sharding
    assert sharding.shard_bounds(64, 8) == list(range(0, 65, 8))
    assert sharding.shard_bounds(10, 4) == [0, 3, 6, 8, 10]
    x = torch.arange(12).view(6, 2)
    assert torch.equal(torch.cat(sharding.split_batch(x, 4)), x)


def test_bench_launcher_spawns_gpus_ranks():
    """`python bench.py --gpus 2` with no launcher around it must start 2 ranks itself (it used to run one rank
    and print n_gpus 1), push the scatter / gather through the timed loop and report the collective world size."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run",
                        "--batch", "3"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["rccl_ranks"] == 2 and line["config"]["global_batch"] == 6
    assert line["config"]["scatter_gather_timed"] is True and "without_collectives" in line and line["dry_run"] is True


def test_bench_rejects_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0 and "must agree" in (r.stderr + r.stdout)


def test_bench_launcher_world_8_with_affinity():
    """The configs[3] launch shape on the CPU: 8 ranks, each pinned to its own cut of the host cores before any GPU call, the
    scatter / gather of a global batch of 8 x B through the timed loop (gloo)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--dry-run",
                        "--batch", "2"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["config"]["rccl_ranks"] == 8 and line["config"]["global_batch"] == 16
    aff = line["config"]["cpu_affinity_rank0"]
    assert aff is not None and aff["cores"] >= 1 and aff["cores"] <= max(1, (os.cpu_count() or 8) // 8 + 1)
    assert line["config"]["host_issue_ms_per_step"] >= 0.0


def test_rank_cpus_partition_the_allowed_cores():
    sys.path.insert(0, ROOT)
    import bench
    allowed = sorted(os.sched_getaffinity(0))
    cuts = [bench.rank_cpus(r, 4)[0] for r in range(4)]
    assert all(cuts) and all(set(c) <= set(allowed) for c in cuts)
    if len(allowed) >= 4:                                       # disjoint when there are enough cores
        assert len(set().union(*map(set, cuts))) == sum(len(c) for c in cuts)


def test_bench_launcher_ends_promptly_when_a_rank_dies():
    """a rank other than the first that crashes must end the job at once (the survivors would otherwise sit in a
    collective until the watchdog fires)."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env["FDN_BENCH_DRY_FAIL_RANK"] = "2"
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "1", "--warmup", "0", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 3 and time.time() - t0 < 120


class _FakeDist:
    """Records how sharding drives the collectives (no process group needed)."""

    def __init__(self):
        self.calls = []

    def get_rank(self):
        return 0

    def scatter(self, out, lst, src=0, async_op=None):
        self.calls.append(("scatter", async_op))
        out.copy_(lst[0])

    def gather(self, t, lst, dst=0, async_op=None):
        self.calls.append(("gather", async_op))
        lst[0].copy_(t)


def test_sharded_step_is_strictly_serial():
    """scatter -> forward -> gather never overlap.  Both collectives are issued blocking (async_op=False
    explicitly, so a later "optimisation" to async prefetch has to delete this test), in that order, around the forward."""
    sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
    from fdn_hip import sharding
    d = _FakeDist()
    order = []
    x = torch.arange(6.0).view(2, 3)
    bufs = [torch.empty(2, 3)]

    def fwd(t):
        order.append(list(d.calls))
        return t * 2

    sharding.sharded_step(d, fwd, torch.empty(2, 3), [x], bufs)
    assert d.calls == [("scatter", False), ("gather", False)]
    assert order == [[("scatter", False)]]                      # the forward ran after the scatter and before the gather
    assert torch.equal(bufs[0], x * 2)


def test_second_stream_is_refused_without_opt_in(monkeypatch):
    """the multi-stream forward is known to corrupt rows on MI355X / ROCm 7.2: it raises
    unless FDN_HIP_ALLOW_MULTISTREAM=1, in forward_streams and in GraphedStep, before anything touches a GPU."""
    sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
    import pytest
    from fdn_hip import pipeline
    monkeypatch.delenv(pipeline.MULTISTREAM_ENV, raising=False)
    with pytest.raises(RuntimeError, match="n_streams=3 refused"):
        pipeline.forward_streams(None, None, torch.zeros(6, 3, 8, 8), n_streams=3)
    with pytest.raises(RuntimeError, match="refused"):
        pipeline.GraphedStep(None, None, n_streams=2)
    monkeypatch.setenv(pipeline.MULTISTREAM_ENV, "1")
    pipeline.GraphedStep(None, None, n_streams=2)               # opt-in: constructs (nothing runs here)


def test_graphed_forward_does_not_travel_with_copies():
    """copy.deepcopy / pickle of a model that has run through pipeline.run must not try to copy HIP graphs."""
    import copy
    import pickle
    sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
    from fdn_hip import pipeline
    net, lp = torch.nn.Linear(2, 2), torch.nn.Linear(2, 2)
    g = pipeline.GraphedForward(net, lp)
    g._graphs["k"] = (object(), None, None, None)               # stands for a captured graph
    net.__dict__["_fdn_graphed"] = g
    c = copy.deepcopy(net)
    assert c.__dict__.get("_fdn_graphed") is None
    st = pickle.loads(pickle.dumps(g))
    assert st.net is None and len(st._graphs) == 0


def _uneven_worker(rank, world, port, total, q, root=0):
    sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
    from fdn_hip import sharding
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(5)
    x_all = torch.rand(total, 3, 8, 12, generator=g)
    seen = []

    def forward(x):                                          # per-sample independent stand-in with a different output shape
        seen.append(x.shape[0])
        y_ = (x * 2.0 + x.flip(-1))[:, :, ::2].contiguous()
        return y_.double() if root else y_                   # (root != 0: the forward also changes the dtype)
    y = sharding.sharded_run(dist, forward, total, torch.empty(1, 3, 8, 12), x_all if rank == root else None, root=root)
    b = sharding.shard_bounds(total, world)
    ok_count = seen == ([b[rank + 1] - b[rank]] if b[rank + 1] > b[rank] else [])
    flags = [None] * world
    dist.all_gather_object(flags, ok_count)
    if rank == root:
        q.put((torch.equal(y, forward(x_all)), all(flags), tuple(y.shape)))
    dist.destroy_process_group()


def test_uneven_batch_with_a_root_that_owns_no_item():
    """root = 1 with a global batch of ONE item - the root holds the batch but owns no item (shard_bounds gives the extra items to the
    lowest ranks), and the forward changes the dtype: rank 0 (which always owns item 0) broadcasts the output's trailing shape and dtype, the root
    contributes an empty block of that dtype, nobody hangs."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33900 + os.getpid() % 2000
    procs = [ctx.Process(target=_uneven_worker, args=(r, 2, port, 1, q, 1)) for r in range(2)]
    for p in procs:
        p.start()
    same, counts_ok, shape = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
    assert same and counts_ok and shape == (1, 3, 4, 12), (same, counts_ok, shape)


def test_global_batch_not_a_multiple_of_the_world_size():
    """scatter_uneven / gather_uneven / sharded_run - a global batch of 3 (2 + 1) and of 1 (1 + 0: a rank without an item) over
    two gloo ranks equals the single-process result bit for bit, every rank's forward sees exactly its own item count (no compute on padding)."""
    ctx = mp.get_context("spawn")
    for k, total in enumerate((3, 1, 5)):
        q = ctx.Queue()
        port = 31700 + (os.getpid() + 7 * k) % 2000
        procs = [ctx.Process(target=_uneven_worker, args=(r, 2, port, total, q)) for r in range(2)]
        for p in procs:
            p.start()
        same, counts_ok, shape = q.get(timeout=120)
        for p in procs:
            p.join(timeout=60)
        assert same and counts_ok and shape == (total, 3, 4, 12), (total, same, counts_ok, shape)
