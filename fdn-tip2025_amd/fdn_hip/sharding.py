"""Batch sharding of the LPNet -> FDN path over the GPUs of one node (host side, no compute of its own).

Images never interact inside the path (no BatchNorm in FDN, LPNet's BatchNorm runs on its running statistics;
SURVEY.md 8e), so N GPUs = N contiguous shards of the batch and replicated weights, the way the reference's own
validation loop deals images to ranks (basicsr/models/image_restoration_model.py:731, `idx % world_size == rank`).
The only collectives are the scatter of the inputs from the root and the gather of the outputs to it
(torch.distributed: backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests - same code).
"""
import torch


def shard_bounds(total, world):
    """Contiguous, balanced shards: rank r owns items [bounds[r], bounds[r+1])."""
    base, extra = divmod(total, world)
    b = [0]
    for r in range(world):
        b.append(b[-1] + base + (1 if r < extra else 0))
    return b


def split_batch(x_all, world):
    """The root's view of a global batch as the list of per-rank shards (no copies)."""
    b = shard_bounds(x_all.shape[0], world)
    return [x_all[b[r]:b[r + 1]] for r in range(world)]


def scatter_batch(dist, shard_like, root_shards, src=0):
    """Every rank receives its shard of the root's batch.  root_shards: list of `world` tensors on the root, else None.
    Shards must have equal shapes (pad the global batch to a multiple of the world size)."""
    out = torch.empty_like(shard_like)
    dist.scatter(out, root_shards if dist.get_rank() == src else None, src=src, async_op=False)      # blocking: see sharded_step
    return out


def gather_batch(dist, out, root_bufs, dst=0):
    """The root receives every rank's output shard into root_bufs (list of `world` tensors on the root, else None)."""
    dist.gather(out.contiguous(), root_bufs if dist.get_rank() == dst else None, dst=dst, async_op=False)


_step_done = {}          # device index -> HIP event recorded behind the last gather of a step


def sharded_step(dist, forward, shard_like, root_shards=None, root_bufs=None, root=0):
    """scatter -> forward on the local shard -> gather: one step of the N-GPU path.  Returns the local output.

    The three stages are strictly SERIAL on the GPU, and that is a rule, not an accident: RCCL's copy kernels
    run on RCCL's own stream, and on MI355X / ROCm 7.2 any kernel that shares the GPU with a bf16-MFMA kernel of another stream can
    return wrong rows.  Both collectives are therefore blocking (`async_op=False`: the compute stream waits for the collective,
    the collective waits for the compute stream), the end of a step is marked with an event that the next step's scatter waits
    on, and prefetching the next shard during compute - worth < 1 % at 88.5 MB per peer - is deliberately not done."""
    cuda = shard_like.is_cuda
    if cuda:
        cur = torch.cuda.current_stream(shard_like.device)
        prev = _step_done.get(shard_like.device.index)
        if prev is not None:
            cur.wait_event(prev)                       # nothing of this step starts before the previous gather has finished
    xin = scatter_batch(dist, shard_like, root_shards, src=root)
    out = forward(xin)
    if cuda and torch.cuda.current_stream(shard_like.device) != cur:
        raise RuntimeError("sharded_step: the forward changed the current HIP stream; one stream per GPU is the rule (DESIGN.md 4.7)")
    gather_batch(dist, out, root_bufs, dst=root)
    if cuda:
        ev = _step_done.get(shard_like.device.index)
        if ev is None:
            ev = _step_done[shard_like.device.index] = torch.cuda.Event()
        ev.record(cur)
    return out


def max_over_ranks(dist, seconds, device):
    """The bench's timing reduction: the slowest rank defines the step time."""
    t = torch.tensor([seconds], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
