"""Batch-level execution helpers for the FDN path (host side, no compute of their own).

Images never interact inside LPNet -> FDN, so a batch can be cut into independent
sub-batches.  Multi-GPU sharding (one process per GPU, bench.py) is that cut; ONE stream per GPU is the
rule inside a process.

`forward_streams(..., n_streams > 1)` - sub-batches on separate HIP streams of one GPU, +1.3 % images/s at
B = 8 since the 1x1 convs moved to the bf16 matrix pipe - is kept for experiments only and is NOT used by
`run`, bench.py or the tests' reference paths: on MI355X / ROCm 7.2 a kernel issuing
v_mfma_f32_32x32x16_bf16 corrupts kernels of OTHER streams that share the GPU with it (wrong rows in up
to half of the neighbour's launches; rocFFT is hit as well as this library's kernels, and a loop of
nothing but compiler-generated MFMAs is enough to trigger it: tools/cross_stream_probe.py,
profiles/r03_cross_stream_probe.txt).  Within one stream kernels never overlap and the
outputs are bit-stable (tests/test_gpu_edge.py::test_single_stream_bit_stable).
"""
import collections
import contextlib
import gc
import os

import torch

from . import storage_dtype

_streams = {}
# Graph captures check only THIS thread's GPU calls: with a process group alive (one rank per GPU) RCCL's watchdog thread polls its events
# from another thread, which the default ("global") capture mode would take for an illegal call during capture and abort the capture
CAPTURE_MODE = "thread_local"
MULTISTREAM_ENV = "FDN_HIP_ALLOW_MULTISTREAM"      # "1": allow n_streams > 1 (experiments; results are NOT bit-stable, see above)


@contextlib.contextmanager
def _capturing(graph):
    """torch.cuda.graph(...) with Python's cyclic garbage collector held off for the duration of the capture.  A forward allocates thousands of Python
    objects; a generational collection that starts INSIDE a capture can run the destructor of an older CUDAGraph / graph-private pool that has just
    become unreachable (hipGraphExecDestroy, pool trimming) while this thread's stream is capturing.  (Round 6: two GPU-suite runs on fresh boxes aborted
    inside the collector around the graph tests, "Fatal Python error: Aborted ... Garbage-collecting", never reproduced afterwards; torch itself collects
    once on entry - this keeps the rest of the capture quiet too.)"""
    was = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph, capture_error_mode=CAPTURE_MODE):
            yield
    finally:
        if was:
            gc.enable()


def _check_streams(n_streams):
    """One HIP stream per GPU is the product rule.  More than one is refused unless the caller opts in
    through the environment: the multi-stream forward is known to return wrong rows on MI355X / ROCm 7.2."""
    if n_streams > 1 and os.environ.get(MULTISTREAM_ENV) != "1":
        raise RuntimeError(
            f"fdn_hip: n_streams={n_streams} refused - kernels of different HIP streams that share the GPU with a bf16-MFMA "
            f"kernel return wrong rows on MI355X / ROCm 7.2 (DESIGN.md 4.7).  Use one stream per GPU, or set {MULTISTREAM_ENV}=1 "
            "for experiments (to bisect, call fdn_hip.set_matrix_pipe('f32') first: the fp32-MFMA forms of every matrix product).")


def _get_streams(device, n):
    key = (device.index if device.index is not None else torch.cuda.current_device(), n)
    if key not in _streams:
        _streams[key] = [torch.cuda.Stream(device=device) for _ in range(n)]
    return _streams[key]


def forward_streams(net, lpnet, x, n_streams=1, keep=None):
    """result = FDN(x, ratio_i=LPNet(x))[0]; n_streams > 1 splits the batch over HIP streams (see the module note: not
    bit-stable on MI355X / ROCm 7.2, experiments only).  keep: a dict that receives the LPNet output as keep["ratio"] (one stream only)."""
    _check_streams(n_streams)
    B = x.shape[0]
    if n_streams <= 1 or B < n_streams:
        with torch.no_grad():
            ratio = lpnet(x)
            if keep is not None:
                keep["ratio"] = ratio
            return net(x, ratio_i=ratio, device=x.device)[0]
    cur = torch.cuda.current_stream(x.device)
    streams = _get_streams(x.device, n_streams)
    bounds = [round(i * B / n_streams) for i in range(n_streams + 1)]
    outs = []
    for i, s in enumerate(streams):
        s.wait_stream(cur)                          # inputs were produced on the caller's stream
        with torch.cuda.stream(s), torch.no_grad():
            xi = x[bounds[i]:bounds[i + 1]].contiguous()
            xi.record_stream(s)
            outs.append(net(xi, ratio_i=lpnet(xi), device=x.device)[0])
    for s, o in zip(streams, outs):
        cur.wait_stream(s)                          # the caller's stream may now consume the outputs
        o.record_stream(cur)
    return torch.cat(outs)


def weights_signature(*modules):
    """What a captured graph depends on besides the input shape: the storage mode, the routing switches that decide WHICH kernels a
    forward launches (the matrix-pipe mode, the optional one-launch FDSA route) and every parameter / buffer of the live module trees
    (count, in-place version counters, addresses).  A captured graph replays the kernels of its capture: after set_matrix_pipe() or a
    change of ops.FDSA_FULL the key differs and the holder captures again."""
    from . import matrix_pipe_mode, ops
    ps = [p for m in modules for p in list(m.parameters()) + list(m.buffers())]
    return (storage_dtype(), matrix_pipe_mode(), bool(ops.FDSA_FULL), int(ops.FDSA_FULL_MAX_C), bool(ops.FDSA_TAIL), bool(ops.FDSA_TAIL_PIN), int(ops.FDSA_TAIL_PIN_MAX_C), str(ops.FFN_TAIL_MODE), bool(ops.SPECTRAL_MLP_FUSED), bool(ops.GEMM_OWN_STATS), bool(ops.UPCONV_GATHER), bool(ops.AFF_MULTIRES),
            len(ps), sum(p._version for p in ps), sum(p.data_ptr() & 0xFFFFFFFF for p in ps))


class GraphedForward:
    """LPNet -> FDN captured once per input shape into a HIP graph and replayed (hipGraphLaunch).

    A forward is ~800 kernel launches; below ~1 MPixel per call the Python/launch overhead dominates
    (256x256: 29 ms eager).  Every entry point of libfdn_hip.so is capture-safe (no allocation, no sync;
    FFT twiddle tables are built by the warm-up call), activations come from a graph-private pool.
    Usage:  g = GraphedForward(net, lpnet); out = g(x)   # out is overwritten by the next call
    """

    MAX_GRAPHS = 4          # captured graphs kept per instance (least recently used goes first: each pins a private memory pool)

    def __init__(self, net, lpnet, warmup=2):
        self.net, self.lpnet, self.warmup = net, lpnet, warmup
        self._graphs = collections.OrderedDict()
        self._seen = collections.OrderedDict()      # shapes met once, not captured yet (a capture costs two forwards + a pool)

    def __deepcopy__(self, memo):
        """A copy of the model (EMA copy, copy.deepcopy(net)) must not try to copy HIP graphs: the copy starts with none and
        shares nothing with this instance (run() builds a fresh GraphedForward for a copied net anyway)."""
        return None

    def __getstate__(self):
        """pickle / torch.save(net): graphs, static tensors and the back-references stay behind."""
        return {"warmup": self.warmup}

    def __setstate__(self, st):
        self.net = self.lpnet = None
        self.warmup = st.get("warmup", 2)
        self._graphs = collections.OrderedDict()
        self._seen = collections.OrderedDict()

    def _weights_signature(self):
        """A captured graph holds raw pointers to the weights and to the derived operands built from them (LayerNorm folds,
        packed MFMA operands), and was recorded in ONE storage mode: any in-place update, re-load or replacement of a parameter
        or buffer (the live module tree is re-read on every call) and any change of fdn_hip.storage_dtype() invalidates it."""
        return weights_signature(self.net, self.lpnet)

    def _eager(self, x):
        with torch.no_grad():
            return self.net(x, ratio_i=self.lpnet(x), device=x.device)[0]

    def _capture(self, x):
        static_x = x.clone()
        s = torch.cuda.Stream(device=x.device)
        s.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(s), torch.no_grad():
            for _ in range(self.warmup):            # builds FFT tables / weight caches outside the capture
                self.net(static_x, ratio_i=self.lpnet(static_x), device=x.device)
        torch.cuda.current_stream(x.device).wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with _capturing(g), torch.no_grad():
            static_out = self.net(static_x, ratio_i=self.lpnet(static_x), device=x.device)[0]
        return g, static_x, static_out, self._weights_signature()

    def __call__(self, x):
        key = (tuple(x.shape), x.device.index)
        hit = self._graphs.get(key)
        if hit is not None and hit[3] != self._weights_signature():  # the weights or the storage mode changed: capture again
            del self._graphs[key]
            hit = None
        if hit is None:
            if key not in self._seen:                                # a shape met for the first time (trailing partial batch, odd frame
                self._seen[key] = True                               # size): run it eagerly; it is captured when it comes back
                while len(self._seen) > 64:
                    self._seen.popitem(last=False)
                return self._eager(x)
            hit = self._capture(x)
            self._graphs[key] = hit
            while len(self._graphs) > self.MAX_GRAPHS:
                self._graphs.popitem(last=False)
        self._graphs.move_to_end(key)
        g, static_x, static_out, _ = hit
        static_x.copy_(x)
        g.replay()
        return static_out


class GraphedStep:
    """One whole step (forward_streams, every sub-batch stream if there are several) captured into ONE HIP graph for a fixed
    input shape and replayed: the ~2,400 launches of a B = 8 720p step become one hipGraphLaunch, which takes the Python launch
    work off the host - what matters when eight ranks share one host (bench.py --graph).  The result tensor is overwritten by
    the next call."""

    def __init__(self, net, lpnet, n_streams=1):
        _check_streams(n_streams)
        self.net, self.lpnet, self.n = net, lpnet, n_streams
        self._g = None
        self._key = None
        self.captures = 0

    def __call__(self, x):
        # the graph holds raw pointers to the weights and their derived operands and was recorded in one storage mode on one
        # device: re-capture when any of them changed (as GraphedForward does), not only when the input shape did
        key = (tuple(x.shape), x.device.index, weights_signature(self.net, self.lpnet))
        if self._g is None or self._key != key:
            self._key = key
            self.captures += 1
            self._x = x.clone()
            for _ in range(2):                       # FFT tables, weight caches, stream pool: built outside the capture
                forward_streams(self.net, self.lpnet, self._x, self.n)
            torch.cuda.synchronize(x.device)
            self._g = torch.cuda.CUDAGraph()
            self._keep = {}
            with _capturing(self._g):
                self._out = forward_streams(self.net, self.lpnet, self._x, self.n, keep=self._keep)
        self._x.copy_(x)
        self._g.replay()
        return self._out

    @property
    def ratio(self):
        """LPNet's output of the last replay (a tensor of the graph's pool, overwritten by the next call); None with several streams."""
        return self._keep.get("ratio") if self._g is not None else None


GRAPH_BELOW_PIXELS = 1 << 20          # B*H*W under which a forward is launch-bound (256 x 256: 23 ms eager vs ~6 ms of kernels)


def run(net, lpnet, x):
    """The default way to run LPNet -> FDN on one GPU: hipGraph replay for small inputs (launch-bound) whose shape recurs, the
    eager forward on the caller's stream for large ones.  Returns result [B,3,H,W] (valid until the next call for the graph
    path).  The GraphedForward lives on the model object (it dies with it)."""
    B, _, H, W = x.shape
    if B * H * W < GRAPH_BELOW_PIXELS:
        g = net.__dict__.get("_fdn_graphed")
        if g is None or g.lpnet is not lpnet or g.net is not net:
            g = net.__dict__["_fdn_graphed"] = GraphedForward(net, lpnet)
        return g(x)
    return forward_streams(net, lpnet, x, 1)
