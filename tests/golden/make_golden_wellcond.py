"""A second 96 x 160 end-to-end frame, CHOSEN to be well-conditioned, and the block-noise conditioning rows of both small frames
(build container only: runs the reference once and the float64 oracle a few hundred times, ~1 h on 4 threads).

    python tests/golden/make_golden_wellcond.py search [first_seed] [n_seeds]   # prints one score line per candidate seed
    python tests/golden/make_golden_wellcond.py write <seed>                    # fdn_tamed_96x160_wc.npz + _cond.npz from the reference
    python tests/golden/make_golden_wellcond.py blocknoise fdn_tamed_96x160     # adds <key>_blocknoise rows to an existing _cond.npz

Why: with the tamed synthetic weights a random 96 x 160 frame has spots where a spectrum bin of an FDSA block sits within fp32 rounding of
zero, so its phase - and 1e-4 of a 16 x 16 window of y - is decided by the last bits of whichever arithmetic evaluates it (parity
policy).  `fdn_tamed_96x160` is such a frame and is held per window (test_fdn_end_to_end_tamed_conditioning); a fixed PSNR floor only makes
sense on a frame WITHOUT such spots.  The model that decides "without" is independent of any fp32 implementation:
  * block noise   float64 oracle, white noise of `level` x rms(t) added to the output t of every TransformerBlock (the size of one fp32
                  rounding per block: 2e-7 reproduces the reference's own median window error of 3e-8; 3e-7 is the stricter search level);
  * FFT noise     float64 oracle, every forward rfft2 fed t + 2e-7 rms(t) randn (make_golden_small_cond.py);
  * fp32 oracle   the frame and one-ulp perturbations of it.
score(seed) = the largest per-window RMS deviation of y from the float64 truth over all of these evaluations.  The search takes candidates
in seed order (a candidate is dropped at its first evaluation above the threshold) and the frame written is the first whose score stays
below THRESH = 3e-6 over the long confirmation run too (a window error of 3e-6 in EVERY window would still be 110 dB; the typical window of
such a frame sits at 3e-8).  Frames below 1e-6 were not found: with these weights nearly every random 96 x 160 frame has a spot that rounding-sized
noise moves by 1e-6 .. 1e-3 (the search log, committed as tests/golden/wellcond_search.txt, shows the distribution).  Nothing here is read by the product path.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import fdn_oracle as O  # noqa: E402
from common import fdn_weights, fixture  # noqa: E402

KEYS, WIN = ("y", "q1", "q2", "q3"), (16, 16, 8, 4)
TAME = 0.03
THRESH = 3e-6
SHAPE = (1, 3, 96, 160)


def window_rms(d, size):
    B, C, H, W = d.shape
    return d.double().pow(2).reshape(B, C, H // size, size, W // size, size).mean((1, 3, 5)).sqrt().reshape(-1)


def frame(seed):
    """the candidate of a seed: uniform [0, 1) image, ratio in [0.3, 0.8) (the recipe of make_golden.py with the seed as the variable)"""
    x = torch.rand(*SHAPE, generator=torch.Generator().manual_seed(seed))
    ratio = 0.3 + 0.5 * torch.rand(SHAPE[0], 1, generator=torch.Generator().manual_seed(seed + 100000))
    return x, ratio


def with_block_noise(level, seed, fn):
    """run fn() with white noise of level * rms added to every TransformerBlock output of the oracle (oracle.tblock is looked up at call time)"""
    gen = torch.Generator().manual_seed(seed)
    real = O.tblock

    def noisy(x, *a, **kw):
        t = real(x, *a, **kw)
        rms = t.pow(2).mean().sqrt()
        return t + level * rms * torch.randn(t.shape, generator=gen, dtype=t.dtype)
    O.tblock = noisy
    try:
        with torch.no_grad():
            return fn()
    finally:
        O.tblock = real


def with_fft_noise(level, seed, fn):
    gen = torch.Generator().manual_seed(seed)
    real = torch.fft.rfft2

    def noisy(t, *a, **kw):
        rms = t.pow(2).mean(dim=(-2, -1), keepdim=True).sqrt()
        return real(t + level * rms * torch.randn(t.shape, generator=gen, dtype=t.dtype), *a, **kw)
    torch.fft.rfft2 = noisy
    try:
        with torch.no_grad():
            return fn()
    finally:
        torch.fft.rfft2 = real


def block_noise_rows(P64, x, ratio, truth, plan):
    """plan: [(level, first_seed, count)] -> {key: [n, windows]} and the list of (level, seed) per row"""
    rows, tags = {k: [] for k in KEYS}, []
    for level, s0, cnt in plan:
        for s in range(s0, s0 + cnt):
            outs = with_block_noise(level, s, lambda: O.fdn_forward(P64, x.double(), ratio.double()))
            for key, t, tr, size in zip(KEYS, outs, truth, WIN):
                rows[key].append(window_rms(t - tr, size).numpy())
            tags.append((level, s))
    return {k: np.stack(v) for k, v in rows.items()}, tags


def score(seed, P32, P64, n_block=8, n_fft=2, n_f32=3, level=3e-7, stop_above=None):
    x, ratio = frame(seed)
    with torch.no_grad():
        truth = O.fdn_forward(P64, x.double(), ratio.double())[0]
    worst, what = 0.0, ""

    def take(y, tag):
        nonlocal worst, what
        e = float(window_rms(y.double() - truth, 16).max())
        if e > worst:
            worst, what = e, tag
    for k in range(n_f32):
        xin = x if k == 0 else x + 6e-8 * torch.randn(x.shape, generator=torch.Generator().manual_seed(100 + k))
        with torch.no_grad():
            take(O.fdn_forward(P32, xin, ratio)[0], f"fp32#{k}")
        if stop_above and worst > stop_above:
            return worst, what
    for k in range(n_block):
        take(with_block_noise(level, 900 + k, lambda: O.fdn_forward(P64, x.double(), ratio.double()))[0], f"block#{k}")
        if stop_above and worst > stop_above:
            return worst, what
    for k in range(n_fft):
        take(with_fft_noise(2e-7, 500 + k, lambda: O.fdn_forward(P64, x.double(), ratio.double()))[0], f"fft#{k}")
    return worst, what


def cmd_search(first, count):
    P32 = fdn_weights(tame=TAME)
    P64 = O.cast_params(P32, torch.float64)
    for seed in range(first, first + count):
        w, what = score(seed, P32, P64, stop_above=THRESH)
        print("seed %d: worst window of y %s%.2e (%s)%s" % (seed, ">= " if w > THRESH else "", w, what, "" if w > THRESH else "   <-- below %.0e: confirming with 32 block-noise seeds" % THRESH), flush=True)
        if w <= THRESH:
            w2, what2 = score(seed, P32, P64, n_block=32, n_fft=6, n_f32=5, stop_above=None)
            print("seed %d: confirmed %.2e (%s)" % (seed, w2, what2), flush=True)
            if w2 <= THRESH:
                print("chosen seed", seed, flush=True)
                return seed
    return None


def cmd_write(seed):
    from _refload import build_ref_fdn, quiet
    from weights import shapes_of, synth_state_dict
    from common import SEED
    x, ratio = frame(seed)
    net = build_ref_fdn(0)
    net.load_state_dict(synth_state_dict(shapes_of(net), SEED, prefix_key="fdn/", tame=TAME), strict=True)
    torch.set_num_threads(8)
    with torch.no_grad(), quiet():
        outs = net(x, ratio_i=ratio)
    name = "fdn_tamed_96x160_wc"
    np.savez_compressed(os.path.join(HERE, name + ".npz"), x=x.numpy(), ratio=ratio.numpy(), y=outs[0].numpy(), q1=outs[1].numpy(), q2=outs[2].numpy(),
                        q3=outs[3].numpy(), tame=np.float32(TAME), frame_seed=np.int64(seed))
    print("wrote", name)
    import make_golden_small_cond
    make_golden_small_cond.main(name)
    cmd_blocknoise(name)


def cmd_blocknoise(name, plan=((2e-7, 2000, 64), (3e-7, 3000, 32))):
    fx = fixture(name)
    P64 = O.cast_params(fdn_weights(tame=float(fx["tame"])), torch.float64)
    z = dict(np.load(os.path.join(HERE, name + "_cond.npz")))
    truth = [torch.from_numpy(z[k + "_f64"]) for k in KEYS]
    rows, tags = block_noise_rows(P64, fx["x"], fx["ratio"], truth, plan)
    for k in KEYS:
        z[k + "_blocknoise"] = rows[k]
    z["blocknoise_level_seed"] = np.array(tags, dtype=np.float64)
    y = rows["y"]
    for (lv, s), r in zip(tags, y):
        top = np.argsort(-r)[:4]
        if r[top[0]] > 1e-6:
            print(name, "block noise %.0e seed %d: windows" % (lv, s), [(int(i), float("%.1e" % r[i])) for i in top], flush=True)
    print(name, "block noise: median window error per level", {lv: float("%.2e" % np.median(y[[i for i, t in enumerate(tags) if t[0] == lv]])) for lv in sorted({t[0] for t in tags})})
    np.savez_compressed(os.path.join(HERE, name + "_cond.npz"), **z)
    print("updated", name + "_cond.npz")


if __name__ == "__main__":
    torch.set_num_threads(int(os.environ.get("FDN_GOLDEN_THREADS", "4")))
    cmd = sys.argv[1]
    if cmd == "search":
        cmd_search(int(sys.argv[2]) if len(sys.argv) > 2 else 200, int(sys.argv[3]) if len(sys.argv) > 3 else 60)
    elif cmd == "write":
        cmd_write(int(sys.argv[2]))
    elif cmd == "blocknoise":
        cmd_blocknoise(sys.argv[2])
