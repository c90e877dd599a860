"""Generate the committed golden fixtures by RUNNING THE REFERENCE (build container only).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz, selfnoise.json

The reference (/root/reference, read-only, never shipped) is imported by path with the shims in
_refload.py.  Each fixture holds inputs, the reference's fp32 outputs and the {key: shape} table
from which tests/weights.py regenerates the exact synthetic weights; LPNet fixtures use the real
checkpoint/LPNet_lolblur.pth, re-exported losslessly as lpnet_lolblur_params.npz (data, 414 k
fp32 values).  Nothing here is read by the product path.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))

from _refload import REF_ROOT, build_ref_fdn, quiet, ref_fdn_module, ref_lpnet_module  # noqa: E402
from weights import shapes_of, synth_state_dict  # noqa: E402
import fdn_oracle as O  # noqa: E402

SEED = 7


def rnd(*shape, seed, lo=0.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return lo + (hi - lo) * torch.rand(*shape, generator=g)


def rndn(*shape, seed, std=1.0):
    g = torch.Generator().manual_seed(seed)
    return std * torch.randn(*shape, generator=g)


def save(name, shapes, **arrs):
    out = {k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    out["shapes_json"] = np.frombuffer(json.dumps({k: list(s) for k, s in shapes.items()}).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: tuple(v.shape) for k, v in out.items() if k != "shapes_json"})


def load_synth(mod, name, tame=None):
    shapes = shapes_of(mod)
    sd = synth_state_dict(shapes, SEED, prefix_key=name + "/", tame=tame)
    mod.load_state_dict(sd, strict=True)
    return shapes


def run2(fn):
    """Run fn at 8 threads and at 1 thread -> (out8, self-PSNR list)."""
    torch.set_num_threads(8)
    with torch.no_grad(), quiet():
        o8 = fn()
    torch.set_num_threads(1)
    with torch.no_grad(), quiet():
        o1 = fn()
    torch.set_num_threads(8)
    o8 = o8 if isinstance(o8, (tuple, list)) else (o8,)
    o1 = o1 if isinstance(o1, (tuple, list)) else (o1,)
    return o8, [O.psnr(a, b) for a, b in zip(o8, o1)]


def main():
    R = ref_fdn_module()
    noise = {}
    torch.manual_seed(0)

    # ---- FDSA / FDFFN at the three channel widths ------------------------------------
    for c, shp in ((32, (2, 32, 32, 32)), (64, (1, 64, 16, 24)), (128, (1, 128, 16, 16))):
        for cls, tag in ((R.FDSA, "fdsa"), (R.FDFFN, "fdffn")):
            name = f"{tag}_c{c}"
            m = cls(c, False).eval()
            shapes = load_synth(m, name)
            x = rndn(*shp, seed=c + 1)
            (y,), n = run2(lambda: m(x))
            noise[name] = n
            save(name, shapes, x=x, y=y)

    # ---- FCAFFN: power-of-two, radix-3/5 and radix-23 sizes ---------------------------
    for c, shp in ((32, (2, 32, 32, 32)), (32, (1, 32, 24, 40)), (64, (1, 64, 46, 40)), (128, (1, 128, 16, 16))):
        b, _, h, w = shp
        name = f"fcaffn_c{c}_{h}x{w}"
        m = R.FCAFFN(c, False).eval()
        shapes = load_synth(m, name)
        x = rndn(*shp, seed=3)
        amp = rnd(b, 3, h, w // 2 + 1, seed=4, lo=0.0, hi=30.0)
        pha = rnd(b, 3, h, w // 2 + 1, seed=5, lo=-3.1, hi=3.1)
        img = rnd(b, 3, h, w, seed=6)
        (y,), n = run2(lambda: m(x, amp, pha, img))
        noise[name] = n
        save(name, shapes, x=x, amp=amp, pha=pha, img=img, y=y)

    # ---- TransformerBlock (encoder / decoder flavour), Fuse, Down/Upsample -------------
    for tag, light in (("enc", True), ("dec", False)):
        name = f"tblock_{tag}_c32"
        m = R.TransformerBlock(dim=32, att=True, use_light=light, use_img=light).eval()
        shapes = load_synth(m, name, tame=None)
        sd = m.state_dict()                          # keep the block well conditioned
        for k in sd:
            if k.endswith("project_out.weight"):
                sd[k] = sd[k] * 0.2
        m.load_state_dict(sd)
        x = rndn(2, 32, 32, 32, seed=11)
        amp = rnd(2, 3, 32, 17, seed=12, hi=30.0)
        pha = rnd(2, 3, 32, 17, seed=13, lo=-3.1, hi=3.1)
        img = rnd(2, 3, 32, 32, seed=14)
        (y,), n = run2(lambda: m((x, amp, pha, img))[0])
        noise[name] = n
        save(name, shapes, x=x, amp=amp, pha=pha, img=img, y=y, po_scale=np.float32(0.2))

    name = "fuse_n32"
    m = R.Fuse(32).eval()
    shapes = load_synth(m, name)
    enc, dnc = rndn(1, 32, 16, 24, seed=21), rndn(1, 32, 16, 24, seed=22)
    (y,), n = run2(lambda: m(enc, dnc, None, None, None))
    noise[name] = n
    save(name, shapes, enc=enc, dnc=dnc, y=y)

    for cls, name, shp in ((R.Downsample, "downsample_c32", (2, 32, 16, 24)), (R.Upsample, "upsample_c64", (2, 64, 8, 12))):
        m = cls(shp[1]).eval()
        shapes = load_synth(m, name)
        x = rndn(*shp, seed=31)
        (y,), n = run2(lambda: m(x))
        noise[name] = n
        save(name, shapes, x=x, y=y)

    name = "patch_embed_3_32"
    m = R.OverlapPatchEmbed(3, 32).eval()
    shapes = load_synth(m, name)
    x = rnd(2, 3, 16, 24, seed=32)
    (y,), n = run2(lambda: m(x))
    save(name, shapes, x=x, y=y)

    # ---- MAR pieces ------------------------------------------------------------------
    name = "freblock_c12"
    m = R.FreBlock(12).eval()
    shapes = load_synth(m, name)
    x = rndn(1, 12, 24, 40, seed=41, std=0.5)
    (y,), n = run2(lambda: m(x))
    noise[name] = n
    save(name, shapes, x=x, y=y)

    name = "fourier_fuse_84_12"
    m = R.fourier_fuse(84, 12).eval()
    shapes = load_synth(m, name)
    x1, x2, x4 = rndn(1, 12, 16, 24, seed=42, std=0.5), rndn(1, 24, 16, 24, seed=43, std=0.5), rndn(1, 48, 16, 24, seed=44, std=0.5)
    (y,), n = run2(lambda: m(x1, x2, x4))
    noise[name] = n
    save(name, shapes, x1=x1, x2=x2, x4=x4, y=y)

    name = "mar_full"
    m = R.MAR(True).eval()
    shapes = load_synth(m, name)
    x = rnd(2, 3, 64, 96, seed=45)
    ratio = torch.tensor([[0.62], [0.35]]).view(2, 1, 1, 1)
    outs, n = run2(lambda: m(x, ratio))
    noise[name] = n
    save(name, shapes, x=x, ratio=ratio.view(2, 1), y3=outs[0], y2=outs[1], y1=outs[2])

    # ---- LPNet with the real checkpoint ------------------------------------------------
    lp = ref_lpnet_module().I_predict_net().eval()
    lsd = torch.load(os.path.join(REF_ROOT, "checkpoint", "LPNet_lolblur.pth"), map_location="cpu")["params"]
    lp.load_state_dict(lsd, strict=True)
    np.savez_compressed(os.path.join(HERE, "lpnet_lolblur_params.npz"), **{k: v.numpy() for k, v in lsd.items()})
    x = rnd(2, 3, 96, 128, seed=51)
    (y,), n = run2(lambda: lp(x))
    noise["lpnet_real"] = n
    x2 = rnd(1, 3, 736, 1280, seed=52)
    with torch.no_grad():
        y2 = lp(x2)
    save("lpnet_real", shapes_of(lsd), x=x, y=y, y_736x1280_seed52=y2)

    # ---- end to end, tamed synthetic weights -----------------------------------------
    net = build_ref_fdn(0)
    shapes = shapes_of(net)
    for name, shp, tame in (("fdn_tamed_64", (2, 3, 64, 64), 0.03), ("fdn_tamed_96x160", (1, 3, 96, 160), 0.03)):
        net.load_state_dict(synth_state_dict(shapes, SEED, prefix_key="fdn/", tame=tame), strict=True)
        x = rnd(*shp, seed=61)
        ratio = rnd(shp[0], 1, seed=62, lo=0.3, hi=0.8)
        outs, n = run2(lambda: net(x, ratio_i=ratio))
        noise[name] = n
        save(name, {} if name != "fdn_tamed_64" else shapes, x=x, ratio=ratio, y=outs[0], q1=outs[1], q2=outs[2], q3=outs[3],
             tame=np.float32(tame))

    # untamed self-noise of the reference (documentation of ill-conditioning, fact 9)
    net.load_state_dict(synth_state_dict(shapes, SEED, prefix_key="fdn/", tame=None), strict=True)
    x = rnd(1, 3, 64, 64, seed=61)
    ratio = torch.tensor([[0.5]])
    _, n = run2(lambda: net(x, ratio_i=ratio))
    noise["fdn_untamed_64 (not a fixture)"] = n

    # ---- caller harness: uint8 in -> uint8 out -----------------------------------------
    net.load_state_dict(synth_state_dict(shapes, SEED, prefix_key="fdn/", tame=0.03), strict=True)
    g = torch.Generator().manual_seed(71)
    img = torch.randint(0, 256, (70, 90, 3), generator=g, dtype=torch.uint8).numpy()
    # reference steps, inference_fdn_lolblur.py:47-75 (cv2 absent: BGR->RGB is a channel flip)
    t = torch.from_numpy(np.ascontiguousarray((img.astype(np.float32) / 255.0)[:, :, ::-1].transpose(2, 0, 1))).float().unsqueeze(0)
    h, w = t.shape[-2:]
    hn, wn = (32 - h % 32) % 32, (32 - w % 32) % 32
    tp = torch.nn.functional.pad(t, (0, wn, 0, hn), mode="reflect")
    with torch.no_grad(), quiet():
        ratio = lp(tp)
        res = net(tp, ratio_i=ratio)[0][:, :, :h, :w]
    r = res.squeeze(0).float().clamp_(0, 1).numpy().transpose(1, 2, 0)[:, :, ::-1]
    out_u8 = (r * 255.0).round().astype(np.uint8)
    save("harness_u8", {}, img=img, padded=tp, ratio=ratio, result=res, out_u8=out_u8, tame=np.float32(0.03))

    with open(os.path.join(HERE, "selfnoise.json"), "w") as f:
        json.dump({"what": "reference 8-thread vs 1-thread PSNR (dB) per fixture output", "psnr": noise}, f, indent=1)
    print(json.dumps(noise, indent=1))


if __name__ == "__main__":
    main()
