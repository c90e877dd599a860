"""bf16-STORAGE mode (BASELINE.json configs[2]; the reference forces fp32 before every FFT, FDN_arch.py:411,460,585-589, so
this configuration is defined by the build: DESIGN.md section 3).  Block-internal activations of the FDSA / FDFFN blocks of
levels 1-2 travel between kernels as bf16; all arithmetic is fp32.

Two kinds of checks:
  * exactness of the format: a kernel reading bf16 equals the fp32 kernel fed the same (already rounded) values bit for bit,
    and a kernel writing bf16 equals round-to-nearest-even of its fp32 output bit for bit - bf16 is storage, nothing else;
  * the accuracy that storage format costs against the fp32 oracle, per block and end to end (tolerances stated below).
"""
import pytest
import torch

import fdn_oracle as O
from common import fdn_weights, fixture, fixture_weights, rel_rms

pytestmark = pytest.mark.gpu
BF = torch.bfloat16

# tolerances of the bf16-storage configuration against the fp32 oracle (measured on MI355X: see DESIGN.md section 2)
BLOCK_REL_RMS = 6e-3          # one FDSA / FDFFN block, relative RMS error of its output (bf16 has 8 significant bits: 2^-9 = 2e-3 per store)
E2E_PSNR_DB = 55.0            # tamed end-to-end FDN, PSNR of the restored image against the fp32 reference fixture


@pytest.fixture(scope="module")
def A():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    import fdn_hip
    fdn_hip.lib()
    from basicsr.models.archs import FDN_arch
    yield FDN_arch
    fdn_hip.set_storage_dtype("f32")


@pytest.fixture(autouse=True)
def _fp32_after():
    yield
    import fdn_hip
    fdn_hip.set_storage_dtype("f32")


def dev(t):
    return t.to("cuda:0").contiguous()


def load(mod, sd):
    mod.load_state_dict(sd, strict=True)
    return mod.to("cuda:0").eval()


def _rnd(*s, seed):
    return torch.randn(*s, generator=torch.Generator().manual_seed(seed))


@pytest.mark.parametrize("C,H,W", [(32, 32, 64), (64, 24, 40), (24, 16, 24), (48, 8, 16)])
def test_bf16_is_storage_only(A, C, H, W):
    """Every bf16 load form reads exactly the stored values; every bf16 store form is round-to-nearest-even of the fp32 result."""
    from fdn_hip import ops
    E, Hd, B = int(C * 1.2), int(C * 2.7), 2
    x = dev(_rnd(B, C, H, W, seed=1))
    st = ops.chan_stats(x)
    g, b_ = dev(_rnd(C, seed=2) * 0.1 + 1), dev(_rnd(C, seed=3) * 0.1)
    # project_in: bf16 out
    wi = dev(_rnd(Hd, C, seed=4) / C ** 0.5)
    h32 = ops.conv1x1(x, wi, ln=(st, g, b_))
    h16 = ops.conv1x1(x, wi, ln=(st, g, b_), out_dtype=BF)
    assert h16.dtype == BF and torch.equal(h16, h32.to(BF))
    # fdffn_mid: bf16 in / out
    w0, w2 = dev(_rnd(Hd, 1, 3, 3, seed=5) / 3), dev(_rnd(Hd, 1, 3, 3, seed=6) / 3)
    fa, fp = dev(_rnd(Hd, 1, 1, 8, 5, seed=7) * 0.2 + 1), dev(_rnd(Hd, 1, 1, 8, 5, seed=8) * 0.5)
    y32 = ops.fdffn_mid(h16.float(), w0, w2, fa, fp)
    assert torch.equal(ops.fdffn_mid(h16, w0, w2, fa, fp, out_dtype=torch.float32), y32)
    y16 = ops.fdffn_mid(h16, w0, w2, fa, fp)
    assert y16.dtype == BF and torch.equal(y16, y32.to(BF))
    # gate: bf16 in / out
    wg = dev(_rnd(2 * Hd, 1, 3, 3, seed=9) / 3)
    g32 = ops.dwconv_gate(y16.float(), wg)
    assert torch.equal(ops.dwconv_gate(y16, wg, out_dtype=torch.float32), g32)
    g16 = ops.dwconv_gate(y16, wg)
    assert g16.dtype == BF and torch.equal(g16, g32.to(BF))
    # project_out: bf16 in (narrow TAIL form at C <= 32, K-streaming form above), residual + statistics
    wo = dev(_rnd(C, Hd, seed=10) / Hd ** 0.5)
    o32 = ops.conv1x1(g16.float(), wo, res=x, want_stats=True)
    o16 = ops.conv1x1(g16, wo, res=x, want_stats=True)
    assert o16.dtype == torch.float32 and torch.equal(o16, o32) and torch.equal(o16._fdn_stats, o32._fdn_stats)
    # FDSA: fused front half writes bf16, the tail reads it
    wh = dev(_rnd(4 * E, C, seed=11) / C ** 0.5)
    dw, fw = dev(_rnd(4 * E, 1, 3, 3, seed=12) / 3), dev(_rnd(E, 1, 1, 8, 5, seed=13) * 0.2 + 1)
    wpk = ops.fdsa_pack(wh, g, b_)
    a32 = ops.fdsa_fused(x, st, wpk, dw, fw)
    a16 = ops.fdsa_fused(x, st, wpk, dw, fw, out_dtype=BF)
    assert a16.dtype == BF and torch.equal(a16, a32.to(BF))
    wp = dev(_rnd(C, 3 * E, seed=14) / (3 * E) ** 0.5)
    g3, b3 = dev(_rnd(3 * E, seed=15) * 0.1 + 1), dev(_rnd(3 * E, seed=16) * 0.1)
    t32 = ops.fdsa_out(a16.float(), wp, g3, b3, res=x, want_stats=True)
    t16 = ops.fdsa_out(a16, wp, g3, b3, res=x, want_stats=True)
    assert t16 is not None and torch.equal(t16, t32) and torch.equal(t16._fdn_stats, t32._fdn_stats)


@pytest.mark.parametrize("name,cls,c", [("fdsa_c32", "FDSA", 32), ("fdsa_c64", "FDSA", 64), ("fdffn_c32", "FDFFN", 32), ("fdffn_c64", "FDFFN", 64)])
def test_bf16_block_accuracy(A, name, cls, c):
    """One block in bf16-storage mode against the fp32 reference fixture."""
    import fdn_hip
    fx = fixture(name)
    sd = fixture_weights(name, fx["shapes"])
    m = load(getattr(A, cls)(c), sd)
    fdn_hip.set_storage_dtype("bf16")
    with torch.no_grad():
        got = m(dev(fx["x"]))
    err = rel_rms(got.cpu(), fx["y"])
    print(f"bf16 storage {name}: relative RMS error {err:.2e}")
    assert got.dtype == torch.float32 and err < BLOCK_REL_RMS, err


@pytest.mark.parametrize("name", ["fdn_tamed_64", "fdn_tamed_96x160"])
def test_bf16_end_to_end(A, name):
    import fdn_hip
    fx = fixture(name)
    m = load(A.FDN(), fdn_weights(tame=float(fx["tame"])))
    fdn_hip.set_storage_dtype("bf16")
    with torch.no_grad():
        outs = m(dev(fx["x"]), ratio_i=dev(fx["ratio"]))
    p = O.psnr(outs[0].cpu(), fx["y"])
    print(f"bf16 storage {name}: PSNR {p:.1f} dB vs the fp32 reference fixture")
    assert p > E2E_PSNR_DB, p
    for got, key in zip(outs[1:], ("q1", "q2", "q3")):          # MAR does not use bf16 storage: unchanged
        assert O.psnr(got.cpu(), fx[key]) > 100.0, key


def test_bf16_mode_moves_fewer_bytes_not_different_launches(A):
    """The mode changes tensor dtypes only: same entry points, bf16 tensors between them at levels 1-2, fp32 at level 3."""
    import fdn_hip
    from fdn_hip import ops
    assert ops.block_storage(32, 64) == torch.float32
    fdn_hip.set_storage_dtype("bf16")
    assert ops.block_storage(32, 64) == BF and ops.block_storage(64, 64) == BF and ops.block_storage(128, 64) == torch.float32
    with pytest.raises(ValueError):
        fdn_hip.set_storage_dtype("fp8")
