"""fdn_fdsa_full - the whole FDSA sub-block in one launch (csrc/fdsa_full.hip) - against the two-kernel route it replaces
(fdn_fdsa_fused -> fdn_fdsa_out, same C ABI, same inputs), against the float64 oracle with the conditioning-aware bound of the
block tests, and on inputs built to break the pivot-shifted LayerNorm algebra (large channel means, constant channels, dark and
zero patches)."""
import pytest
import torch

import fdn_oracle as O
from common import assert_close_cond, fixture, fixture_weights, rel_rms

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    import fdn_hip
    fdn_hip.lib()
    from basicsr.models.archs import FDN_arch
    return FDN_arch


def dev(t):
    return t.to("cuda:0").contiguous()


def _rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def _weights(C, seed, gain=1.0, beta=0.1):
    E = int(C * 1.2)
    return {
        "to_hidden.weight": _rnd(4 * E, C, 1, 1, seed=seed) / C ** 0.5,
        "to_hidden_dw.weight": _rnd(4 * E, 1, 3, 3, seed=seed + 1) / 3,
        "project_out.weight": _rnd(C, 3 * E, 1, 1, seed=seed + 2) / (3 * E) ** 0.5,
        "fft": _rnd(E, 1, 1, 8, 5, seed=seed + 3) * 0.2 + 1.0,
        **{f"norm{i}.body.weight": _rnd(E, seed=seed + 3 + i) * 0.2 + gain for i in (1, 2, 3)},
        **{f"norm{i}.body.bias": _rnd(E, seed=seed + 6 + i) * beta for i in (1, 2, 3)},
    }


def _run(A, C, x, sd, ln, res, full):
    from fdn_hip import ops
    m = A.FDSA(C)
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda:0").eval()
    xd = dev(x)
    lnp = None
    if ln is not None:
        lnp = (ops.chan_stats(xd), dev(ln[0]), dev(ln[1]))
    old = ops.FDSA_FULL, ops.FDSA_FULL_MAX_C
    ops.FDSA_FULL, ops.FDSA_FULL_MAX_C = full, 64          # (every width the kernel has a form for, not only the ones the product routes to it)
    try:
        with torch.no_grad():
            y = m.fused(xd, ln=lnp, res=xd if res else None)
    finally:
        ops.FDSA_FULL, ops.FDSA_FULL_MAX_C = old
    torch.cuda.synchronize()
    return y


def _truth(C, x, sd, ln, res):
    D = torch.float64
    P = {"a." + k: v.to(D) for k, v in sd.items()}
    xin = O.ln_chan(x.to(D), ln[0].to(D), ln[1].to(D)) if ln is not None else x.to(D)
    y = O.fdsa(xin, P, "a")
    return y + x.to(D) if res else y


@pytest.mark.parametrize("C,H,W,B,ln,res", [(32, 32, 64, 2, True, True), (32, 40, 48, 1, True, True), (32, 16, 32, 1, False, False),
                                            (64, 16, 40, 2, True, True), (64, 24, 24, 1, False, True), (24, 24, 80, 2, True, True),
                                            (48, 8, 8, 3, True, True), (32, 8, 16, 1, True, False)])
def test_fdsa_full_equals_pair_and_oracle(A, C, H, W, B, ln, res):
    from fdn_hip import ops
    x = _rnd(B, C, H, W, seed=1) * 1.5 + 0.3
    sd = _weights(C, seed=10 + C)
    lnw = (_rnd(C, seed=3) * 0.2 + 1.0, _rnd(C, seed=4) * 0.1) if ln else None
    got = _run(A, C, x, sd, lnw, res, True)
    ref = _run(A, C, x, sd, lnw, res, False)
    assert torch.isfinite(got).all()
    t64 = _truth(C, x, sd, lnw, res)
    e_got, e_ref = assert_close_cond(got, ref, t64, f"fdsa_full C={C} {H}x{W}")
    print(f"fdsa_full C={C} {H}x{W}: relative RMS error vs float64: one launch {e_got:.2e}, two launches {e_ref:.2e}")
    if res:                                     # the LayerNorm statistics of the result travel with it
        st, st_ref = got._fdn_stats.cpu(), ops.chan_stats(got).cpu()
        assert rel_rms(st, st_ref) < 1e-5


def test_fdsa_full_is_the_route_taken(A):
    """The module must really have taken the one-launch route in the test above (a silent fallback would make it vacuous)."""
    from fdn_hip import ops
    calls = []
    orig = ops.fdsa_full

    def spy(*a, **k):
        y = orig(*a, **k)
        calls.append(y is not None)
        return y
    ops.fdsa_full = spy
    try:
        for C, H, W in ((32, 16, 32), (64, 16, 24), (24, 8, 16), (48, 8, 8)):
            _run(A, C, _rnd(1, C, H, W, seed=2), _weights(C, seed=5), None, True, True)
    finally:
        ops.fdsa_full = orig
    assert calls == [True] * 4


@pytest.mark.parametrize("C", [32, 64])
def test_fdsa_full_pivot_stress(A, C):
    """Inputs that make the LayerNorm groups badly scaled: norm gains / biases far from 1 / 0, an input with a large common offset,
    constant and zero patches, a dark region.  The one-launch route is held to the same float64-referenced bound as the
    two-launch route (which evaluates the LayerNorms directly)."""
    H, W, B = 32, 48, 2
    x = _rnd(B, C, H, W, seed=31) * 0.7
    x[:, :, :8, :16] = 0.0                       # zero patches: every spectrum bin replaced by 1e-10 (1 + i)
    x[:, :, 8:16, :8] = 2.5                      # constant patch
    x[:, :, 16:, 24:] *= 1e-3                    # dark region
    x[1] += 4.0                                  # large common offset on one image
    sd = _weights(C, seed=77, gain=3.0, beta=2.0)
    for res in (True, False):
        got = _run(A, C, x, sd, None, res, True)
        ref = _run(A, C, x, sd, None, res, False)
        assert torch.isfinite(got).all()
        assert_close_cond(got, ref, _truth(C, x, sd, None, res), f"fdsa_full stress C={C} res={res}")


@pytest.mark.parametrize("name,c", [("fdsa_c32", 32), ("fdsa_c64", 64), ("fdsa_c32_edge", 32)])
def test_fdsa_full_reference_fixtures(A, name, c, monkeypatch):
    """The reference's own outputs (tests/golden) through the one-launch route, with the block tests' bound."""
    from fdn_hip import ops
    monkeypatch.setattr(ops, "FDSA_FULL", True)
    monkeypatch.setattr(ops, "FDSA_FULL_MAX_C", 64)
    fx = fixture(name)
    sd = fixture_weights(name, fx["shapes"])
    m = A.FDSA(c)
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda:0").eval()
    with torch.no_grad():
        got = m(dev(fx["x"]))
    D = torch.float64
    t64 = O.fdsa(fx["x"].to(D), {"." + k: v.to(D) for k, v in sd.items()}, "")
    assert_close_cond(got, fx["y"], t64, name)


def test_fdsa_full_batch_slice_and_bitstable(A, monkeypatch):
    """x as a batch slice of a larger tensor; two runs bit-identical; batch of 3 equals 3 singles bit for bit."""
    from fdn_hip import ops
    C, H, W = 32, 16, 48
    big = dev(_rnd(5, C, H, W, seed=11))
    sd = _weights(C, seed=3)
    m = A.FDSA(C)
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda:0").eval()
    x = big[1:4]
    monkeypatch.setattr(ops, "FDSA_FULL", True)
    with torch.no_grad():
        st = ops.chan_stats(x.contiguous())
        lnp = (st, dev(torch.ones(C)), dev(torch.zeros(C)))
        a = m.fused(x, ln=lnp)
        b = m.fused(x, ln=lnp)
        singles = torch.cat([m.fused(x[i:i + 1].contiguous(), ln=(st[i:i + 1].contiguous(),) + lnp[1:]) for i in range(3)])
    assert torch.equal(a, b) and torch.equal(a, singles)
