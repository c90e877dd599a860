"""fdn_fdsa_fused_tail (round 6): the FDSA sub-block in one launch WITHOUT new arithmetic - the workgroup that produced a tile's (out1|out2|out3|v_value)
planes runs fdn_fdsa_out's arithmetic on them itself, the hand-off lives in a ring of per-resident-workgroup blocks, and at level 1 the launch also runs the
project_in of the FDFFN that follows.  The contract is BIT-IDENTITY with the launches it replaces (fdn_fdsa_fused + fdn_fdsa_out, fdn_conv1x1 with the
LayerNorm prologue), on ragged sizes, with and without LayerNorm / residual / statistics, across repeated launches that share one ring; accuracy against
float64 is then the pair's (tests/test_gpu_parity.py::test_fdsa runs the module, i.e. this route)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    import fdn_hip
    fdn_hip.lib()
    from fdn_hip import ops as o
    yield o
    fdn_hip.set_matrix_pipe("bf16")


def _case(ops, B, C, H, W, seed, ln=True, res=True, edge=False):
    g = torch.Generator(device="cuda:0").manual_seed(seed)
    r = lambda *s: torch.randn(*s, device="cuda:0", generator=g)
    E, Hd = int(C * 1.2), int(C * 2.7)
    x = r(B, C, H, W)
    if edge:                                   # zero patches, tiny rows, an exact constant, -0.0: the replace_denormals paths
        x[:, :, :8, :16] = 0.0
        x[:, 1::5, 8:16, :] *= 1e-12
        x[:, :, 16:24, 8:24] = 0.5
        x[:, :, -8:, :8] = -0.0
    d = dict(x=x, stats=ops.chan_stats(x) if ln else None, gm=r(C) if ln else None, bt=r(C) if ln else None,
             wh=r(4 * E, C) / C ** .5, dw=r(4 * E, 1, 3, 3), fw=r(E, 1, 1, 8, 5), wp=r(C, 3 * E) / (3 * E) ** .5, g3=r(3 * E), b3=r(3 * E),
             wi=r(Hd, C) / C ** .5, g2=r(C), b2=r(C), res=x if res else None, E=E, Hd=Hd, C=C)
    d["wpk"] = ops.fdsa_pack(d["wh"], d["gm"], d["bt"])
    return d


def _pair(ops, d, pin=False):
    o = ops.fdsa_fused(d["x"], d["stats"], d["wpk"], d["dw"], d["fw"])
    y = ops.fdsa_out(o, d["wp"], d["g3"], d["b3"], res=d["res"], want_stats=True)
    h = ops.conv1x1(y, d["wi"], ln=(y._fdn_stats, d["g2"], d["b2"]), cache=(ops.WeightCache(), "pi")) if pin else None
    return y, y._fdn_stats, h


def _one(ops, d, pin=False, want_stats=True):
    img = ops.fdsa_tail_pack(d["wp"], d["g3"], d["b3"], d["C"], pin=ops.fold_ln(d["wi"], None, d["g2"], d["b2"]) if pin else None)
    assert img is not None
    y = ops.fdsa_fused_tail(d["x"], d["stats"], d["wpk"], d["dw"], d["fw"], img, res=d["res"], want_stats=want_stats, Hd=d["Hd"] if pin else 0)
    assert y is not None
    return y, getattr(y, "_fdn_stats", None), getattr(y, "_fdn_pin", None)


@pytest.mark.parametrize("B,C,H,W", [(1, 32, 16, 40), (2, 32, 24, 72), (1, 24, 32, 64), (3, 24, 8, 40), (1, 64, 16, 40), (2, 64, 24, 72), (2, 48, 32, 64), (1, 48, 8, 24),
                                     (2, 32, 184, 320), (1, 64, 96, 160)])
@pytest.mark.parametrize("ln,res", [(True, True), (False, False)])
def test_one_launch_equals_the_pair_bit_for_bit(ops, B, C, H, W, ln, res):
    d = _case(ops, B, C, H, W, seed=B * 1000 + C + H + W, ln=ln, res=res, edge=H >= 24)
    ya, sa, _ = _pair(ops, d)
    yb, sb, _ = _one(ops, d)
    torch.cuda.synchronize()
    assert torch.isfinite(yb).all()
    assert torch.equal(ya, yb), f"max |diff| {(ya - yb).abs().max().item():.3e}"
    assert torch.equal(sa, sb)
    yc, sc, _ = _one(ops, d, want_stats=False)              # without the statistics output: the same result
    assert sc is None and torch.equal(ya, yc)


@pytest.mark.parametrize("B,C,H,W", [(1, 32, 16, 40), (2, 32, 40, 96), (2, 24, 32, 64), (1, 24, 8, 40), (2, 32, 184, 320), (2, 64, 24, 72), (1, 64, 48, 96)])
def test_project_in_inside_the_launch_equals_conv1x1(ops, B, C, H, W):
    """Level 1 (the default route) and C = 64 (optional, ops.FDSA_TAIL_PIN_MAX_C = 64): h = project_in(LN(out)) from the registers of the tail's epilogue -
    v_permlane32_swap moves the MFMA result layout into the operand layout - equals fdn_conv1x1(FDN_PRO_LN)'s strip kernel bit for bit."""
    d = _case(ops, B, C, H, W, seed=7 * C + H + W, edge=True)
    ya, sa, ha = _pair(ops, d, pin=True)
    yb, sb, hb = _one(ops, d, pin=True)
    torch.cuda.synchronize()
    assert torch.equal(ya, yb) and torch.equal(sa, sb)
    assert hb is not None and hb.shape == ha.shape and torch.isfinite(hb).all()
    assert torch.equal(ha, hb), f"h: max |diff| {(ha - hb).abs().max().item():.3e}"


@pytest.mark.parametrize("B,C,H,W", [(2, 32, 40, 96), (1, 24, 16, 48)])
def test_project_in_stored_as_bf16_is_the_rounded_fp32_result(ops, B, C, H, W):
    """bf16-storage mode (BASELINE configs[2]): h leaves the launch as bf16 - exactly round-to-nearest-even of the fp32 h of the fp32 mode (storage only)."""
    d = _case(ops, B, C, H, W, seed=3 * C + W, edge=True)
    img = ops.fdsa_tail_pack(d["wp"], d["g3"], d["b3"], C, pin=ops.fold_ln(d["wi"], None, d["g2"], d["b2"]))
    y32 = ops.fdsa_fused_tail(d["x"], d["stats"], d["wpk"], d["dw"], d["fw"], img, res=d["x"], want_stats=True, Hd=d["Hd"])
    y16 = ops.fdsa_fused_tail(d["x"], d["stats"], d["wpk"], d["dw"], d["fw"], img, res=d["x"], want_stats=True, Hd=d["Hd"], h_dtype=torch.bfloat16)
    torch.cuda.synchronize()
    assert torch.equal(y32, y16) and y16._fdn_pin.dtype == torch.bfloat16
    assert torch.equal(y16._fdn_pin, y32._fdn_pin.to(torch.bfloat16))


def test_ring_is_shared_and_left_clean(ops):
    """One ring serves every launch of the stream: different shapes and widths in turn, twice each - same results every time - and every launch hands all
    its blocks back (the flag words at the head of the ring are zero again)."""
    cases = [_case(ops, 2, 32, 64, 96, 1), _case(ops, 1, 64, 32, 64, 2), _case(ops, 4, 24, 16, 48, 3), _case(ops, 1, 32, 736, 1280, 4)]
    first = [_one(ops, d, pin=d["C"] <= 32) for d in cases]
    torch.cuda.synchronize()
    for _ in range(2):
        for d, (y0, s0, h0) in zip(cases, first):
            y, s, h = _one(ops, d, pin=d["C"] <= 32)
            assert torch.equal(y, y0) and torch.equal(s, s0) and (h0 is None or torch.equal(h, h0))
    torch.cuda.synchronize()
    assert ops._fdsa_scratch, "no ring was allocated"
    for scr in ops._fdsa_scratch.values():
        assert int(scr[:4096].view(torch.int32).abs().sum()) == 0, "a workgroup did not give its block back"


def test_graph_replay_of_the_one_launch_route(ops):
    d = _case(ops, 2, 32, 48, 80, 11)
    img = ops.fdsa_tail_pack(d["wp"], d["g3"], d["b3"], 32, pin=ops.fold_ln(d["wi"], None, d["g2"], d["b2"]))
    run = lambda: ops.fdsa_fused_tail(d["x"], d["stats"], d["wpk"], d["dw"], d["fw"], img, res=d["x"], want_stats=True, Hd=d["Hd"])
    ref = run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        out = run()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref) and torch.equal(out._fdn_pin, ref._fdn_pin) and torch.equal(out._fdn_stats, ref._fdn_stats)


def test_modes_without_a_form_are_refused_not_approximated(ops):
    """fdn_set_matrix_pipe(2) keeps the level-2 tail on its fp32-MFMA form (fdn_fdsa_out): the one-launch route returns FDN_ERR_UNSUPPORTED there (the
    mirror takes the pair); mode 1 (no bf16 MFMA at all) refuses every width; a width without a form has no operand image."""
    import fdn_hip
    d64, d32 = _case(ops, 1, 64, 16, 32, 5), _case(ops, 1, 32, 16, 32, 6)
    img64, img32 = ops.fdsa_tail_pack(d64["wp"], d64["g3"], d64["b3"], 64), ops.fdsa_tail_pack(d32["wp"], d32["g3"], d32["b3"], 32)
    try:
        fdn_hip.set_matrix_pipe("bf16-narrow")
        assert ops.fdsa_fused_tail(d64["x"], d64["stats"], d64["wpk"], d64["dw"], d64["fw"], img64, res=d64["x"]) is None
        assert ops.fdsa_fused_tail(d32["x"], d32["stats"], d32["wpk"], d32["dw"], d32["fw"], img32, res=d32["x"]) is not None
        fdn_hip.set_matrix_pipe("f32")
        assert ops.fdsa_fused_tail(d32["x"], d32["stats"], d32["wpk"], d32["dw"], d32["fw"], img32, res=d32["x"]) is None
    finally:
        fdn_hip.set_matrix_pipe("bf16")
    g = torch.Generator(device="cuda:0").manual_seed(1)
    assert ops.fdsa_tail_pack(torch.randn(128, 459, device="cuda:0", generator=g), torch.ones(459, device="cuda:0"), torch.zeros(459, device="cuda:0"), 128) is None
