"""GPU tests of the column FFT kernels with a compile-time plan (H = 23 * {32,16,8}: the 720p pyramid, 17 * {32,16,8}: 1080p
levels 2-3, 34 * 32 = 1088: 1080p level 1) against torch.fft in float64, in all three modes, with partial column tiles, and of the sin / cos the modulation
evaluates (large arguments take the table-driven reduction and, inside the FCAFFN kernel, the cold second pass)."""
import math

import numpy as np
import pytest
import torch

from common import rel_rms

pytestmark = pytest.mark.gpu

PLANNED_H = [736, 368, 184, 544, 272, 136, 1088, 640, 320, 160, 416, 208, 104]       # + the LOL-Blur (640 x 1120) and padded LOL-v1 (416 x 608) pyramids


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    import fdn_hip
    fdn_hip.lib()
    from fdn_hip import ops as o
    return o


def dev(t):
    return t.to("cuda:0").contiguous()


def _rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def _rd(v):
    return torch.where((v < 1e-10) & (v > -1e-10), torch.full_like(v, 1e-10), v)


@pytest.mark.parametrize("H", PLANNED_H)
@pytest.mark.parametrize("Wf", [641, 33, 7])
def test_cols_fwd_matches_fft(ops, H, Wf):
    z = _rnd(2, 3, H, Wf, 2, seed=H + Wf)
    mag, ang = ops.fft_cols_fwd(dev(z), True, True, rd_before=False, fix_real=False)
    ref = torch.fft.fft(torch.view_as_complex(z.double()), dim=2)
    assert rel_rms(mag.cpu(), ref.abs()) < 2e-6
    got = torch.polar(mag.cpu().double(), ang.cpu().double())
    assert rel_rms(torch.view_as_real(got), torch.view_as_real(ref)) < 3e-6
    only_abs, none = ops.fft_cols_fwd(dev(z), True, False, rd_before=False, fix_real=False)
    assert none is None and torch.equal(only_abs, mag)


def test_cols_fwd_fix_real_and_denormals(ops):
    """fix_real forces Im = +0 at the four self-conjugate bins (angle +pi for negative real bins); rd_before replaces
    (-1e-10, 1e-10) components by 1e-10 before abs / angle (FDN_arch.py:548-553)."""
    H, W = 368, 64
    x = -torch.ones(1, 1, H, W)                       # DC bin = -H*W: angle must be +pi, not -pi
    z = ops.rfft_rows(dev(x))
    mag, ang = ops.fft_cols_fwd(z, True, True, rd_before=False, fix_real=True)
    assert abs(ang[0, 0, 0, 0].item() - math.pi) < 1e-6
    zero = torch.zeros(1, 1, H, W // 2 + 1, 2)
    mag, ang = ops.fft_cols_fwd(dev(zero), True, True, rd_before=True, fix_real=False)
    assert torch.allclose(mag.cpu(), torch.full_like(mag.cpu(), 2 ** 0.5 * 1e-10), rtol=1e-6)
    assert torch.allclose(ang.cpu(), torch.full_like(ang.cpu(), math.pi / 4), rtol=1e-6)


@pytest.mark.parametrize("H", PLANNED_H)
def test_cols_inv_polar_matches_ifft(ops, H):
    Wf, Hin, Wfin = 41, H + 2, 45                    # leading (H, Wf) slice of wider planes, as fourier_fuse crops them
    mag = _rnd(2, 2, Hin, Wfin, seed=H).abs() + 0.1
    pha = (torch.rand(2, 2, Hin, Wfin, generator=torch.Generator().manual_seed(H + 1)) * 2 - 1) * math.pi
    z = ops.fft_cols_inv_polar(dev(mag), dev(pha), H, Wf)
    spec = torch.polar(mag[:, :, :H, :Wf].double(), pha[:, :, :H, :Wf].double())
    ref = torch.fft.ifft(spec, dim=2) * H             # the column pass is unnormalised; irfft_rows carries the scale
    assert rel_rms(z.cpu(), torch.view_as_real(ref)) < 3e-6


def _fcaffn_ref(z, amp, pha, wxa, wxp):
    """float64 restatement of FDN_arch.py:411-418 for the column pass (forward FFT over H, modulation, unnormalised inverse);
    the phase is formed in float32 like the kernel does, so that large phases compare bin for bin."""
    Z = torch.fft.fft(torch.view_as_complex(z.double()), dim=2)
    Zr = torch.complex(_rd(Z.real.float()).double(), _rd(Z.imag.float()).double())
    A = torch.einsum("ci,bihw->bchw", wxa.double(), amp.double())
    ph = torch.einsum("ci,bihw->bchw", wxp, pha).double() if wxp.abs().max() < 100 else (wxp[:, 0].view(1, -1, 1, 1) * pha[:, :1]).double()
    out = Zr * A * torch.polar(torch.ones_like(ph), -ph)
    return torch.view_as_real(torch.fft.ifft(out, dim=2) * Z.shape[2])


@pytest.mark.parametrize("H,C,Wf", [(736, 8, 73), (368, 16, 161), (184, 8, 161), (544, 8, 20), (272, 3, 9), (136, 16, 33), (1088, 8, 41),
                                    (640, 8, 57), (320, 16, 33), (160, 8, 141), (416, 8, 305), (208, 3, 153), (104, 16, 77)])
def test_cols_fcaffn_matches_reference_math(ops, H, C, Wf):
    B = 2
    z = _rnd(B, C, H, Wf, 2, seed=H)
    amp = _rnd(B, 3, H, Wf, seed=H + 1).abs()
    pha = (torch.rand(B, 3, H, Wf, generator=torch.Generator().manual_seed(H + 2)) * 2 - 1) * math.pi
    wxa, wxp = _rnd(C, 3, seed=H + 3), _rnd(C, 3, seed=H + 4)
    got = ops.fft_cols_fcaffn(dev(z).clone(), dev(amp), dev(pha), dev(wxa), dev(wxp))
    ref = _fcaffn_ref(z, amp, pha, wxa, wxp)
    assert rel_rms(got.cpu(), ref) < 5e-6


def test_cols_fcaffn_large_phases_take_the_full_range_path(ops):
    """|phase| >= 8192 in some bins of some threads: the kernel's first pass flags it and the thread redoes its bins with the
    full-range sin / cos.  wxp = (w, 0, 0) keeps the float32 phase a single exact product, so the reference sees the same angles."""
    B, C, H, Wf = 1, 8, 736, 24
    z = _rnd(B, C, H, Wf, 2, seed=5)
    amp = _rnd(B, 3, H, Wf, seed=6).abs()
    pha = (torch.rand(B, 3, H, Wf, generator=torch.Generator().manual_seed(7)) * 2 - 1) * math.pi
    wxa = _rnd(C, 3, seed=8)
    wxp = torch.zeros(C, 3)
    wxp[:, 0] = torch.tensor([1.0, 3000.0, 2.6e3, 1e5, 4e9, -7e3, 2.0, 1e20])     # channels 0 and 6 stay on the fast path
    got = ops.fft_cols_fcaffn(dev(z).clone(), dev(amp), dev(pha), dev(wxa), dev(wxp))
    ref = _fcaffn_ref(z, amp, pha, wxa, wxp)
    for ch in range(C):
        assert rel_rms(got[:, ch].cpu(), ref[:, ch]) < 5e-6, ch


def test_sincos_whole_float_range(ops):
    g = torch.Generator().manual_seed(11)
    mags = torch.cat([torch.rand(200000, generator=g) * 16000.0,                                   # both sides of the 8192 switch
                      torch.exp(torch.rand(200000, generator=g) * (88.0 - 9.0) + 9.0),             # 8e3 .. 1.6e38, log-uniform
                      torch.tensor([8191.9995, 8192.0, 8192.001, 3.4028235e38, 1e-30, 0.0, 2.0 ** 40, 2.0 ** 100])])
    x = torch.cat([mags, -mags]).float()
    sn, cs = ops.sincos(dev(x))
    xd = x.double().numpy()
    assert np.abs(sn.cpu().double().numpy() - np.sin(xd)).max() < 2e-7
    assert np.abs(cs.cpu().double().numpy() - np.cos(xd)).max() < 2e-7
    bad = torch.tensor([float("inf"), float("-inf"), float("nan")])
    sn, cs = ops.sincos(dev(bad))
    assert torch.isnan(sn).all() and torch.isnan(cs).all()


PLANNED_W = [1280, 640, 320, 1920, 960, 480, 608, 304, 1120, 560, 280]      # (35 x 16 / 8 / 4: 7 / 7 / 6 row groups per workgroup, a partly filled first stage)


@pytest.mark.parametrize("W", PLANNED_W)
@pytest.mark.parametrize("rows", [37, 64, 3, 113])
def test_rfft_rows_planned(ops, W, rows):
    """Row r2c with a compile-time plan (half-length 20|30 x 32|16|8), incl. a last workgroup with fewer rows than it holds."""
    x = _rnd(1, 1, rows, W, seed=W + rows)
    z = ops.rfft_rows(dev(x))
    ref = torch.view_as_real(torch.fft.rfft(x.double(), dim=-1))
    assert rel_rms(z.cpu(), ref) < 2e-6
    assert z[..., 0, 1].abs().max().item() == 0.0 and z[..., W // 2, 1].abs().max().item() == 0.0     # DC / Nyquist exactly real


@pytest.mark.parametrize("W", PLANNED_W)
def test_irfft_rows_planned(ops, W):
    """Row c2r: plain, with the residual epilogue, and on the leading (H, W/2+1) slice of wider / taller spectra."""
    H, Hin, Wf = 21, 23, W // 2 + 1
    Wfin = Wf + 3
    z = _rnd(2, 2, Hin, Wfin, 2, seed=W)
    res = _rnd(2, 2, H, W, seed=W + 1)
    scale = 2.0 / (H * W)
    zc = torch.view_as_complex(z.double())[:, :, :H, :Wf]
    ref = torch.fft.irfft(zc, n=W, dim=-1) * (W / 2) * scale
    got = ops.irfft_rows(dev(z), H, W, scale)
    assert rel_rms(got.cpu(), ref) < 3e-6
    got = ops.irfft_rows(dev(z), H, W, scale, res=dev(res), alpha=0.75)
    assert rel_rms(got.cpu(), ref + 0.75 * res.double()) < 3e-6


def test_rows_round_trip_at_bench_shape(ops):
    x = _rnd(1, 4, 736, 1280, seed=3)
    z = ops.rfft_rows(dev(x))
    back = ops.irfft_rows(z, 736, 1280, 2.0 / 1280)
    assert rel_rms(back.cpu(), x) < 2e-6


@pytest.mark.parametrize("C,H,W", [(32, 9, 1280), (64, 5, 640), (8, 11, 320), (3, 4, 960), (24, 6, 608), (48, 5, 304), (32, 9, 1120), (64, 13, 560), (8, 61, 280)])
def test_rfft_rows_ln_equals_layernorm_then_rfft(ops, C, H, W):
    """Row r2c with the channel LayerNorm applied on load against fdn_layernorm_chan -> fdn_rfft_rows and against float64."""
    B = 2
    x = _rnd(B, C, H, W, seed=C + W) * 1.7 + 0.4
    g, b = _rnd(C, seed=1) * 0.3 + 1.0, _rnd(C, seed=2) * 0.2
    xd = dev(x)
    got = ops.rfft_rows_ln(xd, ops.chan_stats(xd), dev(g), dev(b))
    two = ops.rfft_rows(ops.layernorm_chan(xd, dev(g), dev(b)))
    assert rel_rms(got.cpu(), two.cpu()) < 1e-6
    x64 = x.double()
    xn = (x64 - x64.mean(1, keepdim=True)) / torch.sqrt(x64.var(1, unbiased=False, keepdim=True) + 1e-5)
    xn = xn * g.double().view(1, -1, 1, 1) + b.double().view(1, -1, 1, 1)
    assert rel_rms(got.cpu(), torch.view_as_real(torch.fft.rfft(xn, dim=-1))) < 3e-6


@pytest.mark.parametrize("W,rows", [(1280, 19), (640, 5), (26, 7), (1282, 3)])
def test_rfft_rows_padded_pitch(ops, W, rows):
    """Rows written with a pitch of whole 128-byte lines (planned and generic kernels): the bins are the dense result, the padding zeros."""
    x = _rnd(1, 1, rows, W, seed=W)
    Wf = W // 2 + 1
    pitch = ops.spec_pitch(Wf)
    z = ops.rfft_rows(dev(x), pitch=pitch)
    assert z.shape[-2] == pitch
    dense = ops.rfft_rows(dev(x))
    assert torch.equal(z[..., :Wf, :], dense)
    assert z[..., Wf:, :].abs().max().item() == 0.0 if pitch > Wf else True


def test_fcaffn_chain_with_padded_spectrum(ops):
    """rfft (padded rows) -> column pass on the padded width -> irfft with the pitch as row stride equals the dense chain."""
    B, C, H, W = 1, 8, 184, 320
    x = _rnd(B, C, H, W, seed=9)
    Wf = W // 2 + 1
    amp = _rnd(B, 3, H, Wf, seed=10).abs()
    pha = (torch.rand(B, 3, H, Wf, generator=torch.Generator().manual_seed(11)) * 2 - 1) * math.pi
    wxa, wxp = _rnd(C, 3, seed=12), _rnd(C, 3, seed=13)
    outs = []
    for pitch in (None, ops.spec_pitch(Wf)):
        z = ops.rfft_rows(dev(x), pitch=pitch)
        ops.fft_cols_fcaffn(z, dev(amp), dev(pha), dev(wxa), dev(wxp))
        outs.append(ops.irfft_rows(z, H, W, 2.0 / (H * W)))
    assert rel_rms(outs[1].cpu(), outs[0].cpu()) < 1e-6
