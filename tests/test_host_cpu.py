"""CPU: host-side logic of the drop-in modules and the C-ABI library (no GPU compute)."""
import ctypes
import os
import re

import pytest
import torch

import __graft_entry__ as entry  # noqa: F401  (puts the package on sys.path)
from common import fdn_shapes, fdn_weights, lpnet_weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import fdn_hip
    if not os.path.isfile(fdn_hip.lib_path()):
        entry.build()
    return fdn_hip.lib()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "fdn_hip.h")).read()
    names = sorted(set(re.findall(r"\b(fdn_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"libfdn_hip.so does not export {n}"
    assert lib.fdn_abi_version() >= 1
    assert lib.fdn_error_string(1).decode().startswith("invalid argument")


def test_ctypes_prototypes_match_header(lib):
    """fdn_hip/_abi.py (argtypes of every entry point) is exactly what include/fdn_hip.h declares, and the loader applied it."""
    import importlib.util
    import ctypes as C
    from fdn_hip._abi import PROTOTYPES
    spec = importlib.util.spec_from_file_location("gen_abi_table", os.path.join(ROOT, "tools", "gen_abi_table.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    assert PROTOTYPES == gen.parse_header(os.path.join(ROOT, "include", "fdn_hip.h"))
    assert lib.fdn_fdsa_fused.argtypes is not None and len(lib.fdn_fdsa_fused.argtypes) == len(PROTOTYPES["fdn_fdsa_fused"][1])
    with pytest.raises(C.ArgumentError):
        lib.fdn_fdsa_core(None, None, None, None, 1.5, 38, 32, 32, None)            # a float where an int belongs
    with pytest.raises(TypeError):
        lib.fdn_fdsa_core(None, None, None, None, 1, 38, 32, 32)          # one argument short


def test_argument_validation_without_gpu(lib):
    # NULL pointers / bad sizes are rejected before any launch
    assert lib.fdn_fdsa_core(None, None, None, None, 1, 38, 32, 32, None) == 1
    assert lib.fdn_rfft_rows(None, None, ctypes.c_long(4), 16, ctypes.c_long(0), None) == 1
    # (round 6) the one-launch FDSA route: sizes of its operand image / ring are host arithmetic, a width without a form has none
    assert lib.fdn_fdsa_fused_tail(None, ctypes.c_long(0), None, None, None, None, None, None, None, None, None, None, 1, 32, 38, 32, 32, 0, 0, None) == 1
    assert lib.fdn_fdsa_tail_pack_floats(32, 38, 32, 0) == 4096 and lib.fdn_fdsa_tail_pack_floats(32, 38, 32, 86) > 4096
    assert lib.fdn_fdsa_tail_pack_floats(64, 76, 64, 0) == 512 + 3 * 5 * 2 * 3 * 64 * 4 and lib.fdn_fdsa_tail_pack_floats(128, 153, 128, 0) == 0
    assert lib.fdn_fdsa_tail_pack_floats(48, 57, 48, 129) == 0                      # no project_in behind the C = 48 tail
    ring = lib.fdn_fdsa_scratch_floats(8, 38, 736, 1280)
    assert ring == lib.fdn_fdsa_scratch_floats(1, 38, 64, 64) == 16384 + 4096 * 4 * 38 * 256      # a ring: independent of the image


def test_state_dict_layout_matches_reference():
    from basicsr.models.archs.FDN_arch import FDN
    from basicsr.models.archs.LPNet_arch import I_predict_net
    net = FDN()
    sd = net.state_dict()
    ref = fdn_shapes()                      # key -> shape table captured from the reference FDN()
    assert len(sd) == 1503 and set(sd) == set(ref)
    assert all(tuple(sd[k].shape) == tuple(ref[k]) for k in ref)
    net.load_state_dict(fdn_weights(), strict=True)
    assert sum(p.numel() for p in net.parameters()) == 8030489
    assert not any(p.requires_grad for p in net.net_a.parameters())      # FDN_arch.py:858-859
    lp = I_predict_net()
    lp.load_state_dict(lpnet_weights(), strict=True)
    assert len(lp.state_dict()) == 292


def test_star_import_surface():
    ns = {}
    exec("from basicsr.models.archs.FDN_arch import *\nfrom basicsr.models.archs.LPNet_arch import *", ns)
    for name in ("FDN", "F", "torch", "nn", "np", "rearrange", "I_predict_net", "transforms"):
        assert name in ns, name
    from basicsr.models.archs import define_network
    assert type(define_network({"type": "FDN"})).__name__ == "FDN"


def test_no_cpu_fallback():
    from basicsr.models.archs.FDN_arch import FDN
    from basicsr.models.archs.LPNet_arch import I_predict_net
    import fdn_hip
    with pytest.raises(fdn_hip.FdnHipError):
        FDN().eval()(torch.rand(1, 3, 32, 32), ratio_i=torch.rand(1, 1))
    with pytest.raises(fdn_hip.FdnHipError):
        I_predict_net().eval()(torch.rand(1, 3, 32, 32))
    with pytest.raises(ValueError):
        FDN().eval()(torch.rand(1, 3, 32, 32))


def test_product_path_never_imports_oracle():
    pkg = os.path.join(ROOT, "fdn-tip2025_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".sh")):
                txt = open(os.path.join(dp, f)).read()
                assert "fdn_oracle" not in txt and "oracle/" not in txt, os.path.join(dp, f)


def test_harness_rejects_cpu_tensors():
    """The GPU pre/post stage has no host fallback either: CPU tensors are refused before any library call."""
    import torch
    import fdn_hip
    from fdn_hip import harness
    assert harness.padded_size(720, 1280) == (736, 1280) and harness.padded_size(64, 96) == (64, 96)
    with pytest.raises(fdn_hip.FdnHipError):
        harness.preprocess(torch.zeros(1, 40, 40, 3, dtype=torch.uint8))
    with pytest.raises(fdn_hip.FdnHipError):
        harness.postprocess(torch.zeros(1, 3, 64, 64), 40, 40)


def test_lolv1_state_dict_layout_matches_reference():
    """FDN_lolv1 (dim 24): same 1503 keys / shapes as the reference module, so its checkpoint loads strict."""
    from basicsr.models.archs.fdnlol24_arch import FDN_lolv1
    from common import lolv1_shapes, lolv1_weights
    net = FDN_lolv1()
    sd, ref = net.state_dict(), lolv1_shapes()
    assert len(sd) == 1503 and set(sd) == set(ref)
    assert all(tuple(sd[k].shape) == tuple(ref[k]) for k in ref)
    assert sd["net_p.patch_embed.proj.weight"].shape[0] == 24 and "net_a.net.Encoder.0.cat.weight" in sd
    net.load_state_dict(lolv1_weights(), strict=True)


def test_tiling_origins_match_oracle_and_reject_cpu():
    import torch
    import fdn_hip
    import fdn_oracle as O
    from fdn_hip import tiling
    for (h, w, ch, cw) in ((720, 1280, 256, 256), (100, 150, 64, 64), (96, 96, 96, 96), (1088, 1920, 512, 640)):
        assert tiling.tile_origins(h, w, ch, cw) == O.grids_indices(h, w, ch, cw)[2]
    with pytest.raises(fdn_hip.FdnHipError):
        tiling.split(torch.zeros(1, 3, 64, 64), 32, 32)
    with pytest.raises(fdn_hip.FdnHipError):
        tiling.tile_origins(64, 64, 128, 32)


def test_sincos_large_argument_table():
    """The large-argument reduction of fdn_sincos (csrc/common.hpp) restated with exact rationals on the committed table
    (csrc/sincos_table.inc, made by tools/gen_sincos_table.py): x * 2/pi mod 4 = m * T[e] mod 4 for |x| = m * 2^e."""
    import math
    import os
    import struct
    from fractions import Fraction

    import numpy as np

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rows = []
    for line in open(os.path.join(root, "fdn-tip2025_amd", "csrc", "sincos_table.inc")):
        if line.startswith("{"):
            hi, lo = line.strip().strip("{},").split(",")
            rows.append((float.fromhex(hi.strip()), float.fromhex(lo.strip())))
    assert len(rows) == 115
    rng = np.random.default_rng(3)
    xs = np.concatenate([np.exp(rng.uniform(9.02, 88.7, 400)), [8192.0, 2.0 ** 40, 3.4028235e38]]).astype(np.float32)
    for x in xs:
        bits = struct.unpack("<I", struct.pack("<f", float(x)))[0]
        eb, m = bits >> 23, (bits & 0x7FFFFF) | 0x800000
        assert Fraction(m) * Fraction(2) ** (eb - 150) == Fraction(float(x))
        fh, fl = rows[eb - 140]
        exact = Fraction(m) * Fraction(fh)
        p = float(exact)                                   # the rounded fp64 product
        pe = float(exact - Fraction(p))                    # fma(m, fh, -p): exact
        kq = round(p)                                      # (ties cannot occur within the tolerance below)
        fr = (p - kq) + float(Fraction(m) * Fraction(fl) + Fraction(pe))
        r, q = fr * (math.pi / 2), kq & 3
        s, c = math.sin(r), math.cos(r)
        sn = (s, c, -s, -c)[q]
        cs = (c, -s, -c, s)[q]
        assert abs(sn - math.sin(float(x))) < 1e-12 and abs(cs - math.cos(float(x))) < 1e-12, float(x)


def test_ratio_sweep_grid_and_names_follow_the_reference():
    """inference_fdn_multi_r.py:55,84: `for i in np.arange(0, 1, 0.01)` ... `"./multi_r/{}.png".format(i)` - the driver's grid and the
    file names it writes are exactly those (format of the numpy float64 loop variable)."""
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "fdn-tip2025_amd"))
    import inference_fdn_multi_r as M
    vals = M.sweep_values(0.0, 1.0, 0.01)
    ref = [i for i in np.arange(0, 1, 0.01)]
    assert len(vals) == 100 and all(float(a) == float(b) for a, b in zip(vals, ref))
    assert [M.output_name(v) for v in vals[:3]] == ["{}.png".format(i) for i in ref[:3]] == ["0.0.png", "0.01.png", "0.02.png"]
    assert M.output_name(vals[29]) == "{}.png".format(ref[29])


def test_no_waterfall_loops_in_the_patch_kernels(tmp_path):
    """The compiler wrapped every halo load of fdffn_mid's channel loop in a waterfall loop (the strength-reduced plane offset lived in a vector
    register: 12 loops, ~150 issue slots per channel).  Compile patchfft.hip to gfx950 assembly and hold the count at 0."""
    import importlib.util
    import subprocess
    root = ROOT
    out = tmp_path / "patchfft.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "--cuda-device-only", "-S",
                    os.path.join(root, "fdn-tip2025_amd", "csrc", "patchfft.hip"), "-o", str(out)], check=True, timeout=600)
    spec = importlib.util.spec_from_file_location("isa_waterfall", os.path.join(root, "tools", "isa_waterfall.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    stats = mod.waterfalls(str(out))
    kernels = [k for k in stats if "fdffn_mid_kernel" in k or "fdsa_fused_kernel" in k or "fdsa_core_kernel" in k]
    assert len(kernels) >= 10, sorted(stats)
    assert {k: stats[k][0] for k in kernels if stats[k][0]} == {}


def test_reference_driver_imports_through_the_shadow(tmp_path):
    """this package FIRST on PYTHONPATH, a reference checkout after it.  The checkout here is a stub laid out like the
    reference (basicsr/ WITHOUT an __init__.py, basicsr/utils/__init__.py with the four helpers, basicsr/models/__init__.py that must NOT run,
    a reference-only arch module); the import lines are those of /root/reference/inference_fdn_lolblur.py:1-6, executed verbatim in a fresh
    interpreter.  FDN / I_predict_net must come from this package, the helpers from the checkout."""
    import subprocess
    import sys
    ck = tmp_path / "checkout"
    (ck / "basicsr" / "utils").mkdir(parents=True)
    (ck / "basicsr" / "models" / "archs").mkdir(parents=True)
    (ck / "basicsr" / "utils" / "__init__.py").write_text(
        "def get_root_logger(*a, **k): return 'logger'\ndef imwrite(*a, **k): return 'imwrite'\n"
        "def tensor2img(*a, **k): return 'tensor2img'\ndef img2tensor(*a, **k): return 'img2tensor'\ndef scandir(*a, **k): return []\n")
    (ck / "basicsr" / "models" / "__init__.py").write_text("raise RuntimeError('the training stack of the checkout must not be imported')\n")
    (ck / "basicsr" / "models" / "archs" / "__init__.py").write_text("raise RuntimeError('the registry of the checkout must not be imported')\n")
    (ck / "basicsr" / "models" / "archs" / "FDN_arch.py").write_text("raise RuntimeError('shadowing failed: the reference FDN_arch was imported')\n")
    (ck / "basicsr" / "models" / "archs" / "mar_arch.py").write_text("MAR_ONLY_IN_CHECKOUT = 1\n")
    driver_lines = (
        "from basicsr.utils import get_root_logger, imwrite, tensor2img\n"
        "\n"
        "from basicsr.models.archs.FDN_arch import *\n"
        "\n"
        "from basicsr.models.archs.LPNet_arch import *\n"
        "from basicsr.utils import img2tensor\n")
    check = (
        "import basicsr, os\n"
        "pkg = os.environ['FDN_PKG']\n"
        "assert FDN.__module__ == 'basicsr.models.archs.FDN_arch' and os.path.realpath(sys.modules[FDN.__module__].__file__).startswith(pkg), sys.modules[FDN.__module__].__file__\n"
        "assert os.path.realpath(sys.modules[I_predict_net.__module__].__file__).startswith(pkg)\n"
        "assert transforms is not None and F is torch.nn.functional and rearrange is not None and np is not None and nn is torch.nn\n"
        "assert get_root_logger() == 'logger' and imwrite() == 'imwrite' and tensor2img() == 'tensor2img' and img2tensor() == 'img2tensor'\n"
        "from basicsr.models.archs.fdnlol24_arch import FDN_lolv1\n"
        "from basicsr.models.archs import mar_arch, define_network\n"
        "assert mar_arch.MAR_ONLY_IN_CHECKOUT == 1\n"
        "assert type(define_network({'type': 'FDN'})).__name__ == 'FDN'\n"
        "print('shadow ok')\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([entry.PKG, str(ck)]), FDN_PKG=os.path.realpath(entry.PKG))
    r = subprocess.run([sys.executable, "-c", "import sys\n" + driver_lines + check], env=env, capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0 and "shadow ok" in r.stdout, r.stderr[-2000:]


def test_reference_model_registry_is_reached_on_demand(tmp_path):
    """`from basicsr.models import create_model` (the reference's test.py / train.py) - the shadow package runs the checkout's
    basicsr/models/__init__.py the first time one of its names is asked for, and says so when that fails or when no checkout follows."""
    import subprocess
    import sys
    ck = tmp_path / "checkout"
    (ck / "basicsr" / "models").mkdir(parents=True)
    (ck / "basicsr" / "models" / "__init__.py").write_text("def create_model(opt):\n    return ('model of the checkout', opt)\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([entry.PKG, str(ck)]))
    code = ("from basicsr.models import create_model\nassert create_model(3) == ('model of the checkout', 3)\n"
            "import basicsr.models as m\n"
            "try:\n    m.no_such_name\nexcept AttributeError as e:\n    assert 'drop-in' in str(e)\nelse:\n    raise SystemExit('no error')\nprint('registry ok')\n")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0 and "registry ok" in r.stdout, r.stderr[-2000:]
    (ck / "basicsr" / "models" / "__init__.py").write_text("import a_module_the_training_stack_needs\n")
    code = ("try:\n    from basicsr.models import create_model\nexcept ImportError as e:\n"
            "    assert 'on demand' in str(e) and 'a_module_the_training_stack_needs' in str(e), str(e)\n    print('clear error')\n")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0 and "clear error" in r.stdout, r.stderr[-2000:]
    code = ("import basicsr.models as m\ntry:\n    m.create_model\nexcept AttributeError as e:\n    assert 'no reference checkout' in str(e), str(e)\n    print('clear error')\n")
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PYTHONPATH=entry.PKG), capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0 and "clear error" in r.stdout, r.stderr[-2000:]


def test_bench_describes_calls_by_name():
    """bench.py's roofline figures come from the ARGUMENTS of the C-ABI calls it wraps.  They are read by NAME (fdn_hip/_abi.py ARG_NAMES,
    generated from the header), so an inserted or reordered parameter of a later ABI version cannot silently shift a shape: every parser gets
    a call built from the header's own prototype in which each integer parameter carries a distinct value, and the group key it returns must
    show the values of the parameters it names - then the same call with two parameters swapped IN THE TABLE must change the key or raise."""
    import importlib
    import bench
    from fdn_hip._abi import ARG_NAMES, PROTOTYPES
    spec = importlib.util.spec_from_file_location("gen_abi_table", os.path.join(ROOT, "tools", "gen_abi_table.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    assert ARG_NAMES == gen.parse_names(os.path.join(ROOT, "include", "fdn_hip.h"))
    assert set(ARG_NAMES) == set(PROTOTYPES) and all(len(ARG_NAMES[n]) == len(PROTOTYPES[n][1]) for n in PROTOTYPES)
    # every entry point the step's by_entry_point_ms can name either has a parser or is a helper without a shape figure; the heavy ones must have one
    heavy = ("fdn_conv1x1", "fdn_fdsa_fused", "fdn_fdsa_out", "fdn_fdsa_core", "fdn_fdffn_mid", "fdn_ffn_tail", "fdn_dwconv_gate", "fdn_conv2d",
             "fdn_fft_cols_fcaffn", "fdn_rfft_rows_ln", "fdn_irfft_rows", "fdn_rfft_rows", "fdn_fcaffn_in", "fdn_fcaffn_in_packed", "fdn_chan_stats")
    assert all(n in bench.DESCRIBERS for n in heavy)
    assert all(n in PROTOTYPES for n in bench.DESCRIBERS)
    shape_params = ("C", "E", "Hd", "G", "Cin", "Cout", "rows", "planes", "N", "H", "W", "P", "Wf")
    derived = {"fdn_rfft_rows_ln": {"C", "H"}}               # (shown as rows = B * C * H)
    for name, fn in bench.DESCRIBERS.items():
        if name == "fdn_conv1x1":
            continue                                             # one descriptor struct: fields are named by construction
        names, kinds = ARG_NAMES[name], PROTOTYPES[name][1]
        vals = {n: (None if k == "P" else 0 if n.endswith("bf16") or n in ("form", "pad") else 101 + 2 * i) for i, (n, k) in enumerate(zip(names, kinds))}
        if "stride" in vals:
            vals["stride"] = 1
        args = [ctypes.c_long(v) if k == "L" and v is not None else v for v, k in ((vals[n], k) for n, k in zip(names, kinds))]
        key, flops, byts = bench.describe_call(name, args)
        assert key.startswith(name + "[") and byts > 0, (name, key)
        for pn in shape_params:
            if pn in vals and vals[pn] is not None and pn not in derived.get(name, ()):
                assert re.search(r"(?<!\d)%d(?!\d)" % vals[pn], key), f"{name}: parameter {pn}={vals[pn]} not in key {key}"
        ints = [n for n in names if n in shape_params and n not in derived.get(name, ())]
        if len(ints) >= 2:                                       # a table in another order = a different ABI: the same positional call must not read the same
            swapped = list(names)
            i, j = names.index(ints[0]), names.index(ints[1])
            swapped[i], swapped[j] = swapped[j], swapped[i]
            ARG_NAMES[name] = swapped
            try:
                key2, _, byts2 = bench.describe_call(name, args)
            finally:
                ARG_NAMES[name] = names
            assert key2 != key, (name, key, key2)
    with pytest.raises(TypeError):
        bench.describe_call("fdn_fdsa_fused", [None] * 3)       # a call that does not match the declared parameter count is refused


def test_graph_key_covers_every_routing_switch():
    """A captured HIP graph replays the kernels of its capture: every process-wide switch that decides WHICH kernels a forward launches must be part of
    pipeline.weights_signature, or a holder would replay a stale route (ops.FFN_TAIL_MODE was missing).  Checked on CPU modules."""
    import torch
    from fdn_hip import ops, pipeline
    m = torch.nn.Linear(3, 3)
    base = pipeline.weights_signature(m)
    flips = {"FDSA_FULL": True, "FDSA_FULL_MAX_C": 64, "FDSA_TAIL": False, "FDSA_TAIL_PIN": False, "FFN_TAIL_MODE": "split", "SPECTRAL_MLP_FUSED": False,
             "GEMM_OWN_STATS": False, "UPCONV_GATHER": False, "AFF_MULTIRES": False}
    # every module-level switch of ops.py whose initial value is a bool or None (the routing switches; thresholds and width tables are constants)
    import re
    src = open(ops.__file__).read()
    switches = set(re.findall(r"^([A-Z][A-Z0-9_]*) = (?:True|False|None)\b", src, flags=re.M))
    assert switches <= set(flips), f"routing switches missing from this test (and maybe from the graph key): {sorted(switches - set(flips))}"
    for name, val in flips.items():
        old = getattr(ops, name)
        assert old != val, name
        setattr(ops, name, val)
        try:
            assert pipeline.weights_signature(m) != base, f"ops.{name} is not part of the graph key"
        finally:
            setattr(ops, name, old)
    assert pipeline.weights_signature(m) == base
    with torch.no_grad():
        m.weight.add_(1.0)                      # an in-place weight update is seen too
    assert pipeline.weights_signature(m) != base
