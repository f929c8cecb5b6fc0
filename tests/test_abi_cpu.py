"""CPU checks of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol
include/mulut.h declares, fails loudly (no CPU fallback), and the host-side helpers mirror the
reference's conventions.  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from mulut_amd import _native, lut_io
from mulut_amd.options import TestOptions as _TestOptions


def declared_functions():
    txt = open(os.path.join(ROOT, "include", "mulut.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mulut_[a-z_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = _native.load()
    names = declared_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_native.EXPORTS) == names
    assert lib.mulut_version() == 100


def test_no_cpu_fallback():
    """Without a GPU the boundary refuses to create a context (and the Python engine raises)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _native.load()
    h = ctypes.c_void_p()
    rc = lib.mulut_create(0, ctypes.byref(h))
    assert rc == -7 and not h.value
    assert b"no CPU path" in lib.mulut_strerror(rc)
    from mulut_amd import MuLUTEngine, MuLUTError
    with pytest.raises(MuLUTError):
        MuLUTEngine(0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "mulut_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, fn)).read()
                assert "oracle" not in src.replace("# oracle", ""), fn


def test_strerror_codes():
    lib = _native.load()
    assert lib.mulut_strerror(0) == b"ok"
    assert lib.mulut_strerror(-2) == b"Mode not implemented."
    for code in range(-9, 0):
        assert lib.mulut_strerror(code) != b"unknown error"


def test_lut_file_naming_and_loading():
    # reader-side naming, sr/4_test_lut.py:331-332
    assert lut_io.lut_file_name("LUT_ft", 4, 4, 2, "d") == "LUT_ft_x4_4bit_int8_s2_d.npy"
    d = lut_io.load_lut_dict(os.path.join(GOLDEN, "luts"), 2, "sdy", 4, 4, "LUT_ft")
    assert sorted(d) == ["s1_d", "s1_s", "s1_y", "s2_d", "s2_s", "s2_y"]
    assert d["s1_s"].shape == (83521, 1) and d["s2_y"].shape == (83521, 16) and d["s2_y"].dtype == np.int8
    with pytest.raises(FileNotFoundError):
        lut_io.load_lut_dict(os.path.join(GOLDEN, "luts"), 3, "sdy", 4, 4, "LUT_ft")
    with pytest.raises(ValueError):       # config-1 trap: s1_s is (83521,1), not (83521,16)
        lut_io.load_lut_dict(os.path.join(GOLDEN, "luts"), 1, "s", 4, 4, "LUT_ft")


def test_cli_flags_match_reference(tmp_path):
    opt = _TestOptions().parse(["--stages", "2", "--modes", "sdy", "-e", str(tmp_path / "exp"), "--testDir", "x",
                               "--resultRoot", "y", "--lutName", "LUT_ft", "-r", "4", "--interval", "4"])
    assert (opt.stages, opt.modes, opt.scale, opt.interval, opt.lutName) == (2, "sdy", 4, 4, "LUT_ft")
    assert opt.testDir == "x" and opt.resultRoot == "y" and os.path.isdir(opt.expDir)
    assert not os.path.exists(os.path.join(opt.expDir, "code"))     # documented deviation: no save_code()
    dflt = _TestOptions().initialize(__import__("argparse").ArgumentParser()).parse_args([])
    assert (dflt.stages, dflt.modes, dflt.scale, dflt.interval) == (2, "sdy", 4, 4)
    assert dflt.testDir == "../data/SRBenchmark" and dflt.resultRoot == "../results" and dflt.lutName == "LUT_ft"


def test_metrics_on_reference_outputs():
    """PSNR/SSIM restatement reproduces the reference's published Set5 number (30.61 dB / 0.8655)."""
    from PIL import Image
    from mulut_amd.metrics import modcrop, psnr, rgb2ycbcr, ssim
    ps, ss = [], []
    for fn in sorted(os.listdir(os.path.join(GOLDEN, "Set5", "HR"))):
        gt = modcrop(np.array(Image.open(os.path.join(GOLDEN, "Set5", "HR", fn))), 4)
        out = np.array(Image.open(os.path.join(GOLDEN, "Set5", "ref_out", fn[:-4] + "_LUT_ft_4bit.png")))
        y0, y1 = rgb2ycbcr(gt)[:, :, 0], rgb2ycbcr(out)[:, :, 0]
        ps.append(psnr(y0, y1, 4))
        ss.append(ssim(y0, y1))
    assert "%.2f" % np.mean(ps) == "30.61"
    assert abs(np.mean(ss) - 0.8655) < 2e-4


def test_lut_inspector():
    recs = lut_io.inspect_lut_dir(os.path.join(GOLDEN, "luts"))
    assert len(recs) == 6 and all(r["rows_ok"] and r["int8_valued"] for r in recs)
    by = {(r["stage"], r["mode"]): r for r in recs}
    assert by[(2, "d")]["v_num"] == 16 and by[(2, "d")]["upscale"] == 4 and by[(2, "d")]["min"] == -120
    assert by[(1, "s")]["v_num"] == 1 and by[(1, "s")]["lut_name"] == "LUT_ft" and by[(1, "s")]["bits"] == 4


def test_tube2_private_registers_stay_private(tmp_path):
    """stage_tube2_kernel keeps the rows of the pass in flight in v88..v127 across asm statements (tools/gen_tube2_asm.py): the
    compiler must never allocate them.  _native.build() refuses to install a library whose assembly fails this audit; here the
    audit runs on the current sources, and on a doctored listing to show that it bites."""
    import shutil
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc")
    assert _native.audit_tube2_isa() == 3
    bad = tmp_path / "bad.s"
    bad.write_text("\n".join("_ZN5mulut18stage_tube2_kernelILi%dELi147EEEvNS_9StageArgsENS_8BandArgsE:\n\tv_mov_b32 v%d, v1\n\ts_endpgm" % (k, 90 if k == 2 else 3)
                             for k in range(3)))
    with pytest.raises(RuntimeError, match="tube2 audit failed"):
        _native.audit_tube2_isa(str(bad))


def test_tube2_blocks_match_their_generator(tmp_path):
    """mulut_tube2_asm.inc is generated (tools/gen_tube2_asm.py) and committed: the two must not drift apart."""
    import subprocess
    import sys
    out = str(tmp_path / "gen.inc")
    env = {k: v for k, v in os.environ.items() if not k.startswith("TUBE2_")}
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_tube2_asm.py"), out], env=env, stdout=subprocess.DEVNULL)
    assert open(out).read() == open(os.path.join(ROOT, "mulut_amd", "csrc", "mulut_tube2_asm.inc")).read()
