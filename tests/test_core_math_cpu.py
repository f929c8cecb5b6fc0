"""CPU unit test of mulut_amd/csrc/mulut_core.h -- the per-site integer math every gfx950 kernel is
built from -- compiled by g++ (tests/host_emul/emul.cpp) and checked against the oracle and the
golden fixtures.  This is a test of shared source, NOT a CPU product path."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from oracle import c_oracle

HERE = os.path.join(ROOT, "tests", "host_emul")


@pytest.fixture(scope="module")
def emul():
    so = os.path.join(HERE, "libemul.so")
    src = os.path.join(HERE, "emul.cpp")
    hdr = os.path.join(ROOT, "mulut_amd", "csrc", "mulut_core.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", so, src])
    L = ctypes.CDLL(so)
    L.emul_stage.restype = ctypes.c_int
    return L


def run_emul(L, luts, modes, is_last, img_hwc, u):
    img = np.ascontiguousarray(img_hwc.transpose(2, 0, 1))
    C, H, W = img.shape
    keep = [np.ascontiguousarray(t, dtype=np.int8) for t in luts]
    arr = (ctypes.c_void_p * len(keep))(*[t.ctypes.data for t in keep])
    out = np.empty((H * u, W * u, C), np.uint8)
    rc = L.emul_stage(arr, modes.encode(), len(modes), int(is_last), ctypes.c_void_p(img.ctypes.data), H, W, C, u,
                      ctypes.c_void_p(out.ctypes.data))
    assert rc == 0
    return out


def test_core_math_two_stage_sdy(emul, pipe_fx, shipped_luts):
    for name in sorted({k.split("/")[1] for k in pipe_fx.files if k.startswith("s2sdy/")}):
        img = pipe_fx["in/" + name]
        s1 = run_emul(emul, [shipped_luts["s1_" + m] for m in "sdy"], "sdy", False, img, 1)
        assert np.array_equal(s1, pipe_fx["s2sdy/%s/stage1" % name]), name
        fin = run_emul(emul, [shipped_luts["s2_" + m] for m in "sdy"], "sdy", True, s1, 4)
        assert np.array_equal(fin, pipe_fx["s2sdy/%s/final" % name]), name


@pytest.mark.parametrize("u", [1, 2, 3, 4])
@pytest.mark.parametrize("modes", ["s", "sd", "sdy", "ysd"])
def test_core_math_vs_oracle_random(emul, u, modes):
    rng = np.random.default_rng(u * 10 + len(modes))
    luts = [rng.integers(-128, 128, (17 ** 4, u * u), dtype=np.int8) for _ in modes]   # includes -128
    for C, H, W in ((3, 9, 14), (1, 5, 3), (2, 7, 6)):
        img = rng.integers(0, 256, (H, W, C), dtype=np.uint8)
        for last in (True, False):
            got = run_emul(emul, luts, modes, last, img, u)
            want = c_oracle.stage(luts, modes, last, img, u)
            assert np.array_equal(got, want), (u, modes, C, last)


def test_core_math_extreme_tables(emul):
    """all +127 / all -128 tables drive the SWAR fields and the clip to their limits"""
    img = np.random.default_rng(5).integers(0, 256, (6, 5, 3), dtype=np.uint8)
    for val in (127, -128, 0):
        luts = [np.full((17 ** 4, 16), val, np.int8)] * 3
        got = run_emul(emul, luts, "sdy", True, img, 4)
        want = c_oracle.stage(luts, "sdy", True, img, 4)
        assert np.array_equal(got, want)
        l1 = [np.full((17 ** 4, 1), val, np.int8)] * 3
        assert np.array_equal(run_emul(emul, l1, "sdy", False, img, 1), c_oracle.stage(l1, "sdy", False, img, 1))


@pytest.mark.parametrize("modes", ["sdysd", "sdysdysd"])
def test_core_math_many_modes_extreme_tables(emul, modes):
    """u = 4 with more than four modes: the merged rotation-pair accumulators would overflow their 16-bit fields
    (4 M 16 255 > 65535), so the per-rotation form is used -- all-(+127) / all-(-128) tables drive it to the limit"""
    img = np.random.default_rng(9).integers(0, 256, (5, 7, 3), dtype=np.uint8)
    rng = np.random.default_rng(len(modes))
    for val in (127, -128, None):
        luts = [np.full((17 ** 4, 16), val, np.int8) if val is not None else rng.integers(-128, 128, (17 ** 4, 16), dtype=np.int8)
                for _ in modes]
        assert np.array_equal(run_emul(emul, luts, modes, True, img, 4), c_oracle.stage(luts, modes, True, img, 4)), (modes, val)


def test_band_pair_index_math(emul):
    emul.emul_check_band_pair.restype = ctypes.c_long
    assert emul.emul_check_band_pair(5) == 0


def test_full_table_pair_math_for_one_byte_rows(emul):
    """mulut_core.h full-table pairs (first stage, detailed tiles): rows / weights == the scalar simplex"""
    emul.emul_check_full_pair1.restype = ctypes.c_long
    assert emul.emul_check_full_pair1(5) == 0


def test_slab_pair_index_math_and_raw_byte_accumulation(emul):
    """mulut_core.h slab pairs: rows / weights of the packed pair math == the scalar simplex for every key combination
    (sampled), and F / H raw-byte accumulation == the 16-bit field sums up to the 4-mode bound"""
    emul.emul_check_slab_pair.restype = ctypes.c_long
    assert emul.emul_check_slab_pair(5) == 0


def test_tube_pair_index_math(emul):
    """tube band of the final-stage kernel: injective slot map on its 991 rows, bank property, packed pair math"""
    emul.emul_check_tube_pair.restype = ctypes.c_long
    assert emul.emul_check_tube_pair(5) == 0


def test_tube_pair1_index_math_for_every_slot_size(emul):
    """tube pairs of the 1-byte-row kernel family with 4-, 8- and 24-byte slots (u = 1, 2, 3): byte offsets and weights of the packed pair
    math == the scalar simplex for every in-tube key combination (sampled); field layout of the u == 3 band"""
    emul.emul_check_tube_pair1.restype = ctypes.c_long
    assert emul.emul_check_tube_pair1(5) == 0


def test_float_epilogue_validity_table():
    """rhe_f32_valid() outcome per mode count (documented in DESIGN.md): the GPU uses the float epilogue
    only where it is proven exact, so this is informational -- but M = 3 (sdy) must be on the fast path."""
    src = open(os.path.join(ROOT, "mulut_amd", "csrc", "mulut_core.h")).read()
    assert "rhe_f32_valid" in src
