"""Pins the CPU oracle (oracle/) to outputs of the reference itself (tests/golden/gen_golden.py).
CPU only. Everything here is the checker checking itself -- no product code involved."""
import os
import re

import numpy as np
import pytest
from PIL import Image

from conftest import GOLDEN
from oracle import c_oracle, np_port


def synthetic_lut(seed, vnum):
    rng = np.random.default_rng(seed)
    return rng.integers(-127, 128, size=(17 ** 4, vnum), dtype=np.int8)


def _pass_keys(fx):
    return sorted(k for k in fx.files if k.startswith("out/"))


def test_c_oracle_pass_matches_reference(pass_fx, shipped_luts):
    n = 0
    for key in _pass_keys(pass_fx):
        _, name, u, mode, r = key.split("/")
        u, r = int(u[1:]), int(r[1:])
        img = pass_fx["in/" + name].transpose(2, 0, 1)
        lut = shipped_luts["s%d_%s" % (2 if u == 4 else 1, mode)]
        got = c_oracle.pass_q(lut, img, r, u, mode)
        assert np.array_equal(got, pass_fx[key]), key
        n += 1
    assert n == 7 * 2 * 3 * 4


def test_np_port_pass_matches_reference(pass_fx, shipped_luts):
    for key in _pass_keys(pass_fx):
        _, name, u, mode, r = key.split("/")
        if name not in ("rand_19x13x3", "ties_9x11x3", "extreme_12x10x3", "one_1x1x1"):
            continue
        u, r = int(u[1:]), int(r[1:])
        img = pass_fx["in/" + name].astype(np.float32)
        lut = shipped_luts["s%d_%s" % (2 if u == 4 else 1, mode)].astype(np.float32)
        p = np_port.PAD[mode]
        rimg = np.rot90(img, r)
        h, w, _ = rimg.shape
        img_in = np.pad(rimg, ((0, p), (0, p), (0, 0)), mode="edge").transpose(2, 0, 1)
        out = np_port.four_simplex_interp(lut, img_in, h, w, 4, 4 - r, upscale=u, mode=mode)
        assert out.dtype == np.float64
        assert np.array_equal(out * 16, pass_fx[key].astype(np.float64)), key


def test_bad_mode_raises():
    lut = synthetic_lut(0, 1)
    with pytest.raises(ValueError):
        c_oracle.pass_q(lut, np.zeros((1, 3, 3), np.uint8), 0, 1, "x")
    with pytest.raises(ValueError):
        np_port.four_simplex_interp(lut.astype(np.float32), np.zeros((1, 4, 4), np.float32), 3, 3, 4, 0, 1, "x")


def test_pipeline_2stage_sdy_matches_reference(pipe_fx, shipped_luts):
    names = sorted({k.split("/")[1] for k in pipe_fx.files if k.startswith("s2sdy/")})
    assert len(names) >= 9
    l1 = [shipped_luts["s1_" + m] for m in "sdy"]
    for name in names:
        img = pipe_fx["in/" + name]
        st1 = c_oracle.stage(l1, "sdy", False, img, 1)
        assert np.array_equal(st1, pipe_fx["s2sdy/%s/stage1" % name]), name
        fin = c_oracle.pipeline(shipped_luts, 2, "sdy", 4, img)
        assert np.array_equal(fin, pipe_fx["s2sdy/%s/final" % name]), name
        if img.size <= 19 * 13 * 3:
            f32 = {k: v.astype(np.float32) for k, v in shipped_luts.items()}
            outs = np_port.run_stages(f32, 2, "sdy", 4, img, return_all=True)
            assert np.array_equal(outs[0], pipe_fx["s2sdy/%s/stage1" % name])
            assert np.array_equal(outs[1], pipe_fx["s2sdy/%s/final" % name])


def test_pipeline_other_configs_match_reference(pipe_fx):
    keys = sorted({"/".join(k.split("/")[:2]) for k in pipe_fx.files if k.startswith("synth_")})
    assert keys
    for key in keys:
        cfg, name = key.split("/")
        m = re.match(r"synth_S(\d)_([sdy]+)_x(\d)", cfg)
        stages, modes, scale = int(m.group(1)), m.group(2), int(m.group(3))
        luts = {}
        for s in range(stages):
            last = (s + 1) == stages
            for mode in modes:
                seed = 1000 * stages + 100 * scale + 10 * s + "sdy".index(mode)
                luts["s%d_%s" % (s + 1, mode)] = synthetic_lut(seed, scale * scale if last else 1)
        img = pipe_fx["in/" + name]
        fin = c_oracle.pipeline(luts, stages, modes, scale, img)
        assert np.array_equal(fin, pipe_fx["%s/stage%d" % (key, stages)]), key
        cur = img
        for s in range(stages - 1):
            cur = c_oracle.stage([luts["s%d_%s" % (s + 1, mm)] for mm in modes], modes, False, cur, 1)
            assert np.array_equal(cur, pipe_fx["%s/stage%d" % (key, s + 1)]), (key, s)


def test_config1_single_s_stage(pipe_fx, shipped_luts):
    for name in ("rand_19x13x3", "smooth_24x20x3"):
        fin = c_oracle.pipeline({"s1_s": shipped_luts["s2_s"]}, 1, "s", 4, pipe_fx["in/" + name])
        assert np.array_equal(fin, pipe_fx["cfg1_s2s_as_s1/%s/final" % name])


def test_set5_matches_reference_pngs(shipped_luts):
    """The reference's five committed outputs (results/sr_x2sdy/Set5/X4) are its only golden vectors."""
    lr_dir = os.path.join(GOLDEN, "Set5", "LR_bicubic", "X4")
    for fn in sorted(os.listdir(lr_dir)):
        lr = np.array(Image.open(os.path.join(lr_dir, fn)))
        want = np.array(Image.open(os.path.join(GOLDEN, "Set5", "ref_out", fn[:-4] + "_LUT_ft_4bit.png")))
        got = c_oracle.pipeline(shipped_luts, 2, "sdy", 4, lr)
        assert np.array_equal(got, want), fn
