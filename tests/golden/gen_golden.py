#!/usr/bin/env python3
"""Generate the committed golden fixtures by RUNNING THE REFERENCE in this container.

This script is the only place in the repo that touches /root/reference. It imports the
reference's own `sr/4_test_lut.py` (per-pass function `FourSimplexInterpFaster`, :14-237)
and `sr/5_test_lut.py` (the fork's working driver `process_single_image`, :241-323) and
records inputs + outputs as small .npz files next to this script. It refuses to run when
/root/reference is absent, so it is inert on the GPU box. Nothing of the reference's source
is copied: only data (inputs, outputs, the shipped .npy LUTs and Set5 PNGs) is written.

    python tests/golden/gen_golden.py            # rewrites tests/golden/*.npz + data dirs

cv2 is not installed here; `common/utils.py:3` imports it at module top only for
`cv2.getGaussianKernel` (SSIM), so a 5-line stub module is inserted.
"""
import importlib.util
import os
import shutil
import sys
import tempfile
import types
from types import SimpleNamespace

import numpy as np
from PIL import Image

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _load_reference():
    if not os.path.isdir(REF):
        raise SystemExit("gen_golden.py: /root/reference not present - fixtures are generated "
                         "in the authoring container only")
    cv2 = types.ModuleType("cv2")

    def getGaussianKernel(ksize, sigma):
        i = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2.0
        k = np.exp(-(i * i) / (2.0 * sigma * sigma))
        return (k / k.sum()).reshape(-1, 1)

    cv2.getGaussianKernel = getGaussianKernel
    sys.modules.setdefault("cv2", cv2)
    sys.path.insert(0, REF)
    mods = []
    for name, fn in (("ref_test4", "sr/4_test_lut.py"), ("ref_test5", "sr/5_test_lut.py")):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, fn))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        mods.append(m)
    return mods


def synthetic_lut(seed, vnum):
    """Seeded synthetic int8 table; the same function lives in mulut_amd/lut_io.py."""
    rng = np.random.default_rng(seed)
    return rng.integers(-127, 128, size=(17 ** 4, vnum), dtype=np.int8)


def make_inputs():
    """Small HWC uint8 images: random, LSB ties, extremes, smooth."""
    rng = np.random.default_rng(1234)
    cases = {}
    cases["rand_19x13x3"] = rng.integers(0, 256, (19, 13, 3), dtype=np.uint8)
    cases["rand_8x31x1"] = rng.integers(0, 256, (8, 31, 1), dtype=np.uint8)
    cases["rand_5x4x2"] = rng.integers(0, 256, (5, 4, 2), dtype=np.uint8)
    cases["rand_2x7x1"] = rng.integers(0, 256, (2, 7, 1), dtype=np.uint8)
    cases["one_1x1x1"] = np.array([[[200]]], dtype=np.uint8)
    # LSB ties: all pixels share the same low nibble -> every simplex comparison is a tie
    t = (rng.integers(0, 16, (9, 11, 3)) * 16 + 7).astype(np.uint8)
    cases["ties_9x11x3"] = t
    # extremes: only 0, 15, 16, 239, 240, 255 (MSB 15 -> corner index 16 is exercised)
    ext = np.array([0, 15, 16, 239, 240, 255], dtype=np.uint8)
    cases["extreme_12x10x3"] = ext[rng.integers(0, len(ext), (12, 10, 3))]
    cases["const255_6x6x1"] = np.full((6, 6, 1), 255, dtype=np.uint8)
    yy, xx = np.mgrid[0:24, 0:20]
    sm = np.stack([(yy * 9 + xx * 3) % 256, (yy * 2 + xx * 11 + 40) % 256, (255 - yy * 7 - xx) % 256], -1)
    cases["smooth_24x20x3"] = sm.astype(np.uint8)
    return cases


def ref_pass(t4, lut_f32, img_hwc_u8, rot_r, upscale, mode):
    """One reference pass exactly as the driver calls it (sr/4_test_lut.py:289-298)."""
    img = img_hwc_u8.astype(np.float32)
    pad = (0, 2) if mode in "dy" else (0, 1)
    rimg = np.rot90(img, rot_r)
    h, w, _ = rimg.shape
    img_in = np.pad(rimg, (pad, pad, (0, 0)), mode="edge").transpose((2, 0, 1))
    out = t4.FourSimplexInterpFaster(lut_f32, img_in, h, w, 4, 4 - rot_r, upscale=upscale, mode=mode)
    k = np.asarray(out) * 16.0
    ki = np.rint(k).astype(np.int32)
    assert np.array_equal(ki.astype(np.float64), k), "reference pass is not a multiple of 1/16"
    return ki  # (C, H*u, W*u) int32 == 16*out


def ref_stages(t4, luts, img_hwc_u8, stages, modes, scale):
    """Stage loop of sr/4_test_lut.py:279-306 around the reference's own pass function;
    returns the list of per-stage images (intermediate ones as uint8)."""
    img = img_hwc_u8.astype(np.float32)
    outs = []
    for s in range(stages):
        pred = 0
        last = (s + 1) == stages
        upscale = scale if last else 1
        avg, bias = (len(modes), 0) if last else (len(modes) * 4, 127)
        for mode in modes:
            pad = (0, 2) if mode in "dy" else (0, 1)
            for r in range(4):
                rimg = np.rot90(img, r)
                h, w, _ = rimg.shape
                img_in = np.pad(rimg, (pad, pad, (0, 0)), mode="edge").transpose((2, 0, 1))
                pred = pred + t4.FourSimplexInterpFaster(luts["s%d_%s" % (s + 1, mode)], img_in, h, w, 4, 4 - r,
                                                         upscale=upscale, mode=mode)
        img = np.round(np.clip(np.clip(pred / avg + bias, 0, 255).transpose((1, 2, 0)), 0, 255))
        img = img.astype(np.uint8) if last else img.astype(np.float32)
        outs.append(img.astype(np.uint8))
    return outs


def ref_driver(t5, luts, img_hwc_u8, stages, modes, scale):
    """End-to-end through the reference's own driver (sr/5_test_lut.py:241-323), via PNG files."""
    opt = SimpleNamespace(stages=stages, modes=list(modes), scale=scale, interval=4)
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "in.png")
        dst = os.path.join(td, "out", "o.png")
        arr = img_hwc_u8
        if arr.shape[2] == 1:
            Image.fromarray(arr[:, :, 0]).save(src)          # gray -> driver replicates to 3 channels
        else:
            Image.fromarray(arr).save(src)
        t5.process_single_image(src, dst, opt, luts)
        return np.array(Image.open(dst))


def main():
    t4, t5 = _load_reference()
    lut_dir = os.path.join(REF, "models/sr_x2sdy")
    # ---- data files (not code): shipped LUTs + Set5 ----
    out_luts = os.path.join(HERE, "luts")
    os.makedirs(out_luts, exist_ok=True)
    shipped = {}
    for s in (1, 2):
        for m in "sdy":
            fn = "LUT_ft_x4_4bit_int8_s%d_%s.npy" % (s, m)
            shutil.copyfile(os.path.join(lut_dir, fn), os.path.join(out_luts, fn))
            os.chmod(os.path.join(out_luts, fn), 0o644)
            shipped["s%d_%s" % (s, m)] = np.load(os.path.join(lut_dir, fn))
    for sub, src in (("LR_bicubic/X4", "data/SRBenchmark/Set5/LR_bicubic/X4"), ("HR", "data/SRBenchmark/Set5/HR"),
                     ("ref_out", "results/sr_x2sdy/Set5/X4")):
        d = os.path.join(HERE, "Set5", sub)
        os.makedirs(d, exist_ok=True)
        for fn in sorted(os.listdir(os.path.join(REF, src))):
            shutil.copyfile(os.path.join(REF, src, fn), os.path.join(d, fn))
            os.chmod(os.path.join(d, fn), 0o644)

    # the one DIV2K LR sample the reference ships: real-photo content for bench.py --dist real
    os.makedirs(os.path.join(HERE, "DIV2K_LR_X4"), exist_ok=True)
    shutil.copyfile(os.path.join(REF, "data/DIV2K/LR/X4/0001x4.png"), os.path.join(HERE, "DIV2K_LR_X4", "0001x4.png"))
    os.chmod(os.path.join(HERE, "DIV2K_LR_X4", "0001x4.png"), 0o644)

    f32 = {k: v.astype(np.float32) for k, v in shipped.items()}
    luts2 = {k: f32[k].reshape(-1, 16 if k.startswith("s2") else 1) for k in f32}

    inputs = make_inputs()

    # ---- G2: per-pass fixtures (16*out as int32) ----
    g2 = {}
    pass_cases = ["rand_19x13x3", "rand_8x31x1", "rand_5x4x2", "rand_2x7x1", "one_1x1x1", "ties_9x11x3",
                  "extreme_12x10x3"]
    for name in pass_cases:
        img = inputs[name]
        g2["in/" + name] = img
        for u, st in ((1, 1), (4, 2)):
            for m in "sdy":
                for r in range(4):
                    g2["out/%s/u%d/%s/r%d" % (name, u, m, r)] = ref_pass(t4, luts2["s%d_%s" % (st, m)], img, r, u, m)
    np.savez_compressed(os.path.join(HERE, "pass_fixtures.npz"), **g2)

    # ---- G3: pipeline fixtures ----
    g3 = {}
    crop = np.array(Image.open(os.path.join(REF, "data/DIV2K/LR/X4/0001x4.png")))[100:164, 200:264, :3]
    inputs["div2k_crop_64x64x3"] = np.ascontiguousarray(crop)
    for name, img in inputs.items():
        g3["in/" + name] = img
        # 2-stage sdy x4, shipped tables: stage outputs by the reference's loop, final by its own driver
        st = ref_stages(t4, luts2, img, 2, "sdy", 4)
        fin = ref_driver(t5, luts2, img, 2, "sdy", 4)
        if img.shape[2] == 3:
            assert np.array_equal(fin, st[1])
        elif img.shape[2] == 1:
            assert np.array_equal(fin, np.repeat(st[1], 3, axis=2))
        g3["s2sdy/%s/stage1" % name] = st[0]
        g3["s2sdy/%s/final" % name] = st[1]
    # other stage/mode combinations on a few inputs
    combos = [(1, "s", 4), (1, "sdy", 4), (2, "sd", 4), (2, "y", 4), (3, "sdy", 4), (4, "sdy", 2), (2, "s", 3)]
    for stages, modes, scale in combos:
        luts = {}
        for s in range(stages):
            last = (s + 1) == stages
            for mi, m in enumerate(modes):
                seed = 1000 * stages + 100 * scale + 10 * s + "sdy".index(m)
                luts["s%d_%s" % (s + 1, m)] = synthetic_lut(seed, scale * scale if last else 1).astype(np.float32)
        for name in ("rand_19x13x3", "rand_5x4x2", "extreme_12x10x3", "smooth_24x20x3"):
            img = inputs[name]
            st = ref_stages(t4, luts, img, stages, modes, scale)
            if img.shape[2] == 3:
                fin = ref_driver(t5, luts, img, stages, modes, scale)
                assert np.array_equal(fin, st[-1])
            key = "synth_S%d_%s_x%d/%s" % (stages, modes, scale, name)
            for i, o in enumerate(st):
                g3["%s/stage%d" % (key, i + 1)] = o
    # config-1 analogue: shipped s2_s table used as a 1-stage 's' x4 model
    for name in ("rand_19x13x3", "smooth_24x20x3"):
        st = ref_stages(t4, {"s1_s": luts2["s2_s"]}, inputs[name], 1, "s", 4)
        g3["cfg1_s2s_as_s1/%s/final" % name] = st[0]
    np.savez_compressed(os.path.join(HERE, "pipeline_fixtures.npz"), **g3)

    # ---- G1: Set5 end-to-end through the reference driver, compared with its committed PNGs ----
    g1 = {}
    for fn in sorted(os.listdir(os.path.join(HERE, "Set5", "LR_bicubic/X4"))):
        lr = np.array(Image.open(os.path.join(HERE, "Set5", "LR_bicubic/X4", fn)))
        if lr.ndim == 2:
            lr = lr[:, :, None]
        out = ref_driver(t5, luts2, lr, 2, "sdy", 4)
        committed = np.array(Image.open(os.path.join(HERE, "Set5", "ref_out", fn[:-4] + "_LUT_ft_4bit.png")))
        assert np.array_equal(out, committed), fn
        g1[fn[:-4] + "/shape"] = np.array(out.shape)
        print("Set5", fn, "reference run == committed PNG", out.shape)
    print("fixtures written:", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
