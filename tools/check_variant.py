#!/usr/bin/env python3
"""Bit-exactness of a variant build against the shipped library on natural / noise / real frames (GPU box):
    python tools/check_variant.py k1pkmad [--h 270 --w 480 --frames 3]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from ab_bench import build_variant
from mulut_amd import MuLUTEngine, load_lut_dict
from mulut_amd.synth import natural_frames, noise_frames, real_frames
ap = argparse.ArgumentParser(); ap.add_argument("variant"); ap.add_argument("--h", type=int, default=270); ap.add_argument("--w", type=int, default=480); ap.add_argument("--frames", type=int, default=3)
a = ap.parse_args()
luts = load_lut_dict(os.path.join(ROOT, "tests", "golden", "luts"), 2, "sdy", 4, 4, "LUT_ft")
e0 = MuLUTEngine(0).configure(2, "sdy", 4, 4).set_lut_dict(luts)
e1 = MuLUTEngine(0, lib_path=build_variant(a.variant)).configure(2, "sdy", 4, 4).set_lut_dict(luts)
png = os.path.join(ROOT, "tests", "golden", "DIV2K_LR_X4", "0001x4.png")
bad = 0
for name, fr in (("natural", natural_frames(a.frames, a.h, a.w, 3, 5)), ("noise", noise_frames(a.frames, a.h, a.w, 3, 5)), ("real", real_frames(a.frames, a.h, a.w, png, 5))):
    x = torch.from_numpy(fr).cuda()
    y0, y1 = e0.pipeline(x), e1.pipeline(x)
    same = bool(torch.equal(y0, y1))
    print(name, "identical" if same else "DIFFERENT")
    if not same:
        d = (y0.int() - y1.int()).abs()
        nz = torch.nonzero(d)
        print("   differing bytes %d of %d, max |diff| %d, first at %s, stage-1 outputs equal: %s" % (nz.shape[0], d.numel(), int(d.max()), nz[0].tolist(),
              bool(torch.equal(e0.stage(1, x), e1.stage(1, x)))))
        m1 = (e0.stage(1, x).int() - e1.stage(1, x).int()).abs()
        nz1 = torch.nonzero(m1)
        if nz1.shape[0]:
            print("   stage 1: differing %d, max %d, first at %s; values %d vs %d" % (nz1.shape[0], int(m1.max()), nz1[0].tolist(), int(e0.stage(1, x)[tuple(nz1[0].tolist())]), int(e1.stage(1, x)[tuple(nz1[0].tolist())])))
    bad += not same
sys.exit(1 if bad else 0)
