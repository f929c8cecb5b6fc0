#!/usr/bin/env python3
"""Bit-compare every final-stage kernel variant against the full-table kernel on smooth / photo / noise frames."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mulut_amd import MuLUTEngine, load_lut_dict
from mulut_amd.synth import natural_frames, noise_frames, real_frames

luts = load_lut_dict(os.path.join(ROOT, "tests", "golden", "luts"), 2, "sdy", 4, 4, "LUT_ft")
lib = [a.split("=",1)[1] for a in sys.argv[1:] if a.startswith("--lib=")]
e = MuLUTEngine(0, lib_path=lib[0] if lib else None).configure(2, "sdy", 4, 4).set_lut_dict(luts)
png = os.path.join(ROOT, "tests", "golden", "DIV2K_LR_X4", "0001x4.png")
ok = True
for (h, w) in ((150, 200), (37, 129), (270, 480)):
    fr = np.concatenate([natural_frames(1, h, w, 3, 1), noise_frames(1, h, w, 3, 1), real_frames(1, h, w, png, 1)])
    x = torch.from_numpy(fr).cuda()
    e.set_tuning("final_stage_kernel", 1)
    want = e.pipeline(x).clone()
    e.set_tuning("first_stage_kernel", 2)
    want = e.pipeline(x).clone()
    for first in (0, 2, 3):
        for sel in (5, 6):
            for det, t2 in (((0, 1), (1, 1), (0, 0)) if sel == 6 else ((0, 1), (0, 0))):      # detailed tiles of the hybrid: anchor slabs / full-table gathers; tube2 / tube
                e.set_tuning("final_stage_kernel", sel).set_tuning("first_stage_kernel", first).set_tuning("detail_kernel", det).set_tuning("tube_pipelined", t2)
                got = e.pipeline(x)
                same = torch.equal(got, want)
                ok &= same
                bad = (got != want)
                where = "" if same else " first at %s" % (tuple(int(v) for v in bad.nonzero()[0]),)
                print(h, w, "first", first, "final", sel, "detail", det, "pipelined", t2, "OK" if same else "MISMATCH %d%s" % (int(bad.sum()), where))
sys.exit(0 if ok else 1)
