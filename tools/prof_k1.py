#!/usr/bin/env python3
"""Phase shares of stage_u1t_kernel from its probe build (k1prof).  (GPU box)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from ab_bench import build_variant
from mulut_amd import MuLUTEngine, load_lut_dict
from mulut_amd.synth import natural_frames
so = build_variant("k1prof")
if "--build-only" in sys.argv:
    sys.exit(0)
luts = load_lut_dict(os.path.join(ROOT, "tests", "golden", "luts"), 2, "sdy", 4, 4, "LUT_ft")
e = MuLUTEngine(0, lib_path=so).configure(2, "sdy", 4, 4).set_lut_dict(luts)
e.set_tuning("first_stage_kernel", 0)
x = torch.from_numpy(natural_frames(2, 1080, 1920, 3, 0)).cuda().repeat(4, 1, 1, 1).contiguous()
e.debug_read(32, True)
for _ in range(3):
    e.pipeline(x)
e.debug_read(32, True)
for _ in range(5):
    e.pipeline(x)
d = e.debug_read(32, True)[16:20]
tot = sum(d)
for name, v in zip(("bands + first barrier", "routing statistic", "tile load + barrier", "sites"), d):
    print("%-24s %.3f" % (name, v / tot))
