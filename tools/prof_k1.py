#!/usr/bin/env python3
"""Phase shares of stage_u1t_kernel from its probe build (k1prof).  (GPU box)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from ab_bench import build_variant
from mulut_amd import MuLUTEngine, load_lut_dict
from mulut_amd.synth import natural_frames
so = build_variant("k1prof2" if "--fine" in sys.argv else "k1prof")
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
dd = e.debug_read(32, True)
d = dd[16:20]
tot = sum(d)
for name, v in zip(("bands + first barrier", "routing statistic", "tile load + barrier", "sites + list pass"), d):
    print("%-36s %.3f" % (name, v / tot))
life, real, waves = dd[20], dd[21], dd[22]
clock = life / max(real, 1) * 0.1
passes = 5 * 8 * 1080 * 1920 * 3 * 12 / 64.0          # pass-waves of the five timed launches
print("waves %d  in-kernel clock %.3f GHz  wave lifetime %.1f us" % (waves, clock, life / max(waves, 1) / clock / 1e3))
if "--fine" in sys.argv:
    f = dd[24:28]
    for name, v in zip(("window reads + neighbourhood test", "pixel bodies (24 passes per step)", "stores + dirty byte", "barrier + list pass"), f):
        print("   %-36s %.3f of the wave's life, %.1f wave-cycles per pass" % (name, v / life, v / passes))
print("wave-cycles per pass-wave %.1f  (x 1/8 at 8 waves per SIMD = %.1f SIMD cycles per pass-wave)" % (life / passes, life / passes / 8))
