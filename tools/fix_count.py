#!/usr/bin/env python3
"""How many entries does the final stage's fix-up list hold?  (GPU box)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mulut_amd import MuLUTEngine, load_lut_dict
from mulut_amd.synth import natural_frames
luts = load_lut_dict(os.path.join(ROOT, "tests", "golden", "luts"), 2, "sdy", 4, 4, "LUT_ft")
e = MuLUTEngine(0).configure(2, "sdy", 4, 4).set_lut_dict(luts)
x = torch.from_numpy(natural_frames(2, 1080, 1920, 3, 0)).cuda().repeat(4, 1, 1, 1).contiguous()
e.pipeline(x)
d = e.last_detail_counters()
print("samples", 8 * 1080 * 1920 * 3, "fix entries", d["fix_pixels"], "slab samples", sum(d["samples_per_anchor"]), "items", d["items"])
for thr in (256, 512):
    e.set_tuning("hybrid_oob_per_1024", thr)
    e.pipeline(x)
    d = e.last_detail_counters()
    print("thr", thr, "fix entries", d["fix_pixels"], "slab samples", sum(d["samples_per_anchor"]), "items", d["items"])
