#!/usr/bin/env python3
"""Where does the pipelined tube kernel differ from the compiler-scheduled one?  (GPU box)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mulut_amd import MuLUTEngine, load_lut_dict
from mulut_amd.synth import natural_frames, real_frames

luts = load_lut_dict(os.path.join(ROOT, "tests", "golden", "luts"), 2, "sdy", 4, 4, "LUT_ft")
lib = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--lib=")]
brief = "--brief" in sys.argv
e = MuLUTEngine(0, lib_path=lib[0] if lib else None).configure(2, "sdy", 4, 4).set_lut_dict(luts)
png = os.path.join(ROOT, "tests", "golden", "DIV2K_LR_X4", "0001x4.png")
for name, fr in (("natural", natural_frames(1, 150, 200, 3, 1)), ("real", real_frames(1, 150, 200, png, 1))):
    x = torch.from_numpy(fr).cuda()
    e.set_tuning("final_stage_kernel", 5).set_tuning("tube_pipelined", 0)
    want = e.pipeline(x).cpu().numpy().astype(np.int32)
    e.set_tuning("tube_pipelined", 1)
    got = e.pipeline(x).cpu().numpy().astype(np.int32)
    bad = got != want
    print(name, "mismatches", int(bad.sum()), "of", bad.size)
    if not bad.any() or brief:
        continue
    n, yy, xx, cc = np.nonzero(bad)
    print("  per channel", [int((cc == c).sum()) for c in range(3)])
    sub = np.zeros((4, 4), int)
    np.add.at(sub, (yy % 4, xx % 4), 1)
    print("  per sub-pixel\n", sub)
    d = (got - want)[bad]
    print("  diff min/max/mean", d.min(), d.max(), d.mean(), " |diff| hist", np.bincount(np.minimum(np.abs(d), 20)))
    sites = set(zip(yy // 4, xx // 4, cc))
    print("  bad (site, channel) count", len(sites), "of", 150 * 200 * 3)
    ys = np.array([s[0] for s in sites]); xs = np.array([s[1] for s in sites])
    print("  site x % 64 hist (16 bins)", np.bincount((xs % 64) // 4, minlength=16))
    print("  site y % 16 hist", np.bincount(ys % 16, minlength=16))
    for (y, x_, c) in sorted(sites)[:6]:
        print("  site", y, x_, c, "in 5x5:", fr[0, max(y - 2, 0):y + 3, max(x_ - 2, 0):x_ + 3, c].tolist())
        print("     got ", got[0, 4 * y:4 * y + 4, 4 * x_:4 * x_ + 4, c].tolist())
        print("     want", want[0, 4 * y:4 * y + 4, 4 * x_:4 * x_ + 4, c].tolist())
