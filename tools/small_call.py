#!/usr/bin/env python3
"""Set5-sized calls of the 2-stage sdy x4 cascade (one image per mulut_pipeline call), for a kernel trace of small launches:
    rocprofv3 --kernel-trace --stats -- python3 tools/small_call.py [--h 128 --w 128 --reps 50]
Prints host-side milliseconds per call (synchronised) as one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mulut_amd import MuLUTEngine, load_lut_dict  # noqa: E402
from mulut_amd.synth import natural_frames  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--h", type=int, default=128)
    ap.add_argument("--w", type=int, default=128)
    ap.add_argument("--n", type=int, default=1)
    ap.add_argument("--reps", type=int, default=50)
    a = ap.parse_args()
    luts = load_lut_dict(os.path.join(ROOT, "tests", "golden", "luts"), 2, "sdy", 4, 4, "LUT_ft")
    eng = MuLUTEngine(0).configure(2, "sdy", 4, 4).set_lut_dict(luts)
    eng.reserve(a.n, a.h, a.w, 3)
    x = torch.from_numpy(natural_frames(a.n, a.h, a.w, 3, 0)).cuda()
    out = torch.empty((a.n, a.h * 4, a.w * 4, 3), dtype=torch.uint8, device="cuda")
    for _ in range(5):
        eng.pipeline(x, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        eng.pipeline(x, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps * 1e3
    print(json.dumps({"workload": "%d x LR %dx%dx3, 2-stage sdy x4, one mulut_pipeline call" % (a.n, a.h, a.w), "ms_per_call": round(dt, 4)}))


if __name__ == "__main__":
    main()
