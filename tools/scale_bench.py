#!/usr/bin/env python3
"""Timing of a 2-stage sdy cascade at another scale than the headline's (x2, x3): the LDS path (final_stage_kernel 0) against the
gather kernels (final_stage_kernel 1), seeded synthetic tables.   python tools/scale_bench.py --scale 3 [--frames 8] [--dist natural]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mulut_amd import MuLUTEngine  # noqa: E402
from bench import make_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=3)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--dist", default="natural")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--detail", type=int, default=None, help="tile threshold of the routed launch (tuning key final_stage_detail_per_1024)")
    args = ap.parse_args()
    u, rng = args.scale, np.random.default_rng(7)
    eng = MuLUTEngine(0).configure(2, "sdy", u, 4)
    for s in (1, 2):
        for m in "sdy":
            vn = u * u if s == 2 else 1
            base = rng.integers(-20, 21, (17, 17, 17, 17, vn)).astype(np.float32)
            grid = np.indices((17, 17, 17, 17)).astype(np.float32).sum(0)[..., None] * (3.0 if s == 1 else 4.0) - 96.0
            eng.set_lut(s, m, np.clip(np.rint(grid + base), -127, 127).astype(np.int8).reshape(-1, vn))
    eng.reserve(args.frames, args.h, args.w, 3)
    x = torch.from_numpy(make_batch(args.dist, args.frames, args.h, args.w, seed=0)).cuda()
    out = torch.empty((args.frames, args.h * u, args.w * u, 3), dtype=torch.uint8, device="cuda")
    ref = None
    if args.detail is not None:
        eng.set_tuning("final_stage_detail_per_1024", args.detail)
    for sel in (1, 0):
        eng.set_tuning("final_stage_kernel", sel)
        eng.pipeline(x, out=out)
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        same = bool(torch.equal(out, ref))
        ts = []
        for _ in range(args.rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            eng.pipeline(x, out=out)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(json.dumps({"scale": u, "dist": args.dist, "frames": args.frames, "final_stage_kernel": sel, "detail_per_1024": args.detail, "kernel": eng.kernel_name(True),
                          "ms_median": round(float(np.median(ts)), 3), "ms_min": round(min(ts), 3), "identical_to_gather": same}))


if __name__ == "__main__":
    main()
