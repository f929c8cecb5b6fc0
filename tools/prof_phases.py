#!/usr/bin/env python3
"""Phase table of stage_tube2_kernel from its probe build (in-kernel clock stamps summed over all waves into the context's
probe buffer -- never into an output).    python tools/prof_phases.py [--frames 8]      (GPU box)"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from ab_bench import build_variant  # noqa: E402
from mulut_amd import MuLUTEngine, load_lut_dict  # noqa: E402
from mulut_amd.synth import natural_frames  # noqa: E402

PHASES = ["tile decode + next fetch issue", "pipeline prologue", "channels (36 passes + 3 epilogues)", "stores + fix-up list",
          "stash of the next tile", "barrier"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--build-only", action="store_true")
    args = ap.parse_args()
    so = build_variant("t2prof")
    if args.build_only:
        return
    luts = load_lut_dict(os.path.join(ROOT, "tests", "golden", "luts"), 2, "sdy", 4, 4, "LUT_ft")
    e = MuLUTEngine(0, lib_path=so).configure(2, "sdy", 4, 4).set_lut_dict(luts)
    x = torch.from_numpy(natural_frames(2, 1080, 1920, 3, 0)).cuda().repeat(args.frames // 2, 1, 1, 1).contiguous()
    out = torch.empty((args.frames, 4320, 7680, 3), dtype=torch.uint8, device="cuda")
    e.reserve(args.frames, 1080, 1920, 3)
    e.debug_read(16, True)
    for _ in range(3):
        e.pipeline(x, out=out)
    e.debug_read(16, True)
    reps = 5
    for _ in range(reps):
        e.pipeline(x, out=out)
    d = e.debug_read(16, True)
    waves = d[8]
    total = sum(d[:6])
    clock = d[6] / max(d[7], 1) * 0.1
    tiles = reps * args.frames * 120 * 270 / 16.0      # wave tiles per wave... (16 x 4 pixels each)
    rec = {"kernel": "stage_tube2_kernel (probe build t2prof)", "frames": args.frames, "launches": reps, "waves": waves,
           "in_kernel_clock_ghz": round(clock, 3), "wave_lifetime_us": round(d[6] / max(waves, 1) / clock / 1e3, 1),
           "phases": [{"phase": PHASES[k], "share": round(d[k] / total, 4), "cycles_per_tile_and_wave": round(d[k] / (tiles * 16.0), 1)} for k in range(6)]}
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
