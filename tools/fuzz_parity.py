#!/usr/bin/env python3
"""Randomised differential test: HIP path (through the C ABI) vs the CPU oracle over seeded random configurations --
stages 1..4, mode strings over {s,d,y} of length 1..4 (repeats allowed), scale 1..4, C 1..3, ragged sizes, HWC / CHW,
batches, whole-frame / strip calls, every final-stage kernel variant, smooth / photo-like / noisy / constant /
tie-heavy content, random / extreme tables.  Test infrastructure (uses oracle/); prints one JSON summary line.

    python tools/fuzz_parity.py --cases 300 --seed 1
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mulut_amd import MuLUTEngine, synthetic_lut  # noqa: E402
from mulut_amd.engine import LAYOUT_CHW, LAYOUT_HWC  # noqa: E402
from mulut_amd.synth import natural_frames  # noqa: E402
from oracle import c_oracle  # noqa: E402


def content(rng, kind, h, w, c):
    if kind == 0:
        return rng.integers(0, 256, (h, w, c), dtype=np.uint8)
    if kind == 1:
        return natural_frames(1, h, w, c, int(rng.integers(1 << 30)))[0]
    if kind == 2:                                   # ties: few distinct values, many equal LSBs
        return rng.choice(np.array([0, 15, 16, 17, 31, 128, 240, 255], np.uint8), (h, w, c))
    if kind == 3:
        return np.full((h, w, c), int(rng.integers(0, 256)), np.uint8)
    base = natural_frames(1, h, w, c, int(rng.integers(1 << 30)))[0].astype(np.int32)   # smooth + sparse edges
    mask = rng.random((h, w, 1)) < 0.03
    return np.clip(base + mask * rng.integers(-120, 121, (h, w, c)), 0, 255).astype(np.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    fails, done, px = [], 0, 0
    for case in range(args.cases):
        stages = int(rng.integers(1, 5))
        modes = "".join(rng.choice(list("sdy"), int(rng.integers(1, 9 if rng.random() < 0.2 else 5))))
        scale = int(rng.integers(1, 5))
        C = int(rng.integers(1, 4))
        h, w = int(rng.integers(1, 90)), int(rng.integers(1, 150))
        if rng.random() < 0.15:
            h, w = int(rng.integers(60, 200)), int(rng.integers(100, 300))
        table_kind = int(rng.integers(0, 4))
        luts = {}
        for s in range(stages):
            for m in set(modes):
                vn = scale * scale if s + 1 == stages else 1
                if table_kind == 3:
                    luts["s%d_%s" % (s + 1, m)] = np.full((17 ** 4, vn), int(rng.choice([-128, -127, 0, 127])), np.int8)
                else:
                    luts["s%d_%s" % (s + 1, m)] = synthetic_lut(int(rng.integers(1 << 30)), vn)
        e = MuLUTEngine(0).configure(stages, modes, scale, 4).set_lut_dict(luts)
        e.set_tuning("final_stage_kernel", int(rng.choice([0, 1, 5, 6])))
        e.set_tuning("tube_pipelined", int(rng.integers(0, 4) != 0))      # mostly the hand-scheduled kernel (the default)
        e.set_tuning("fix_kernel", int(rng.choice([0, 0, 0, 1, 2])))      # mostly one pass per lane (the default)
        e.set_tuning("hybrid_oob_per_1024", int(rng.choice([0, 16, 128, 512, 1024])))
        e.set_tuning("first_stage_kernel", int(rng.choice([0, 2, 3])))
        e.set_tuning("detail_kernel", int(rng.integers(0, 4) == 0))      # mostly the anchor-slab kernel (the default)
        e.set_tuning("stat_from_first_stage", int(rng.integers(0, 4) != 0))      # mostly on (the default)
        e.set_tuning("first_stage_detail_per_1024", int(rng.choice([0, 64, 256, 1024])))
        e.set_tuning("final_stage_detail_per_1024", int(rng.choice([0, 8, 64, 1024])))      # routing of the x2 / x3 final stages
        n = int(rng.integers(1, 4))
        imgs = np.stack([content(rng, int(rng.integers(0, 5)), h, w, C) for _ in range(n)])
        want = np.stack([c_oracle.pipeline(luts, stages, modes, scale, im) for im in imgs])
        how = int(rng.integers(0, 3))
        if how == 0:
            got = e.pipeline(torch.from_numpy(imgs).cuda()).cpu().numpy()
        elif how == 1:
            got = e.pipeline(torch.from_numpy(np.ascontiguousarray(imgs.transpose(0, 3, 1, 2))).cuda(), layout=LAYOUT_CHW)
            got = got.cpu().numpy().transpose(0, 2, 3, 1)
        else:                                       # strips of the first image, seams must be exact
            halo = e.halo
            k = int(rng.integers(1, 5))
            bounds = np.unique(np.linspace(0, h, k + 1).astype(int))
            parts = []
            for y0, y1 in zip(bounds[:-1], bounds[1:]):
                r0, r1 = max(0, int(y0) - halo), min(h, int(y1) + halo)
                parts.append(e.pipeline_rows(torch.from_numpy(np.ascontiguousarray(imgs[0][r0:r1])).cuda(), r0, int(y0), int(y1), h,
                                             layout=LAYOUT_HWC))
            got = torch.cat(parts, 0).cpu().numpy()[None]
            want = want[:1]
        ok = got.shape == want.shape and np.array_equal(got, want)
        done += 1
        px += int(np.prod(want.shape))
        if not ok:
            fails.append({"case": case, "stages": stages, "modes": modes, "scale": scale, "C": C, "h": h, "w": w, "how": how,
                          "mismatch": int((got != want).sum()) if got.shape == want.shape else "shape"})
        e.close()
    print(json.dumps({"cases": done, "seed": args.seed, "failed": len(fails), "failures": fails[:10], "output_bytes_compared": px,
                      "seconds": round(time.time() - t0, 1)}))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
