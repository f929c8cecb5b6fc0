#!/usr/bin/env python3
"""In-tube statistics of the final stage's input per routing granularity.  CPU only (the oracle computes
the first stage): for one LR 1080x1920x3 frame of a distribution, the share of the 12 passes of a sample that stay in the tube
(max - min of the four MSBs <= 1), the share of samples with every pass in the tube ("clean"), and how both distribute over the
64x16 verdict tiles and over the 16x4 tiles a stage_tube2_kernel wave owns.
    python tools/tube_histogram.py [--dist real] [--h 1080 --w 1920] > profiles/r04_tube_histogram_real.json"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mulut_amd import load_lut_dict  # noqa: E402
from mulut_amd.synth import natural_frames, noise_frames, real_frames  # noqa: E402
from oracle import c_oracle  # noqa: E402

PAT = {"s": [(0, 1), (1, 0), (1, 1)], "d": [(0, 2), (2, 0), (2, 2)], "y": [(1, 1), (1, 2), (2, 1)]}


def rot(r, di, dj):
    return [(di, dj), (dj, -di), (-di, -dj), (-dj, di)][r]


def shifted(h, dy, dx):
    H, W = h.shape[:2]
    ys = np.clip(np.arange(H) + dy, 0, H - 1)
    xs = np.clip(np.arange(W) + dx, 0, W - 1)
    return h[ys][:, xs]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dist", default="real")
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--stage", type=int, default=2, help="2: statistics of the final stage's input (= first-stage output); 1: of the first stage's input")
    a = ap.parse_args()
    png = os.path.join(ROOT, "tests", "golden", "DIV2K_LR_X4", "0001x4.png")
    img = {"real": lambda: real_frames(1, a.h, a.w, png, 0), "natural": lambda: natural_frames(1, a.h, a.w, 3, 0),
           "noise": lambda: noise_frames(1, a.h, a.w, 3, 0)}[a.dist]()[0]
    luts = load_lut_dict(os.path.join(ROOT, "tests", "golden", "luts"), 2, "sdy", 4, 4, "LUT_ft")
    x = img if a.stage == 1 else c_oracle.stage([luts["s1_%s" % m] for m in "sdy"], "sdy", False, img, 1)
    h = (x >> 4).astype(np.int8)        # HWC MSBs
    H, W, C = h.shape
    out_cnt = np.zeros((H, W, C), np.int16)      # passes of the sample that leave the tube
    for m in "sdy":
        for r in range(4):
            hs = [h] + [shifted(h, *rot(r, di, dj)) for di, dj in PAT[m]]
            mx = np.maximum(np.maximum(hs[0], hs[1]), np.maximum(hs[2], hs[3]))
            mn = np.minimum(np.minimum(hs[0], hs[1]), np.minimum(hs[2], hs[3]))
            out_cnt += (mx - mn > 1)
    dirty = out_cnt > 0
    rec = {"dist": a.dist, "stage_input": a.stage, "frame": [H, W, C], "passes_in_tube": round(1 - out_cnt.sum() / (12.0 * H * W * C), 5),
           "samples_clean": round(1 - dirty.mean(), 5), "tiles": {}}
    for name, (tw, th) in (("64x16", (64, 16)), ("16x4", (16, 4)), ("16x16", (16, 16)), ("64x64", (64, 64))):
        Hp, Wp = -(-H // th) * th, -(-W // tw) * tw
        d = np.zeros((Hp, Wp, C), np.float64); d[:H, :W] = dirty
        v = np.zeros((Hp, Wp, C), np.float64); v[:H, :W] = 1
        o = np.zeros((Hp, Wp, C), np.float64); o[:H, :W] = out_cnt
        red = lambda z: z.reshape(Hp // th, th, Wp // tw, tw, C).sum((1, 3, 4))      # noqa: E731
        nd, nv, no = red(d), red(v), red(o)
        share = nd / nv                        # dirty-sample share per tile
        edges = [0, 1e-9, 0.01, 0.02, 0.05, 0.125, 0.25, 0.5, 0.75, 1.0000001]
        hist = np.histogram(share, bins=edges, weights=nv)[0] / nv.sum()
        t = {"tiles": int(share.size), "dirty_share_bins": ["0", "(0,1%]", "(1,2%]", "(2,5%]", "(5,12.5%]", "(12.5,25%]", "(25,50%]", "(50,75%]", "(75,100%]"],
             "samples_in_tiles_by_dirty_share": [round(float(z), 4) for z in hist],
             "pass_in_tube_share_of_tiles_median": round(float(np.median(1 - no / (12 * nv))), 4), "routing": []}
        for thr in (0.02, 0.05, 0.125, 0.25, 0.5):
            fast = share <= thr
            t["routing"].append({"tile_to_tube_kernel_if_dirty_share_le": thr, "samples_on_tube_kernel": round(float(nv[fast].sum() / nv.sum()), 4),
                                 "of_all_samples_on_fixup_list": round(float(nd[fast].sum() / nv.sum()), 5),
                                 "samples_on_slab_path": round(float(nv[~fast].sum() / nv.sum()), 4)})
        rec["tiles"][name] = t
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
