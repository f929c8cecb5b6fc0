#!/usr/bin/env python3
"""Repeat-run stress of ONE engine: the Set5 LR images and a few synthetic frames in random order, thousands of times, the
torch memory pool poisoned between calls; every result is compared with the CPU oracle's (computed once per image).  Catches
results that depend on stale internal buffers or on timing.   python tools/stress_repeat.py [--runs 3000] [--seed 1]"""
import argparse, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from PIL import Image  # noqa: E402
from mulut_amd import MuLUTEngine, load_lut_dict  # noqa: E402
from mulut_amd.synth import natural_frames, noise_frames  # noqa: E402
from oracle import c_oracle  # noqa: E402  (checker only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=3000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--tuning", default="")
    ap.add_argument("--fresh-engine-every", type=int, default=0,
                    help="N > 0: a new engine every N runs, created after the driver's free memory was filled with random bytes "
                         "(its hipMalloc'ed buffers then start from garbage, as in a long-lived process)")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    G = os.path.join(ROOT, "tests", "golden")
    luts = load_lut_dict(os.path.join(G, "luts"), 2, "sdy", 4, 4, "LUT_ft")
    d = os.path.join(G, "Set5", "LR_bicubic", "X4")
    imgs = [(fn, np.array(Image.open(os.path.join(d, fn)))) for fn in sorted(os.listdir(d))]
    imgs += [("noise_%dx%d" % (h, w), noise_frames(1, h, w, 3, h)[0]) for h, w in ((40, 100), (33, 64), (96, 200))]
    imgs += [("natural_%dx%d" % (h, w), natural_frames(1, h, w, 3, w)[0]) for h, w in ((64, 128), (50, 77))]
    mixed = natural_frames(1, 80, 192, 3, 5)[0].copy()
    mixed[:, 96:] = noise_frames(1, 80, 96, 3, 6)[0]
    imgs.append(("mixed_80x192", mixed))
    want = [torch.from_numpy(c_oracle.pipeline(luts, 2, "sdy", 4, im)).cuda() for _, im in imgs]
    dev = [torch.from_numpy(np.ascontiguousarray(im)).cuda() for _, im in imgs]
    e = MuLUTEngine(0).configure(2, "sdy", 4, 4).set_lut_dict(luts)
    for kv in [t for t in args.tuning.split(",") if t]:
        e.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
    bad = []
    for run in range(args.runs):
        if args.fresh_engine_every and run and run % args.fresh_engine_every == 0:
            e.close()
            junk = [torch.randint(0, 256, (256 << 20,), dtype=torch.uint8, device="cuda") for _ in range(8)]
            del junk
            torch.cuda.empty_cache()          # back to the driver, dirty
            e = MuLUTEngine(0).configure(2, "sdy", 4, 4).set_lut_dict(luts)
            for kv in [t for t in args.tuning.split(",") if t]:
                e.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
        i = int(rng.integers(len(imgs)))
        junk = torch.randint(0, 256, (int(rng.integers(1, 9)) << 20,), dtype=torch.uint8, device="cuda")   # poison the pool
        del junk
        got = e.pipeline(dev[i])
        if not torch.equal(got, want[i]):
            diff = (got != want[i]).nonzero()
            bad.append({"run": run, "image": imgs[i][0], "mismatches": int(diff.shape[0]),
                        "first": [int(v) for v in diff[0]], "last": [int(v) for v in diff[-1]],
                        "rows": sorted(set(int(v) // 4 for v in diff[:, 0].tolist()))[:12],
                        "cols": sorted(set(int(v) // 4 for v in diff[:, 1].tolist()))[:24],
                        "chans": sorted(set(diff[:, 2].tolist()))})
            if len(bad) >= 8:
                break
    print(json.dumps({"runs": run + 1, "failed": len(bad), "failures": bad}))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
