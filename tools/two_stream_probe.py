#!/usr/bin/env python3
"""Does running two half batches on two streams (two engines, own work lists) hide the small launch-bound kernels of a step
behind the other half's main kernels?   python tools/two_stream_probe.py [--frames 32]      (GPU box)"""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mulut_amd import MuLUTEngine, load_lut_dict
from mulut_amd.synth import natural_frames, noise_frames, real_frames

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=32)
ap.add_argument("--steps", type=int, default=10)
args = ap.parse_args()
luts = load_lut_dict(os.path.join(ROOT, "tests", "golden", "luts"), 2, "sdy", 4, 4, "LUT_ft")
F, H, W = args.frames, 1080, 1920
png = os.path.join(ROOT, "tests", "golden", "DIV2K_LR_X4", "0001x4.png")
for dist, make in (("natural", lambda: natural_frames(2, H, W, 3, 0)), ("noise", lambda: noise_frames(2, H, W, 3, 0)), ("real", lambda: real_frames(2, H, W, png, 0))):
    x = torch.from_numpy(make()).cuda().repeat(F // 2, 1, 1, 1).contiguous()
    out = torch.empty((F, 4 * H, 4 * W, 3), dtype=torch.uint8, device="cuda")
    for parts in (1, 2, 4):
        engs = [MuLUTEngine(0).configure(2, "sdy", 4, 4).set_lut_dict(luts) for _ in range(parts)]
        streams = [torch.cuda.Stream() for _ in range(parts)]
        n = F // parts
        for e in engs:
            e.reserve(n, H, W, 3)
        def step():
            for p in range(parts):
                with torch.cuda.stream(streams[p]):
                    engs[p].pipeline(x[p * n:(p + 1) * n], out=out[p * n:(p + 1) * n])
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        ref = None
        if parts == 1:
            want = out.clone()
        else:
            ref = bool(torch.equal(out, want))
        print(json.dumps({"dist": dist, "streams": parts, "ms_per_step": round(ms, 3), "gpix_s": round(F * 16 * H * W / ms / 1e6, 1), "same_bytes": ref}), flush=True)
        del engs
    del x, out, want
