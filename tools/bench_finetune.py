#!/usr/bin/env python3
"""Config 4 of BASELINE.json: LUT fine-tune step (forward + backward + Adam) at bs=256, 1x48x48 crops, 2-stage
sdy x4, on one GPU.  Reference point: models/sr_x2sdy/lutft.log logs rT ~= 7.0 s/iter at batch 320 on the
authors' 2022 GPU (0.105 M LR-px/s).   python tools/bench_finetune.py [--bs 256 --iters 20]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mulut_amd.finetune import MuLUT  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=256)
    ap.add_argument("--crop", type=int, default=48)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--data", choices=["natural", "noise"], default="natural",
                    help="natural: crops of the smooth synthetic field (what photographs look like to the tables); noise: uniform random bytes")
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as td:
        for s in (1, 2):
            for m in "sdy":
                src = os.path.join(ROOT, "tests", "golden", "luts", "LUT_ft_x4_4bit_int8_s%d_%s.npy" % (s, m))
                np.save(os.path.join(td, "LUT_x4_4bit_int8_s%d_%s.npy" % (s, m)), np.load(src))
        net = MuLUT(td, 2, "sdy", upscale=4, interval=4).cuda()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, fused=True)       # one launch for the six tables
    g = torch.Generator(device="cuda").manual_seed(0)
    if args.data == "noise":
        x = torch.randint(0, 256, (args.bs, 1, args.crop, args.crop), device="cuda", generator=g).float() / 255.0
    else:
        from mulut_amd.synth import natural_frames
        big = natural_frames(1, 1080, 1920, 1, 0)[0, :, :, 0]
        rng = np.random.default_rng(0)
        ys, xs = rng.integers(0, 1080 - args.crop, args.bs), rng.integers(0, 1920 - args.crop, args.bs)
        x = torch.from_numpy(np.stack([big[a:a + args.crop, b:b + args.crop] for a, b in zip(ys, xs)])[:, None].astype(np.float32) / 255.0).cuda()
    y = torch.rand((args.bs, 1, args.crop * 4, args.crop * 4), device="cuda", generator=g)

    def step():
        opt.zero_grad()
        loss = torch.nn.functional.mse_loss(net(x), y)
        loss.backward()
        opt.step()
        return loss

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.iters
    # forward / backward alone (device time)
    for _ in range(2):      # (second round is the one reported: the first pays the autograd thread's start-up)
        opt.zero_grad()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record(); out = net(x); ev[1].record(); torch.nn.functional.mse_loss(out, y).backward(); ev[2].record()
        torch.cuda.synchronize()
    print(json.dumps({"metric": "LUT fine-tune step (fwd+bwd+Adam), 2-stage sdy x4", "batch": args.bs, "crop": args.crop, "data": args.data,
                      "forward_ms": round(ev[0].elapsed_time(ev[1]), 3), "backward_ms": round(ev[1].elapsed_time(ev[2]), 3),
                      "s_per_iter": round(dt, 5), "lr_Mpx_per_s": round(args.bs * args.crop ** 2 / dt / 1e6, 3),
                      "loss": float(loss.item()),
                      "reference_logged": "7.0 s/iter at batch 320 (models/sr_x2sdy/lutft.log:7-26), unspecified 2022 GPU"}))


if __name__ == "__main__":
    main()
