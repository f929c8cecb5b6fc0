#!/usr/bin/env python3
"""Randomised differential test of the fine-tune kernels (mulut_ft_stage_forward/backward through
mulut_amd.finetune.MuLUT) against the CPU oracle oracle/ft_torch.py (itself pinned to the reference's module by
tests/test_oracle_ft.py): seeded random stages (1-3), mode strings, upscale 1-4, batch / channel / ragged sizes,
uint8-valued and float-valued inputs, random tables.  Forward must agree to 1e-5 (values are k/255: a different
rounding decision would show as >= 1/255); loss to 1e-6; input and table gradients to rtol 2e-4 plus 2e-6 of the
tensor's largest magnitude (float32 sums of signed terms in a different -- atomic-add -- order: an element that
cancels to 1e-5 of the tensor's scale keeps an absolute, not a relative, error).
Test infrastructure; prints one JSON line.

    python tools/fuzz_finetune.py --cases 100 --seed 1
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
from oracle import ft_torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", type=int, default=-1, help="run just this case of the sequence (all draws are still made) and say more")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    fails = []
    worst = {"fwd": 0.0, "gx": 0.0, "gw": 0.0}
    for case in range(args.cases):
        stages = int(rng.integers(1, 4))
        modes = "".join(rng.permutation(list("sdy"))[:int(rng.integers(1, 4))])
        scale = int(rng.integers(1, 5))
        B, C = int(rng.integers(1, 4)), int(rng.integers(1, 3))
        H, W = int(rng.integers(3, 14)), int(rng.integers(3, 14))
        tabs = {}
        for s in range(stages):
            vn = scale * scale if s + 1 == stages else 1
            for m in modes:
                tabs["s%d_%s" % (s + 1, m)] = rng.integers(-127, 128, size=(17 ** 4, vn), dtype=np.int8)
        if rng.random() < 0.5:
            x_np = rng.integers(0, 256, (B, C, H, W)).astype(np.float32) / 255.0
        else:
            x_np = rng.random((B, C, H, W), dtype=np.float32)
        tgt = rng.random((B, C, H * scale, W * scale), dtype=np.float32)
        if args.only >= 0 and case != args.only:
            continue
        with tempfile.TemporaryDirectory() as d:
            for k, t in tabs.items():
                np.save(os.path.join(d, "LUT_x%d_4bit_int8_%s.npy" % (scale, k)), t)
            net = MuLUT(d, stages, modes, upscale=scale, interval=4).cuda()
        # device
        x = torch.from_numpy(x_np).cuda().requires_grad_(True)
        y = net(x)
        loss = torch.nn.functional.mse_loss(y, torch.from_numpy(tgt).cuda())
        loss.backward()
        # oracle (CPU, autograd)
        wref = {k: (torch.from_numpy(v.astype(np.float32)) / 127.0).requires_grad_(True) for k, v in tabs.items()}
        xr = torch.from_numpy(x_np).requires_grad_(True)
        yr = ft_torch.forward(wref, xr, stages, modes, scale)
        lr = torch.nn.functional.mse_loss(yr, torch.from_numpy(tgt))
        lr.backward()
        e_f = float((y.detach().cpu() - yr.detach()).abs().max())
        ok = e_f <= 1e-5 and abs(loss.item() - lr.item()) <= 1e-6
        gx, gxr = x.grad.cpu().numpy(), xr.grad.numpy()
        close = lambda a, b: bool(np.allclose(a, b, rtol=2e-4, atol=1e-7 + 2e-6 * float(np.abs(b).max())))  # noqa: E731
        ok &= close(gx, gxr)
        worst["fwd"] = max(worst["fwd"], e_f)
        worst["gx"] = max(worst["gx"], float(np.abs(gx - gxr).max()))
        if args.only >= 0:
            bad = ~np.isclose(gx, gxr, rtol=2e-4, atol=1e-7)
            print("grad_x: mismatching", int(bad.sum()), "of", bad.size, "max |ref|", float(np.abs(gxr).max()),
                  [(float(a), float(b)) for a, b in zip(gx[bad][:5], gxr[bad][:5])], file=sys.stderr)
        for k in tabs:
            g = getattr(net, "weight_" + k).grad.cpu().numpy()
            gr = wref[k].grad.numpy()
            ok &= close(g, gr)
            worst["gw"] = max(worst["gw"], float(np.abs(g - gr).max()))
            if args.only >= 0:
                bad = ~np.isclose(g, gr, rtol=2e-4, atol=1e-7)
                print(k, "mismatching", int(bad.sum()), "max |ref|", float(np.abs(gr).max()),
                      [(float(a), float(b)) for a, b in zip(g[bad][:5], gr[bad][:5])], file=sys.stderr)
        if not ok:
            fails.append({"case": case, "stages": stages, "modes": modes, "scale": scale, "shape": [B, C, H, W], "fwd_err": e_f})
    print(json.dumps({"cases": args.cases, "seed": args.seed, "failed": len(fails), "failures": fails[:10], "worst_abs_err": worst,
                      "seconds": round(time.time() - t0, 1)}))
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
