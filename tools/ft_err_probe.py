#!/usr/bin/env python3
"""How far are the fine-tune gradients of the HIP path from the CPU oracle on the big batches?  (GPU box)
Prints, per tensor, max |diff|, max |ref| and the largest element-wise relative error among elements above 1 % of max |ref|."""
import os, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mulut_amd.finetune import MuLUT
from mulut_amd.synth import natural_frames
from oracle import ft_torch


def synthetic_lut(seed, vnum):      # as tests/test_gpu_finetune.py
    rng = np.random.default_rng(seed)
    return rng.integers(-127, 128, size=(17 ** 4, vnum), dtype=np.int8)


for shape, kind in (((256, 1, 48, 48), "smooth"), ((16, 1, 48, 48), "u8"), ((256, 1, 48, 48), "u8")):
    stages, modes, scale = 2, "sdy", 4
    rng = np.random.default_rng(241)
    tmp = tempfile.mkdtemp()
    tabs = {}
    for s in range(stages):
        vnum = scale * scale if s + 1 == stages else 1
        for m in modes:
            t = synthetic_lut(3 * s + ord(m), vnum)
            tabs["s%d_%s" % (s + 1, m)] = t
            np.save(os.path.join(tmp, "LUT_x%d_4bit_int8_s%d_%s.npy" % (scale, s + 1, m)), t)
    if kind == "smooth":
        big = natural_frames(1, 1080, 1920, 1, 11)[0, :, :, 0]
        ys, xs = rng.integers(0, 1080 - shape[2], shape[0]), rng.integers(0, 1920 - shape[3], shape[0])
        x = np.stack([big[a:a + shape[2], b:b + shape[3]] for a, b in zip(ys, xs)])[:, None].astype(np.float32) / np.float32(255)
    else:
        x = rng.integers(0, 256, shape).astype(np.float32) / np.float32(255)
    tgt = rng.random((shape[0], shape[1], shape[2] * scale, shape[3] * scale), dtype=np.float32)
    wcpu = {k: torch.from_numpy(v.astype(np.float32) / 127.0).requires_grad_(True) for k, v in tabs.items()}
    xc = torch.from_numpy(x).requires_grad_(True)
    yc = ft_torch.forward(wcpu, xc, stages, modes, scale)
    torch.nn.functional.mse_loss(yc, torch.from_numpy(tgt)).backward()
    # float64 oracle as the yardstick for both
    w64 = {k: torch.from_numpy(v.astype(np.float64) / 127.0).requires_grad_(True) for k, v in tabs.items()}
    net = MuLUT(tmp, stages, modes, upscale=scale, interval=4).cuda()
    xg = torch.from_numpy(x).cuda().requires_grad_(True)
    yg = net(xg)
    torch.nn.functional.mse_loss(yg, torch.from_numpy(tgt).cuda()).backward()
    def stat(name, g, r):
        d = np.abs(g - r); m = np.abs(r).max()
        sig = np.abs(r) > 0.01 * m
        rel = (d[sig] / np.abs(r[sig])).max() if sig.any() else 0.0
        print("%s %-8s max|diff| %.3e  max|ref| %.3e  ratio %.2e  max rel (|ref| > 1%% of max) %.2e" % (shape, name, d.max(), m, d.max() / m, rel))
    stat("gx", xg.grad.cpu().numpy(), xc.grad.numpy())
    for k, w in wcpu.items():
        stat(k, getattr(net, "weight_" + k).grad.cpu().numpy(), w.grad.numpy())
