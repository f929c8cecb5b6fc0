#!/usr/bin/env python3
"""Per-wave time split of the phased final-stage kernel (build with -DMULUT_PROFILE: each wave writes its s_memtime
totals behind the last output frame).  Prints, per content type, the mean share of tile load / barrier wait /
phase compute / epilogue and the number of slow (out-of-band) pairs per wave."""
import os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mulut_amd import MuLUTEngine, _native, load_lut_dict
from mulut_amd.synth import natural_frames, noise_frames, real_frames

so = os.path.join(ROOT, "build", "variants", "libmulut_profile.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
srcs = [os.path.join(_native._CSRC, f) for f in _native.SOURCES]
extra = [a for a in sys.argv[1:] if a.startswith("-D")]
subprocess.check_call([_native._hipcc()] + _native.HIPCC_FLAGS + ["-DMULUT_PROFILE=1"] + extra + ["-o", so] + srcs)
if "--build-only" in sys.argv:
    sys.exit(0)
luts = load_lut_dict(os.path.join(ROOT, "tests", "golden", "luts"), 2, "sdy", 4, 4, "LUT_ft")
e = MuLUTEngine(0, lib_path=so).configure(2, "sdy", 4, 4).set_lut_dict(luts)
variant = 0
e.set_tuning("final_stage_kernel", 3)
N, H, W = 4, 1080, 1920
png = os.path.join(ROOT, "tests", "golden", "DIV2K_LR_X4", "0001x4.png")
out = torch.zeros((N + 1, H * 4, W * 4, 3), dtype=torch.uint8, device="cuda")
for name, fr in (("natural", natural_frames(2, H, W, 3, 0)), ("real", real_frames(2, H, W, png, 0)), ("noise", noise_frames(2, H, W, 3, 0))):
    x = torch.from_numpy(fr).cuda().repeat(2, 1, 1, 1).contiguous()
    for _ in range(2):
        e.pipeline(x, out=out[:N])
    torch.cuda.synchronize()
    raw = out[N].reshape(-1)[:256 * 16 * 64].cpu().numpy().view(np.uint64).reshape(256, 16, 8).astype(np.float64)
    tot = raw[..., 0]
    if variant:
        print("%-8s total ticks/wave %.3g | load %.1f%%  barrier %.1f%%  compute %.1f%%  epilogue %.1f%%  | slow phases/wave %.1f"
              % (name, tot.mean(), 100 * (raw[..., 1] / tot).mean(), 100 * (raw[..., 2] / tot).mean(), 100 * (raw[..., 3] / tot).mean(),
                 100 * (raw[..., 4] / tot).mean(), raw[..., 5].mean()))
    else:
        nf, ns = raw[..., 7], raw[..., 5]
        print("%-8s ticks/wave %.3g | load %.1f%%  barrier %.1f%%  fast pairs %.1f%%  slow pairs %.1f%%  epilogue %.1f%% | pairs/wave fast %.0f slow %.0f (%.2f%%) | ticks per pair: fast %.0f  slow %.0f"
              % (name, tot.mean(), 100 * (raw[..., 1] / tot).mean(), 100 * (raw[..., 2] / tot).mean(), 100 * (raw[..., 3] / tot).mean(),
                 100 * (raw[..., 6] / tot).mean(), 100 * (raw[..., 4] / tot).mean(), nf.mean(), ns.mean(), 100 * ns.sum() / (nf.sum() + ns.sum()),
                 raw[..., 3].sum() / max(nf.sum(), 1), raw[..., 6].sum() / max(ns.sum(), 1)))
