#!/usr/bin/env python3
"""A/B timing of kernel variants in ONE process, interleaved rounds (cdna guide rule 24).

    python tools/ab_bench.py --variants base,a1,a2,a3,a11,a12,a13 [--frames 4] [--rounds 7]

`base` is the shipped library; `aN` is a timing-only build with -DMULUT_ABLATE=N (results are wrong
by design); any other name `x` maps to build/variants/libmulut_x.so built from -DMULUT_VARIANT_x.
Prints median / min device milliseconds per stage and distribution as JSON lines.
"""
import argparse
import json
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mulut_amd import MuLUTEngine, _native, load_lut_dict  # noqa: E402
from mulut_amd.synth import natural_frames, noise_frames, real_frames  # noqa: E402


COMPILER_VARIANTS = {
    "ilp": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
    "mclause": ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"],
    "itilp": ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"],
    "itminreg": ["-mllvm", "-amdgpu-sched-strategy=iterative-minreg"],
    "nosched": ["-mllvm", "-enable-misched=0"],
    "o2": ["-O2"],
    "ftabl1": ["-DMULUT_FT_ABL=1"], "ftabl2": ["-DMULUT_FT_ABL=2"], "ftabl3": ["-DMULUT_FT_ABL=3"], "ftabl4": ["-DMULUT_FT_ABL=4"], "ftabl5": ["-DMULUT_FT_ABL=5"], "ftabl6": ["-DMULUT_FT_ABL=6"], "ftabl7": ["-DMULUT_FT_ABL=7"],      # ft_stage_bwd4 timing ablations
    "ftnt512": ["-DMULUT_FT_B4_SITES=512"], "ftnt768": ["-DMULUT_FT_B4_SITES=768"],
}


def build_variant(name):
    name = name.split("@")[0].split("+")[0]
    if name == "base":
        return _native.build()
    out_dir = os.path.join(ROOT, "build", "variants")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libmulut_%s.so" % name)
    flag = "-DMULUT_ABLATE=%s" % name[1:] if name[0] == "a" and name[1:].isdigit() else "-DMULUT_VARIANT_%s=1" % name
    if name in COMPILER_VARIANTS:        # same source, other compiler options
        flag = COMPILER_VARIANTS[name]
    srcs = [os.path.join(_native._CSRC, f) for f in _native.SOURCES]
    newest = max(os.path.getmtime(os.path.join(_native._CSRC, f)) for f in _native.SOURCES + _native.HEADERS)
    if not os.path.exists(so) or os.path.getmtime(so) < newest:
        subprocess.check_call([_native._hipcc()] + _native.HIPCC_FLAGS + (flag if isinstance(flag, list) else [flag]) + ["-o", so] + srcs)
    return so


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="base")
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--build-only", action="store_true")
    args = ap.parse_args()
    names = args.variants.split(",")
    libs = {n: build_variant(n) for n in names}
    if args.build_only:
        print("built", libs)
        return
    luts = load_lut_dict(os.path.join(ROOT, "tests", "golden", "luts"), 2, "sdy", 4, 4, "LUT_ft")
    engines = {}
    for n in names:
        e = MuLUTEngine(0, lib_path=libs[n]).configure(2, "sdy", 4, 4).set_lut_dict(luts)
        for kv in n.split("+")[1:]:       # name[@...]+key=value: any mulut_set_tuning key
            e.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
        if "@" in n:                      # name@final[:threshold][/first]
            sel = n.split("+")[0].split("@")[1]
            if "/" in sel:
                sel, fst = sel.split("/")
                e.set_tuning("first_stage_kernel", int(fst))
            e.set_tuning("final_stage_kernel", int(sel.split(":")[0]))
            if ":" in sel:
                e.set_tuning("hybrid_oob_per_1024", int(sel.split(":")[1]))
        e.reserve(args.frames, args.h, args.w, 3)
        e.set_stage_timing(True)
        engines[n] = e
    data = {"natural": torch.from_numpy(natural_frames(min(args.frames, 2), args.h, args.w, 3, 0)).cuda(),
            "noise": torch.from_numpy(noise_frames(min(args.frames, 2), args.h, args.w, 3, 0)).cuda()}
    real_png = os.path.join(ROOT, "tests", "golden", "DIV2K_LR_X4", "0001x4.png")
    if os.path.exists(real_png):
        data["real"] = torch.from_numpy(real_frames(min(args.frames, 2), args.h, args.w, real_png, 0)).cuda()
    for k in data:
        data[k] = data[k].repeat((args.frames + 1) // data[k].shape[0], 1, 1, 1)[:args.frames].contiguous()
    out = torch.empty((args.frames, args.h * 4, args.w * 4, 3), dtype=torch.uint8, device="cuda")
    res = {(n, d): [] for n in names for d in data}
    for rnd in range(args.rounds + 1):
        for n in names:
            for d, x in data.items():
                engines[n].pipeline(x, out=out)
                ms = engines[n].last_stage_ms()
                if rnd:                     # round 0 = warm-up
                    res[(n, d)].append(ms)
                if rnd == args.rounds and "clk" in n:     # probe builds: phase clocks of the anchor-slab kernel
                    print(json.dumps({"variant": n, "dist": d, "detail": engines[n].last_detail_counters()}))
    for n in names:
        for d in data:
            a = np.asarray(res[(n, d)]) / args.frames * 1e3     # us per frame
            print(json.dumps({"variant": n, "dist": d, "us_per_frame_median": [round(float(v), 1) for v in np.median(a, 0)],
                              "us_per_frame_min": [round(float(v), 1) for v in a.min(0)]}))


if __name__ == "__main__":
    main()
