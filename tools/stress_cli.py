#!/usr/bin/env python3
"""The CLI twin on Set5, over and over in one process (fresh engine, decode / encode threads each time); every summary line
must be the reference's.   python tools/stress_cli.py [--runs 100]"""
import argparse, contextlib, io, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mulut_amd import test_lut  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--runs", type=int, default=100)
ap.add_argument("--extra", default="", help="extra CLI arguments, space separated")
args = ap.parse_args()
G = os.path.join(ROOT, "tests", "golden")
bad = 0
with tempfile.TemporaryDirectory() as td:
    os.makedirs(os.path.join(td, "SRBenchmark", "Set5"))
    os.symlink(os.path.join(G, "Set5", "HR"), os.path.join(td, "SRBenchmark", "Set5", "HR"))
    os.symlink(os.path.join(G, "Set5", "LR_bicubic"), os.path.join(td, "SRBenchmark", "Set5", "LR_bicubic"))
    exp = os.path.join(td, "models", "sr_x2sdy")
    os.makedirs(exp)
    for fn in os.listdir(os.path.join(G, "luts")):
        os.symlink(os.path.join(G, "luts", fn), os.path.join(exp, fn))
    for run in range(args.runs):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            res = test_lut.main(["--stages", "2", "--modes", "sdy", "-e", exp, "--testDir", os.path.join(td, "SRBenchmark"),
                                 "--resultRoot", os.path.join(td, "results")] + args.extra.split())
        line = buf.getvalue().strip().splitlines()[-1]
        if not line.startswith("Dataset Set5 | AVG LUT PSNR: 30.61 SSIM: 0.865"):
            bad += 1
            print(run, line, [[round(float(v), 3) for v in r] for r in res["Set5"]])
print("runs", args.runs, "bad", bad)
sys.exit(1 if bad else 0)
