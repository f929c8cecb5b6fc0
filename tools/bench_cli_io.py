#!/usr/bin/env python3
"""End-to-end images/s of the CLI twin (PNG decode -> H2D -> cascade -> D2H -> PNG encode + Y-PSNR/SSIM), serial vs with the
host I/O overlapped (eltr.run num_worker), on the Set5 pairs and on synthetic 1080p LR frames written as PNG.
    python tools/bench_cli_io.py [--frames 6]"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mulut_amd.options import TestOptions  # noqa: E402
from mulut_amd.synth import natural_frames  # noqa: E402
from mulut_amd.test_lut import build_engine, eltr  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=6)
    args = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix="mulut_io_")
    try:
        root = os.path.join(tmp, "bench")
        shutil.copytree(os.path.join(ROOT, "tests", "golden", "Set5"), os.path.join(root, "Set5"))
        hr_dir, lr_dir = os.path.join(root, "Syn1080", "HR"), os.path.join(root, "Syn1080", "LR_bicubic", "X4")
        os.makedirs(hr_dir)
        os.makedirs(lr_dir)
        lr = natural_frames(args.frames, 1080, 1920, 3, 0)
        for i in range(args.frames):
            Image.fromarray(lr[i]).save(os.path.join(lr_dir, "f%02d.png" % i))
            Image.fromarray(np.repeat(np.repeat(lr[i], 4, 0), 4, 1)).save(os.path.join(hr_dir, "f%02d.png" % i), compress_level=1)
        exp = os.path.join(tmp, "sr_x2sdy")
        shutil.copytree(os.path.join(ROOT, "tests", "golden", "luts"), exp)
        opt = TestOptions().parse(["--stages", "2", "--modes", "sdy", "-e", exp, "--testDir", root, "--resultRoot", os.path.join(tmp, "res"),
                                   "--deviceMetrics"])
        eng = build_engine(opt)
        eltr("Set5", opt, eng).run(1)                      # warm-up (library load, first launches)
        for ds in ("Set5", "Syn1080"):
            for nw in (1, 4, 8):
                ev = eltr(ds, opt, eng)
                ev.run(nw)
                print(json.dumps({"dataset": ds, "images": len(ev.files), "io_threads": nw, "seconds": round(ev.seconds, 3),
                                  "images_per_s": round(len(ev.files) / ev.seconds, 2)}))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
