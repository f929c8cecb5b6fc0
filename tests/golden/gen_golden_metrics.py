#!/usr/bin/env python3
"""Golden values for the evaluation helpers, produced by RUNNING THE REFERENCE's `common/utils.py`
(`_rgb2ycbcr`, `PSNR`, `cal_ssim`, `modcrop`) in this container -- same rules as gen_golden.py (cv2 stub, refuses to
run without /root/reference, writes data only).

    python tests/golden/gen_golden_metrics.py     # rewrites tests/golden/metrics_fixtures.npz

Cases: the five Set5 outputs of the reference against their HR images (shave 4, as sr/4_test_lut.py:313-315), and
seeded random RGB pairs of odd sizes with shave 0..4.  Stored per case: both uint8 images (random cases only),
shave, PSNR, SSIM.
"""
import os
import sys
import types

import numpy as np
from PIL import Image

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    if not os.path.isdir(REF):
        raise SystemExit("gen_golden_metrics.py: /root/reference not present")
    cv2 = types.ModuleType("cv2")

    def getGaussianKernel(ksize, sigma):
        i = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2.0
        k = np.exp(-(i * i) / (2.0 * sigma * sigma))
        return (k / k.sum()).reshape(-1, 1)

    cv2.getGaussianKernel = getGaussianKernel
    sys.modules.setdefault("cv2", cv2)
    sys.path.insert(0, REF)
    from common.utils import PSNR, _rgb2ycbcr, cal_ssim, modcrop

    def score(gt, out, shave):
        y_gt, y_out = _rgb2ycbcr(gt)[:, :, 0], _rgb2ycbcr(out)[:, :, 0]
        return float(PSNR(y_gt, y_out, shave)), float(cal_ssim(y_gt, y_out))

    fx = {}
    for fn in sorted(os.listdir(os.path.join(HERE, "Set5", "ref_out"))):
        stem = fn.split("_LUT_ft")[0]
        out = np.array(Image.open(os.path.join(HERE, "Set5", "ref_out", fn)))
        gt = modcrop(np.array(Image.open(os.path.join(HERE, "Set5", "HR", stem + ".png"))), 4)
        fx["set5/%s" % stem] = np.array(score(gt, out, 4))
    rng = np.random.default_rng(7)
    for k, (h, w, shave) in enumerate(((37, 23, 2), (64, 48, 4), (11, 11, 0), (45, 130, 3), (12, 40, 1))):
        gt = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        noise = rng.normal(0, 6 + 5 * k, (h, w, 3))
        out = np.clip(np.round(gt + noise), 0, 255).astype(np.uint8)
        name = "rand/%dx%d_s%d" % (h, w, shave)
        fx[name + "/gt"], fx[name + "/out"] = gt, out
        fx[name + "/score"] = np.array(score(gt, out, shave))
    np.savez_compressed(os.path.join(HERE, "metrics_fixtures.npz"), **fx)
    for k, v in fx.items():
        if k.endswith("score") or k.startswith("set5"):
            print(k, v)


if __name__ == "__main__":
    main()
