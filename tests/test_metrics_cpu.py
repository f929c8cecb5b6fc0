"""Host-side evaluation helpers (mulut_amd/metrics.py) against values produced by the reference's common/utils.py
(tests/golden/metrics_fixtures.npz, written by tests/golden/gen_golden_metrics.py)."""
import os

import numpy as np
import pytest
from PIL import Image

from conftest import GOLDEN

from mulut_amd.metrics import modcrop, psnr, rgb2ycbcr, ssim


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(GOLDEN, "metrics_fixtures.npz"))


def _score(gt, out, shave):
    y_gt, y_out = rgb2ycbcr(gt)[:, :, 0], rgb2ycbcr(out)[:, :, 0]
    return psnr(y_gt, y_out, shave), ssim(y_gt, y_out)


def test_random_pairs_match_reference(fx):
    names = sorted({k.rsplit("/", 1)[0] for k in fx.files if k.startswith("rand/")})
    assert len(names) == 5
    for n in names:
        shave = int(n.rsplit("_s", 1)[1])
        p, s = _score(fx[n + "/gt"], fx[n + "/out"], shave)
        assert p == pytest.approx(fx[n + "/score"][0], abs=1e-5), n
        assert s == pytest.approx(fx[n + "/score"][1], abs=1e-12), n


def test_set5_scores_match_reference(fx):
    vals = []
    for k in sorted(k for k in fx.files if k.startswith("set5/")):
        stem = k.split("/")[1]
        out = np.array(Image.open(os.path.join(GOLDEN, "Set5", "ref_out", stem + "_LUT_ft_4bit.png")))
        gt = modcrop(np.array(Image.open(os.path.join(GOLDEN, "Set5", "HR", stem + ".png"))), 4)
        p, s = _score(gt, out, 4)
        assert p == pytest.approx(fx[k][0], abs=1e-5) and s == pytest.approx(fx[k][1], abs=1e-12), stem
        vals.append((p, s))
    m = np.mean(vals, axis=0)
    assert "{:.2f} {:.4f}".format(*m) == "30.61 0.8656"       # sr/4_test_lut.py:260 summary line


def test_scores_do_not_depend_on_concurrent_callers(fx):
    """The CLI scores finished images on several threads at once (eltr.run): the same five Set5 pairs scored by four threads,
    40 rounds, must give the single-threaded numbers every time (a BLAS matmul inside rgb2ycbcr did not)."""
    from concurrent.futures import ThreadPoolExecutor
    pairs = []
    for k in sorted(k for k in fx.files if k.startswith("set5/")):
        stem = k.split("/")[1]
        out = np.array(Image.open(os.path.join(GOLDEN, "Set5", "ref_out", stem + "_LUT_ft_4bit.png")))
        pairs.append((modcrop(np.array(Image.open(os.path.join(GOLDEN, "Set5", "HR", stem + ".png"))), 4), out))
    want = [_score(gt, out, 4) for gt, out in pairs]
    with ThreadPoolExecutor(4) as ex:
        for _ in range(40):
            assert list(ex.map(lambda p: _score(p[0], p[1], 4), pairs)) == want
