"""Evaluation helpers of the reference's test script, restated (common/utils.py:28-101).
Evaluation only -- not on the hot path; NumPy on the host."""
import numpy as np
from scipy import signal


def modcrop(image, modulo):
    """common/utils.py:28-39"""
    if image.ndim == 2:
        h, w = image.shape
        return image[:h - h % modulo, :w - w % modulo]
    if image.shape[2] == 3:
        h, w = image.shape[:2]
        return image[:h - h % modulo, :w - w % modulo, :]
    raise NotImplementedError


_T = np.array([[0.256788235294118, 0.504129411764706, 0.097905882352941],
               [-0.148223529411765, -0.290992156862745, 0.439215686274510],
               [0.439215686274510, -0.367788235294118, -0.071427450980392]])
_O = np.array([16.0, 128.0, 128.0])


def rgb2ycbcr(img, max_val=255):
    """common/utils.py:42-60 (BT.601 studio swing)."""
    off = _O / 255.0 if max_val == 1 else _O
    flat = np.reshape(img, (-1, img.shape[2])).astype(np.float64)
    # three scaled columns added up, NOT `flat @ _T.T`: the CLI scores images on several threads at once, and concurrent
    # dgemm calls of the bundled OpenBLAS (0.3.29, threaded) on these tall 3-column operands were seen to return wrong rows
    # now and then (tools/stress_cli.py: 19 of 300 Set5 runs printed a PSNR off by 0.03-0.6 dB with bit-exact PNGs)
    out = flat[:, 0:1] * _T[:, 0] + flat[:, 1:2] * _T[:, 1] + flat[:, 2:3] * _T[:, 2] + off
    return out.reshape(img.shape)


def psnr(y_true, y_pred, shave_border=4):
    """common/utils.py:63-72 (float32 difference, border shaved)."""
    diff = np.array(y_pred, dtype=np.float32) - np.array(y_true, dtype=np.float32)
    if shave_border > 0:
        diff = diff[shave_border:-shave_border, shave_border:-shave_border]
    rmse = np.sqrt(np.mean(np.power(diff, 2)))
    return 20 * np.log10(255.0 / rmse)


def _gaussian_kernel(ksize=11, sigma=1.5):
    # what cv2.getGaussianKernel(11, 1.5) returns (common/utils.py:78); cv2 is not a dependency here
    i = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2.0
    k = np.exp(-(i * i) / (2.0 * sigma * sigma))
    return (k / k.sum()).reshape(-1, 1)


def ssim(img1, img2):
    """common/utils.py:75-101 (11x11 Gaussian, sigma 1.5, 'valid' windows)."""
    k1, k2, L = 0.01, 0.03, 255
    kx = _gaussian_kernel()
    window = kx * kx.T
    c1, c2 = (k1 * L) ** 2, (k2 * L) ** 2
    a, b = np.float64(img1), np.float64(img2)
    conv = lambda z: signal.convolve2d(z, window, "valid")  # noqa: E731
    mu1, mu2 = conv(a), conv(b)
    s1 = conv(a * a) - mu1 * mu1
    s2 = conv(b * b) - mu2 * mu2
    s12 = conv(a * b) - mu1 * mu2
    m = ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s1 + s2 + c2))
    return float(np.mean(m))
