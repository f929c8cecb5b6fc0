"""ctypes front-end of oracle/libmulut_oracle.so (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

Each function cites the reference lines it restates; images are numpy uint8 arrays.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libmulut_oracle.so")
    src = os.path.join(_HERE, "mulut_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libmulut_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        i, p, c = ctypes.c_int, ctypes.c_void_p, ctypes.c_char
        L.mulut_oracle_pass.argtypes = [p, p, i, i, i, i, i, c, i, p, i]
        L.mulut_oracle_pass.restype = i
        L.mulut_oracle_stage.argtypes = [p, ctypes.c_char_p, i, i, p, i, i, i, i, i, p]
        L.mulut_oracle_stage.restype = i
        L.mulut_oracle_pipeline.argtypes = [p, i, ctypes.c_char_p, i, i, i, p, i, i, i, p]
        L.mulut_oracle_pipeline.restype = i
        _LIB = L
    return _LIB


def _tables(luts):
    keep = [np.ascontiguousarray(t, dtype=np.int8) for t in luts]
    arr = (ctypes.c_void_p * len(keep))(*[t.ctypes.data for t in keep])
    return keep, arr


def _check(rc):
    if rc == -1:
        raise ValueError("Mode not implemented.")     # sr/4_test_lut.py:54
    if rc:
        raise RuntimeError("mulut_oracle error %d" % rc)


def pass_q(lut, img_chw, r, upscale, mode, interval=4):
    """q*FourSimplexInterpFaster(...) (sr/4_test_lut.py:14-237) for driver rotation r, as int32 CHW."""
    lut = np.ascontiguousarray(lut, dtype=np.int8)
    img = np.ascontiguousarray(img_chw, dtype=np.uint8)
    C, H, W = img.shape
    assert lut.size == (2 ** (8 - interval) + 1) ** 4 * upscale * upscale
    out = np.empty((C, H * upscale, W * upscale), dtype=np.int32)
    _check(lib().mulut_oracle_pass(lut.ctypes.data, img.ctypes.data, H, W, C, interval, upscale,
                                   mode.encode()[:1], r, out.ctypes.data, 0))
    return out


def stage(luts, modes, is_last, img_hwc, upscale, interval=4):
    """One stage of sr/4_test_lut.py:279-306 on an HWC uint8 image -> HWC uint8."""
    img = np.ascontiguousarray(np.asarray(img_hwc, dtype=np.uint8).transpose(2, 0, 1))
    C, H, W = img.shape
    keep, arr = _tables(luts)
    out = np.empty((C, H * upscale, W * upscale), dtype=np.uint8)
    _check(lib().mulut_oracle_stage(arr, modes.encode(), len(modes), int(is_last), img.ctypes.data, H, W, C,
                                    interval, upscale, out.ctypes.data))
    return np.ascontiguousarray(out.transpose(1, 2, 0))


def pipeline(lut_dict, stages, modes, scale, img_hwc, interval=4):
    """The whole cascade (sr/4_test_lut.py:279-306); lut_dict keys 's{stage}_{mode}' as in :330."""
    img = np.ascontiguousarray(np.asarray(img_hwc, dtype=np.uint8).transpose(2, 0, 1))
    C, H, W = img.shape
    keep, arr = _tables([lut_dict["s%d_%s" % (s + 1, m)] for s in range(stages) for m in modes])
    out = np.empty((C, H * scale, W * scale), dtype=np.uint8)
    _check(lib().mulut_oracle_pipeline(arr, stages, modes.encode(), len(modes), scale, interval, img.ctypes.data,
                                       H, W, C, out.ctypes.data))
    return np.ascontiguousarray(out.transpose(1, 2, 0))
