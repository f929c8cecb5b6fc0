"""Drop-in twin of the reference's per-pass callable ("Python callable 1").

``FourSimplexInterpFaster(weight, img_in, h, w, interval, rot, upscale=4, mode='s')`` has the
reference's signature, argument meaning, return type (float64 [C, ., .]) and error behaviour
(sr/4_test_lut.py:14-237), but runs on the GPU through ``mulut_pass``.  The reference function is
handed an already rotated + edge-padded image and rotates its result back by ``rot``; the device
kernel folds rotation and padding into its index maps, so this wrapper undoes the caller's
rot90/pad first (both are exact inverses; the pad rows/cols are checked to be replicas).
"""
import numpy as np
import torch

from .engine import MuLUTEngine

_ENGINES = {}
_PAD = {"s": 1, "d": 2, "y": 2}


def _engine(device):
    if device not in _ENGINES:
        _ENGINES[device] = MuLUTEngine(device)
    return _ENGINES[device]


def FourSimplexInterpFaster(weight, img_in, h, w, interval, rot, upscale=4, mode='s', device=0):
    if mode not in _PAD:
        raise ValueError("Mode {} not implemented.".format(mode))
    if interval != 4:
        raise NotImplementedError("only --interval 4 is supported (SURVEY.md quirk 3)")
    weight = np.asarray(weight)
    img_in = np.asarray(img_in)
    table = np.ascontiguousarray(weight.reshape(17 ** 4, upscale * upscale))
    q8 = table.astype(np.int8)
    if not np.array_equal(q8.astype(table.dtype), table):
        raise ValueError("weight must hold int8-valued entries (it is np.load(int8 LUT).astype(float32))")
    pad = _PAD[mode]
    core = img_in[:, :h, :w]
    if img_in.shape[1:] != (h + pad, w + pad) or not (
            np.array_equal(img_in[:, h:, :w], np.repeat(core[:, -1:, :], pad, 1)) and
            np.array_equal(img_in[:, :, w:], np.repeat(img_in[:, :, w - 1:w], pad, 2))):
        raise ValueError("img_in must be the bottom/right edge-padded image (np.pad(..., mode='edge'))")
    u8 = core.astype(np.uint8)
    if not np.array_equal(u8.astype(core.dtype), core):
        raise ValueError("img_in must hold integer values in 0..255")
    r = (4 - rot) % 4                                   # the driver passes rot = 4 - r (:297)
    unrot = np.rot90(u8, -r, axes=(1, 2)).copy(order="C")          # undo the driver's np.rot90(img, r)
    eng = _engine(device)
    # a throw-away 1-stage model whose only stage is this table (u = upscale)
    eng.configure(1, mode, scale=upscale, interval=interval)
    eng.set_lut(1, mode, q8)
    out_q = eng.pass_q(1, mode, r, torch.from_numpy(unrot).to(eng.device))
    return out_q.cpu().numpy().astype(np.float64) / (2 ** interval)
