"""Python binding of the C ABI (include/mulut.h): one MuLUTEngine per GPU.

torch is used only for device memory and streams; every compute call goes to libmulut_hip.so.
"""
import ctypes
import os

import numpy as np
import torch

from . import _native
from .lut_io import load_lut_dict

LAYOUT_CHW, LAYOUT_HWC = 0, 1


class MuLUTError(RuntimeError):
    pass


class MuLUTEngine:
    """Owns a ``mulut_ctx``.  Mirrors what the reference keeps in ``opt`` + ``lutDict``
    (sr/4_test_lut.py:320-333) and runs its stage loop (:279-306) on the GPU."""

    def __init__(self, device=0, lib_path=None):
        self._lib = _native.load(lib_path)
        if not torch.cuda.is_available():
            raise MuLUTError("no GPU visible: mulut_amd has no CPU path")
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        h = ctypes.c_void_p()
        self._check(self._lib.mulut_create(self.device.index, ctypes.byref(h)))
        self._h = h
        self.stages = self.modes = self.scale = None

    # -- plumbing ---------------------------------------------------------------------------
    def _check(self, rc):
        if rc == 0:
            return
        msg = self._lib.mulut_strerror(rc).decode()
        if rc == -2:
            raise ValueError(msg)                      # reference: ValueError("Mode {} not implemented.")
        if rc == -6 and getattr(self, "_h", None):
            msg += ": " + self._lib.mulut_last_hip_error(self._h).decode()
        raise MuLUTError("mulut error %d: %s" % (rc, msg))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mulut_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _need_config(self):
        if self.stages is None:
            self._check(-8)

    def _dev_u8(self, t, name):
        if not (isinstance(t, torch.Tensor) and t.dtype == torch.uint8 and t.is_cuda and t.is_contiguous()):
            raise TypeError("%s must be a contiguous uint8 CUDA tensor" % name)
        if t.device != self.device:
            raise ValueError("%s lives on %s, engine on %s" % (name, t.device, self.device))
        return t

    # -- model --------------------------------------------------------------------------------
    def configure(self, stages, modes, scale=4, interval=4):
        self._check(self._lib.mulut_configure(self._h, int(stages), str(modes).encode(), int(scale), int(interval)))
        self.stages, self.modes, self.scale, self.interval = int(stages), str(modes), int(scale), int(interval)
        # MULUT_TUNING="final_stage_kernel=5,hybrid_oob_per_1024=64": A/B and profiling runs of unmodified drivers
        for kv in filter(None, os.environ.get("MULUT_TUNING", "").split(",")):
            k, v = kv.split("=")
            self.set_tuning(k.strip(), int(v))
        return self

    def set_lut(self, stage, mode, table):
        t = np.ascontiguousarray(table)
        if t.dtype != np.int8:
            raise TypeError("LUT must be int8")
        t = t.reshape(17 ** 4, -1) if t.size % 17 ** 4 == 0 else t
        self._check(self._lib.mulut_set_lut(self._h, int(stage), mode.encode()[:1], t.ctypes.data, t.shape[0],
                                            int(t.shape[1]) if t.ndim == 2 else 0))

    def set_lut_dict(self, lut_dict):
        for key, table in lut_dict.items():
            stage, mode = key[1:].split("_")
            self.set_lut(int(stage), mode, table)
        return self

    def load_luts(self, exp_dir, lut_name="LUT_ft"):
        """The LUT-loading block of sr/4_test_lut.py:322-333."""
        return self.set_lut_dict(load_lut_dict(exp_dir, self.stages, self.modes, self.scale, self.interval, lut_name))

    @property
    def halo(self):
        return self._lib.mulut_halo(self._h)

    def kernel_name(self, final=True):
        return self._lib.mulut_kernel_name(self._h, int(final)).decode()

    # -- compute ------------------------------------------------------------------------------
    def pass_q(self, stage, mode, r, img_chw):
        """q * FourSimplexInterpFaster(...) for driver rotation r on a planar uint8 image -> int32 CHW."""
        self._need_config()
        x = self._dev_u8(img_chw, "img_chw")
        C, H, W = x.shape
        u = self.scale if stage == self.stages else 1
        out = torch.empty((C, H * u, W * u), dtype=torch.int32, device=self.device)
        self._check(self._lib.mulut_pass(self._h, int(stage), mode.encode()[:1], int(r), x.data_ptr(), H, W, C,
                                         out.data_ptr(), self._stream()))
        return out

    @staticmethod
    def _dims(x, layout):
        if x.dim() == 3:
            x = x.unsqueeze(0)
        if x.dim() != 4:
            raise ValueError("expected [H,W,C]/[N,H,W,C] (HWC) or [C,H,W]/[N,C,H,W] (CHW)")
        if layout == LAYOUT_HWC:
            N, H, W, C = x.shape
        else:
            N, C, H, W = x.shape
        return x, N, H, W, C

    def _out(self, N, H, W, C, layout, squeeze):
        shape = (N, H, W, C) if layout == LAYOUT_HWC else (N, C, H, W)
        out = torch.empty(shape, dtype=torch.uint8, device=self.device)
        return out, (out[0] if squeeze else out)

    def stage(self, stage, x, layout=LAYOUT_HWC, out_layout=None):
        """One stage (all modes x 4 rotations + combine): sr/4_test_lut.py:280-306."""
        self._need_config()
        out_layout = layout if out_layout is None else out_layout
        squeeze = x.dim() == 3
        x4, N, H, W, C = self._dims(self._dev_u8(x, "x"), layout)
        u = self.scale if stage == self.stages else 1
        out, ret = self._out(N, H * u, W * u, C, out_layout, squeeze)
        self._check(self._lib.mulut_stage(self._h, int(stage), x4.data_ptr(), layout, out.data_ptr(), out_layout, N, H,
                                          W, C, self._stream()))
        return ret

    def pipeline(self, x, layout=LAYOUT_HWC, out=None):
        """The whole cascade, sr/4_test_lut.py:279-306: uint8 in -> uint8 out (scale x larger)."""
        self._need_config()
        squeeze = x.dim() == 3
        x4, N, H, W, C = self._dims(self._dev_u8(x, "x"), layout)
        if out is None:
            out, ret = self._out(N, H * self.scale, W * self.scale, C, layout, squeeze)
        else:
            ret = self._dev_u8(out, "out")
            if out.numel() != N * H * W * C * self.scale ** 2:
                raise ValueError("out has the wrong size")
        self._check(self._lib.mulut_pipeline(self._h, x4.data_ptr(), out.data_ptr(), N, H, W, C, layout,
                                             self._stream()))
        return ret

    def pipeline_rows(self, band, band_row0, y0, y1, H_full, layout=LAYOUT_HWC):
        """Strip form: `band` holds rows [band_row0, band_row0+rows) of H_full-row images; returns the
        output for LR rows [y0, y1)."""
        self._need_config()
        squeeze = band.dim() == 3
        b4, N, rows, W, C = self._dims(self._dev_u8(band, "band"), layout)
        out, ret = self._out(N, (y1 - y0) * self.scale, W * self.scale, C, layout, squeeze)
        self._check(self._lib.mulut_pipeline_rows(self._h, b4.data_ptr(), int(band_row0), rows, out.data_ptr(), int(y0),
                                                  int(y1), N, int(H_full), W, C, layout, self._stream()))
        return ret

    def reserve(self, N, H, W, C):
        self._check(self._lib.mulut_reserve(self._h, N, H, W, C))

    def set_stage_timing(self, enable=True):
        self._check(self._lib.mulut_set_stage_timing(self._h, int(bool(enable))))

    def last_stage_ms(self):
        """Device milliseconds of each stage of the last pipeline call (needs set_stage_timing(True))."""
        buf = (ctypes.c_float * 8)()
        n = self._lib.mulut_last_stage_ms(self._h, buf, 8)
        if n < 0:
            self._check(n)
        return [float(buf[k]) for k in range(n)]

    def last_kernel_ms(self):
        """Device milliseconds of each stage's dominant kernel alone in the last pipeline call (set_stage_timing(True))."""
        buf = (ctypes.c_float * 8)()
        n = self._lib.mulut_last_kernel_ms(self._h, buf, 8)
        if n < 0:
            self._check(n)
        return [float(buf[k]) for k in range(n)]

    def last_detail_counters(self):
        """Work counters of the detailed-tile path of the last final-stage launch: dict with the samples per anchor MSB that
        went through the anchor-slab kernel, the number of work items and the length of the fix-up list (entries: samples)."""
        buf = (ctypes.c_uint32 * 32)()
        n = self._lib.mulut_last_detail_counters(self._h, buf, 32, self._stream())
        if n < 0:
            self._check(n)
        v = [int(buf[k]) for k in range(n)]
        if n < 18:
            return {"samples_per_anchor": [], "items": 0, "fix_pixels": 0, "probe": []}
        return {"samples_per_anchor": v[:16], "items": v[16], "fix_pixels": v[17], "probe": v[18:]}

    def debug_read(self, words=64, reset=True):
        """The first `words` 64-bit words of the context's probe buffer (written by probe builds of the kernels only)."""
        buf = (ctypes.c_uint64 * words)()
        n = self._lib.mulut_debug_read(self._h, buf, words, 1 if reset else 0, self._stream())
        if n < 0:
            self._check(n)
        return [int(buf[k]) for k in range(n)]

    def eval_y(self, gt_hwc, out_hwc, shave):
        """(PSNR, SSIM) on the Y channel of two device uint8 HWC RGB images, as sr/4_test_lut.py:313-315 scores a result
        (common/utils.py:42-101) -- computed on the device, only two doubles come back."""
        gt = self._dev_u8(gt_hwc, "gt")
        out = self._dev_u8(out_hwc, "out")
        if gt.dim() != 3 or gt.shape[2] != 3 or gt.shape != out.shape:
            raise ValueError("eval_y wants two HWC RGB images of one shape, got %s and %s" % (tuple(gt.shape), tuple(out.shape)))
        H, W = int(gt.shape[0]), int(gt.shape[1])
        n = int(self._lib.mulut_eval_ws_doubles(H, W))
        ws = torch.empty(n, dtype=torch.float64, device=gt.device)
        psnr, ssim = ctypes.c_double(), ctypes.c_double()
        self._check(self._lib.mulut_eval_y(self.device.index, gt.data_ptr(), out.data_ptr(), H, W, int(shave), ws.data_ptr(), n,
                                           ctypes.byref(psnr), ctypes.byref(ssim), self._stream()))
        return psnr.value, ssim.value

    def set_tuning(self, key, value):
        """Performance knobs that never change results, e.g. ("final_stage_kernel", 1)."""
        self._check(self._lib.mulut_set_tuning(self._h, key.encode(), int(value)))
        return self
