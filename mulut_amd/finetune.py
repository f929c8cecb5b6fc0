"""GPU twin of the reference's differentiable LUT module (SURVEY.md 8b "Python callable 2").

``MuLUT(lut_folder, stages, modes, upscale=4, interval=4)`` keeps the reference's constructor, parameter
names (``weight_s{stage}_{mode}``, float32 [83521, u*u] = int8/127, sr/model.py:49-57) and forward contract
(x float32 [B,C,H,W] in 0..1 -> [B,C,H*u,W*u] in 0..1, :289-312), so ``sr/3_finetune_lut.py`` can train it with
the same Adam / cosine schedule and write ``LUT_ft_*.npy`` the same way (:162-169).  Each stage runs as one
forward and one backward HIP kernel (mulut_amd/csrc/mulut_ft.hip) through the C ABI; torch provides autograd
plumbing, parameters and the optimiser only.
"""
import ctypes
import os

import numpy as np
import torch
import torch.nn as nn

from . import _native


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


class _StageFn(torch.autograd.Function):
    """One stage: all modes x 4 rotations, per-pass BPDA rounding, clamp/round of the stage output."""

    @staticmethod
    def forward(ctx, x, modes, is_last, u, *weights):
        lib = _native.load()
        x = x.contiguous()
        stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        # weight = clamp(round_func(weight * 127), -127, 127)       sr/model.py:74-76, the stage's tables in one launch
        ws = [w.detach().contiguous() for w in weights]
        n = ws[0].numel()
        if any(w.numel() != n or w.dtype != torch.float32 for w in ws):
            raise ValueError("the tables of a stage must be float32 of one shape")
        wq_all = torch.empty((len(ws), n), dtype=torch.float32, device=x.device)
        wq = [wq_all[m].view(w.shape) for m, w in enumerate(ws)]
        rc = lib.mulut_ft_quantize(x.device.index, _ptr_array(ws), _ptr_array(wq), len(ws), n, stream)
        if rc:
            raise RuntimeError(lib.mulut_strerror(rc).decode())
        B, C, H, W = x.shape
        out = torch.empty((B, C, H * u, W * u), dtype=torch.float32, device=x.device)
        # where the stage's clamp passes gradient, 16 bits per site: saves the backward a recomputation of the stage forward
        inside = torch.empty((B, C, H, W), dtype=torch.int16, device=x.device)
        rc = lib.mulut_ft_stage_forward_mask(x.device.index, _ptr_array(wq), modes.encode(), int(is_last), int(u), x.data_ptr(),
                                             B, C, H, W, out.data_ptr(), inside.data_ptr(), stream)
        if rc:
            raise (ValueError if rc == -2 else RuntimeError)(lib.mulut_strerror(rc).decode())
        ctx.save_for_backward(x, wq_all, inside, *ws)
        ctx.cfg = (modes, is_last, u)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _native.load()
        modes, is_last, u = ctx.cfg
        x, wq_all, inside, *ws = ctx.saved_tensors
        wq = [wq_all[m].view(w.shape) for m, w in enumerate(ws)]
        gout = gout.contiguous()
        B, C, H, W = x.shape
        g_all = torch.zeros_like(wq_all)
        gwq = [g_all[m].view(w.shape) for m, w in enumerate(ws)]
        gx = torch.zeros_like(x)
        stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        rc = lib.mulut_ft_stage_backward_mask(x.device.index, _ptr_array(wq), modes.encode(), int(is_last), int(u), x.data_ptr(),
                                              gout.data_ptr(), inside.data_ptr(), B, C, H, W, _ptr_array(gwq), gx.data_ptr(), stream)
        if rc:
            raise RuntimeError(lib.mulut_strerror(rc).decode())
        # backward of clamp(round_func(w*127)): round is identity (BPDA), clamp passes inside [-127,127], x127 -- in place, one launch
        rc = lib.mulut_ft_quantize_backward(x.device.index, _ptr_array(ws), _ptr_array(gwq), len(ws), ws[0].numel(), stream)
        if rc:
            raise RuntimeError(lib.mulut_strerror(rc).decode())
        return (gx, None, None, None) + tuple(gwq)


class MuLUT(nn.Module):
    """PyTorch module for LUT-aware fine-tuning on the GPU (twin of sr/model.py:39-312)."""

    def __init__(self, lut_folder, stages, modes, upscale=4, interval=4):
        super().__init__()
        if interval != 4:
            raise NotImplementedError("only interval 4 is supported")
        self.interval, self.upscale, self.stages = interval, upscale, stages
        self.modes = "".join(modes)
        for s in range(stages):
            stage = s + 1
            scale = upscale if stage == stages else 1
            for mode in self.modes:
                # writer-side naming, as the reference's module reads it (sr/model.py:51-53)
                path = os.path.join(lut_folder, "LUT_x{}_{}bit_int8_s{}_{}.npy".format(upscale, interval, stage, mode))
                arr = np.load(path).reshape(-1, scale * scale).astype(np.float32) / 127.0
                self.register_parameter("weight_s{}_{}".format(stage, mode), nn.Parameter(torch.from_numpy(arr)))

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("mulut_amd.finetune.MuLUT has no CPU path; move the module and input to the GPU")
        x = x * 255.0
        for s in range(self.stages):
            stage = s + 1
            last = stage == self.stages
            weights = [getattr(self, "weight_s{}_{}".format(stage, m)) for m in self.modes]
            x = _StageFn.apply(x, self.modes, last, self.upscale if last else 1, *weights)
        return x / 255.0

    def export_int8(self):
        """{ 's{stage}_{mode}': int8 table } as sr/3_finetune_lut.py:162-169 writes LUT_ft_*.npy."""
        out = {}
        for s in range(self.stages):
            for m in self.modes:
                w = getattr(self, "weight_s{}_{}".format(s + 1, m)).detach().cpu().numpy()
                out["s{}_{}".format(s + 1, m)] = np.round(np.clip(w, -1, 1) * 127).astype(np.int8)
        return out
