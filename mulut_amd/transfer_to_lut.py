"""LUT producer: the twin of the reference's sr/2_transfer_to_lut.py (grid enumeration :12-41, pattern remaps :44-66,
int8 quantisation :108-109, file naming :113-115).

    python -m mulut_amd.transfer_to_lut --stages 2 --modes sdy -e ../models/sr_x2sdy      # same flags as the reference

The trained network is evaluated on all 17^4 corner combinations of four pixels; row index of the table =
a*17^3 + b*17^2 + c*17 + d with a (the anchor) slowest.
"""
import os

import numpy as np
import torch

from . import network
from .options import TestOptions


def get_input_tensor(opt, device=None):
    """sr/2_transfer_to_lut.py:12-41 -> float32 [L^4, 1, 2, 2] in 0..1, [[a, b], [c, d]], a slowest, d fastest.
    Sampling points are 0, q, 2q, ..., 240, 255 (the last one is 256 - 1, :15)."""
    base = torch.arange(0, 257, 2 ** opt.interval)
    base[-1] -= 1
    if device is not None:
        base = base.to(device)
    g = torch.cartesian_prod(base, base, base, base)
    return g.reshape(-1, 1, 2, 2).float() / 255.0


def get_mode_input_tensor(input_tensor, mode):
    """sr/2_transfer_to_lut.py:44-66: place the 2x2 grid values at the pattern's positions of a zero 3x3 patch."""
    if mode == "d":
        pos = ((0, 0), (0, 2), (2, 0), (2, 2))
    elif mode == "y":
        pos = ((0, 0), (1, 1), (1, 2), (2, 1))
    else:
        raise ValueError("Mode {} not implemented.".format(mode))
    out = torch.zeros((input_tensor.shape[0], input_tensor.shape[1], 3, 3), dtype=input_tensor.dtype,
                      device=input_tensor.device)
    for (i, j), (si, sj) in zip(pos, ((0, 0), (0, 1), (1, 0), (1, 1))):
        out[:, :, i, j] = input_tensor[:, :, si, sj]
    return out


def lut_file_name(opt, stage, mode):
    """:113-115 -- note `opt.interval`, not 8 - interval, in this script's name (both are 4 for the 4-bit tables)."""
    return "LUT_x{}_{}bit_int8_s{}_{}.npy".format(opt.scale, opt.interval, str(stage), mode)


def transfer_one(model_G, opt, stage, mode, device=None, chunks=100):
    """One table: int8 [L^4, 1, u, u] = round(clamp(net(grid), -1, 1) * 127) (:108-109), in `chunks` batches (:86-101)."""
    input_tensor = get_input_tensor(opt, device)
    if mode != "s":
        input_tensor = get_mode_input_tensor(input_tensor, mode)
    B = input_tensor.size(0) // chunks
    outputs = []
    with torch.no_grad():
        model_G.eval()
        for b in range(chunks):
            batch_input = input_tensor[b * B:] if b == chunks - 1 else input_tensor[b * B:(b + 1) * B]
            batch_output = model_G(batch_input, stage=stage, mode=mode)
            outputs.append(torch.round(torch.clamp(batch_output, -1, 1) * 127).cpu().numpy().astype(np.int8))
    return np.concatenate(outputs, 0)


def transfer(model_G, opt, device=None, save=True):
    """All stages x modes (:80-116); returns {'s{stage}_{mode}': table} and writes the .npy files into opt.expDir."""
    out = {}
    for s in range(opt.stages):
        for mode in opt.modes:
            results = transfer_one(model_G, opt, s + 1, mode, device)
            out["s{}_{}".format(s + 1, mode)] = results
            if save:
                lut_path = os.path.join(opt.expDir, lut_file_name(opt, s + 1, mode))
                np.save(lut_path, results)
                print("Resulting LUT size: ", results.shape, "Saved to", lut_path)
    return out


def main(argv=None):
    opt = TestOptions().parse(argv)
    device = torch.device("cuda", opt.device) if torch.cuda.is_available() else torch.device("cpu")
    modes = [m for m in opt.modes]
    model_cls = getattr(network, opt.model)                       # 'SRNets' (common/option.py default)
    model_G = model_cls(nf=opt.nf, scale=opt.scale, modes=modes, stages=opt.stages).to(device)
    lm = network.load_checkpoint(os.path.join(opt.expDir, "Model_{:06d}.pth".format(opt.loadIter)))
    model_G.load_state_dict(lm.state_dict(), strict=True)
    return transfer(model_G, opt, device)


if __name__ == "__main__":
    main()
