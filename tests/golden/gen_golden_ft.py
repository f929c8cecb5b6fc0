#!/usr/bin/env python3
"""Golden fixtures for the LUT fine-tune path produced by RUNNING THE REFERENCE's
`MuLUT` module (sr/model.py:39-312) on the CPU in this container: forward output and every parameter
gradient of an MSE loss for a few seeded batches.  Only data is written (tests/golden/ft_fixtures.npz).
Inert when /root/reference is absent.   python tests/golden/gen_golden_ft.py
"""
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def synthetic_lut(seed, vnum):
    rng = np.random.default_rng(seed)
    return rng.integers(-127, 128, size=(17 ** 4, vnum), dtype=np.int8)


def main():
    if not os.path.isdir(REF):
        raise SystemExit("gen_golden_ft.py: /root/reference not present")
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "sr"))
    cwd = os.getcwd()
    os.chdir(os.path.join(REF, "sr"))          # model.py does sys.path.insert(0, "../")
    try:
        import model as ref_model               # noqa: E402  (the reference's sr/model.py)
    finally:
        os.chdir(cwd)
    out = {}
    shipped = {k: np.load(os.path.join(REF, "models/sr_x2sdy", "LUT_ft_x4_4bit_int8_%s.npy" % k))
               for k in ("s1_s", "s1_d", "s1_y", "s2_s", "s2_d", "s2_y")}
    cases = [
        # name, stages, modes, scale, lut source, input kind, shape
        ("A_s2sdy_x4_u8", 2, "sdy", 4, "shipped", "u8", (2, 1, 12, 10)),
        ("B_s1s_x4_float", 1, "s", 4, "s2_s", "float", (1, 2, 7, 9)),
        ("C_s2sd_x2_u8", 2, "sd", 2, "synth", "u8", (3, 1, 8, 8)),
        ("D_s2sdy_x4_float", 2, "sdy", 4, "shipped", "float", (1, 1, 9, 11)),
    ]
    for name, stages, modes, scale, src, kind, shape in cases:
        rng = np.random.default_rng(abs(hash(name)) % (2 ** 31))
        rng = np.random.default_rng(sum(map(ord, name)))
        with tempfile.TemporaryDirectory() as td:
            luts = {}
            for s in range(stages):
                vnum = scale * scale if s + 1 == stages else 1
                for m in modes:
                    key = "s%d_%s" % (s + 1, m)
                    if src == "shipped":
                        t = shipped[key].reshape(-1, vnum)
                    elif src == "s2_s":
                        t = shipped["s2_s"].reshape(-1, 16)
                    else:
                        t = synthetic_lut(17 * s + ord(m), vnum)
                    luts[key] = np.ascontiguousarray(t.astype(np.int8))
                    np.save(os.path.join(td, "LUT_x%d_4bit_int8_%s.npy" % (scale, key)), luts[key])
            net = ref_model.MuLUT(td, stages, list(modes), upscale=scale, interval=4)
        if kind == "u8":
            x = rng.integers(0, 256, shape).astype(np.float32) / np.float32(255.0)
        else:
            x = rng.random(shape, dtype=np.float32)
        tgt = rng.random((shape[0], shape[1], shape[2] * scale, shape[3] * scale), dtype=np.float32)
        xt = torch.from_numpy(x).requires_grad_(True)
        y = net(xt)
        loss = torch.nn.functional.mse_loss(y, torch.from_numpy(tgt))
        loss.backward()
        out[name + "/x"] = x
        out[name + "/target"] = tgt
        out[name + "/y"] = y.detach().numpy()
        out[name + "/loss"] = np.float32(loss.item())
        out[name + "/grad_x"] = xt.grad.numpy()
        out[name + "/cfg"] = np.array([stages, scale], dtype=np.int32)
        out[name + "/modes"] = np.frombuffer(modes.encode(), dtype=np.uint8)
        out[name + "/lutsrc"] = np.frombuffer(src.encode(), dtype=np.uint8)   # tables are rebuilt by the tests, not stored
        for key, t in luts.items():
            g = getattr(net, "weight_" + key).grad.numpy()
            rows = np.nonzero(np.abs(g).sum(1))[0]
            out[name + "/grad/" + key + "/rows"] = rows.astype(np.int32)
            out[name + "/grad/" + key + "/vals"] = g[rows]
        print(name, "loss %.6f" % loss.item(), "y", y.shape, {k: int((getattr(net, "weight_" + k).grad.abs().sum(1) > 0).sum())
                                                               for k in luts})
    np.savez_compressed(os.path.join(HERE, "ft_fixtures.npz"), **out)
    print("wrote ft_fixtures.npz", os.path.getsize(os.path.join(HERE, "ft_fixtures.npz")))


if __name__ == "__main__":
    main()
