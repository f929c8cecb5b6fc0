"""Pins the fine-tune CPU oracle (oracle/ft_torch.py) to outputs AND gradients of the reference's own MuLUT module
(tests/golden/ft_fixtures.npz, produced by tests/golden/gen_golden_ft.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import ft_torch


def synthetic_lut(seed, vnum):
    rng = np.random.default_rng(seed)
    return rng.integers(-127, 128, size=(17 ** 4, vnum), dtype=np.int8)


def tables_for(fx, name):
    stages, scale = [int(v) for v in fx[name + "/cfg"]]
    modes = bytes(fx[name + "/modes"]).decode()
    src = bytes(fx[name + "/lutsrc"]).decode()
    out = {}
    for s in range(stages):
        vnum = scale * scale if s + 1 == stages else 1
        for m in modes:
            key = "s%d_%s" % (s + 1, m)
            if src == "shipped":
                t = np.load(os.path.join(GOLDEN, "luts", "LUT_ft_x4_4bit_int8_%s.npy" % key)).reshape(-1, vnum)
            elif src == "s2_s":
                t = np.load(os.path.join(GOLDEN, "luts", "LUT_ft_x4_4bit_int8_s2_s.npy")).reshape(-1, 16)
            else:
                t = synthetic_lut(17 * s + ord(m), vnum)
            out[key] = t.astype(np.int8)
    return out, stages, modes, scale


@pytest.mark.parametrize("name", ["A_s2sdy_x4_u8", "B_s1s_x4_float", "C_s2sd_x2_u8", "D_s2sdy_x4_float"])
def test_ft_oracle_matches_reference(name):
    fx = np.load(os.path.join(GOLDEN, "ft_fixtures.npz"))
    tabs, stages, modes, scale = tables_for(fx, name)
    weights = {k: torch.from_numpy(v.astype(np.float32) / 127.0).requires_grad_(True) for k, v in tabs.items()}
    x = torch.from_numpy(fx[name + "/x"]).requires_grad_(True)
    y = ft_torch.forward(weights, x, stages, modes, scale)
    assert np.array_equal(y.detach().numpy(), fx[name + "/y"])
    loss = torch.nn.functional.mse_loss(y, torch.from_numpy(fx[name + "/target"]))
    loss.backward()
    assert abs(loss.item() - float(fx[name + "/loss"])) < 1e-7
    assert np.allclose(x.grad.numpy(), fx[name + "/grad_x"], rtol=1e-5, atol=1e-9)
    for key, w in weights.items():
        dense = np.zeros((17 ** 4, w.shape[1]), np.float32)
        dense[fx[name + "/grad/" + key + "/rows"]] = fx[name + "/grad/" + key + "/vals"]
        assert np.allclose(w.grad.numpy(), dense, rtol=1e-5, atol=1e-9), key


def test_ft_oracle_bad_mode():
    with pytest.raises(ValueError):
        ft_torch.interp_batch(torch.zeros(17 ** 4, 1), 1, "x", torch.zeros(1, 1, 4, 4), 1)
