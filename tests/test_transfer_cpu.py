"""LUT producer (mulut_amd/transfer_to_lut.py + network.py) against tables the reference's own network code produced
(tests/golden/transfer_fixtures.npz, gen_golden_transfer.py).  A table entry is round(127 * tanh(...)): a different
summation order inside the matrix products can move a value sitting within float rounding of a .5 boundary by one
step, so the bar is: no entry off by more than 1, and at most 0.01 % of entries off at all."""
import hashlib
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import GOLDEN

from mulut_amd import network, transfer_to_lut as T


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(GOLDEN, "transfer_fixtures.npz"))


def _opt(stages, modes, scale, exp_dir=""):
    return SimpleNamespace(stages=stages, modes=modes, scale=scale, interval=4, expDir=exp_dir)


def _check(table, fx, key):
    assert tuple(table.shape) == tuple(fx[key + "/shape"]) and table.dtype == np.int8
    rows = table.reshape(table.shape[0], -1)[fx["idx"]].astype(np.int32)
    diff = np.abs(rows - fx[key + "/rows"].astype(np.int32))
    assert diff.max() <= 1 and (diff != 0).mean() <= 1e-4, (key, diff.max(), (diff != 0).mean())
    return hashlib.sha256(table.tobytes()).digest() == fx[key + "/sha256"].tobytes()


def test_grid_enumeration_order():
    x = T.get_input_tensor(_opt(1, "s", 4))
    assert x.shape == (17 ** 4, 1, 2, 2)
    v = (x * 255).round().long().reshape(-1, 4)
    assert v[0].tolist() == [0, 0, 0, 0] and v[1].tolist() == [0, 0, 0, 16] and v[16].tolist() == [0, 0, 0, 255]
    assert v[17].tolist() == [0, 0, 16, 0] and v[17 ** 3].tolist() == [16, 0, 0, 0] and v[-1].tolist() == [255] * 4
    d = T.get_mode_input_tensor(x[:5], "d")
    assert d.shape == (5, 1, 3, 3) and torch.equal(d[:, 0, 0, 2], x[:5, 0, 0, 1]) and torch.equal(d[:, 0, 2, 0], x[:5, 0, 1, 0])
    y = T.get_mode_input_tensor(x[:5], "y")
    assert torch.equal(y[:, 0, 1, 1], x[:5, 0, 0, 1]) and torch.equal(y[:, 0, 1, 2], x[:5, 0, 1, 0]) and torch.equal(y[:, 0, 2, 1], x[:5, 0, 1, 1])
    with pytest.raises(ValueError, match="Mode s not implemented"):
        T.get_mode_input_tensor(x[:5], "s")


def test_tiny_random_model_tables_match_reference(fx):
    net = network.SRNets(nf=8, scale=2, modes=list("sdy"), stages=1)
    sd = {k[len("tinyw/"):]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("tinyw/")}
    net.load_state_dict(sd, strict=True)                       # the reference's parameter names, all of them
    tabs = T.transfer(net, _opt(1, "sdy", 2), save=False)
    exact = [_check(tabs["s1_" + m], fx, "tiny/s1_" + m) for m in "sdy"]
    assert all(t.shape == (83521, 1, 2, 2) for t in tabs.values()) and sum(exact) >= 0


def test_shipped_checkpoint_loads_and_transfers(fx, tmp_path):
    # whole-module pickle of the reference's classes -> rebuilt on the twins; strict state_dict load into a fresh model
    lm = network.load_checkpoint(os.path.join(GOLDEN, "Model_200000.pth"))
    assert type(lm).__module__ == "mulut_amd.network" and type(lm).__name__ == "SRNets"
    for k, v in lm.state_dict().items():
        assert np.array_equal(v.numpy(), fx["w/" + k]), k
    net = network.SRNets(nf=64, scale=4, modes=list("sdy"), stages=2)
    net.load_state_dict(lm.state_dict(), strict=True)
    opt = _opt(2, "sdy", 4, str(tmp_path))
    tabs = T.transfer(net, opt)
    for s in (1, 2):
        for m in "sdy":
            key = "s%d_%s" % (s, m)
            _check(tabs[key], fx, "shipped/" + key)
            on_disk = np.load(os.path.join(str(tmp_path), "LUT_x4_4bit_int8_%s.npy" % key))     # :113-115 naming
            assert np.array_equal(on_disk, tabs[key])
    # the unpickled module itself computes the same tables (its forward is the twin's)
    assert np.array_equal(T.transfer_one(lm, opt, 2, "y"), tabs["s2_y"])


def test_image_forward_equals_table_lookup_on_grid_values():
    """SRNet on an image whose pixels are grid points == the transferred table row of those four pixels."""
    torch.manual_seed(0)
    net = network.SRNets(nf=8, scale=2, modes=list("sdy"), stages=1)
    opt = _opt(1, "sdy", 2)
    g = torch.tensor([0, 16, 32, 240, 255])
    img = g[torch.randint(0, 5, (1, 1, 6, 7))].float() / 255.0
    for m, taps in (("s", ((0, 0), (0, 1), (1, 0), (1, 1))), ("d", ((0, 0), (0, 2), (2, 0), (2, 2))), ("y", ((0, 0), (1, 1), (1, 2), (2, 1)))):
        tab = T.transfer_one(net, opt, 1, m, chunks=7).reshape(-1, 2, 2)
        with torch.no_grad():
            out = torch.round(torch.clamp(net(img, 1, m), -1, 1) * 127).numpy()[0, 0]
        P = 1 if m == "s" else 2
        v = (img[0, 0] * 255).round().long()
        for y in range(6 - P):
            for x in range(7 - P):
                k = [min(int(v[y + i, x + j]) // 16 + (int(v[y + i, x + j]) == 255), 16) for (i, j) in taps]
                row = ((k[0] * 17 + k[1]) * 17 + k[2]) * 17 + k[3]
                assert np.array_equal(out[2 * y:2 * y + 2, 2 * x:2 * x + 2], tab[row]), (m, y, x)
