"""The LUT-convertible SR network of the reference (sr/model.py:15-36 `SRNets`, common/network.py:16-227), restated.

Each `SRNet` is a per-site MLP over the four pixels of its sampling pattern: 4 -> nf, four densely connected
nf-wide layers, 5*nf -> upscale^2, tanh.  The reference runs it as unfold -> Conv2d stack -> fold; here the four
taps are gathered as shifted views and the layers are plain matrix products on [sites, features] (one GEMM per
layer -- rocBLAS / hipBLASLt on the GPU), which is the same arithmetic on the same weights.  Class names, module
tree and parameter names are the reference's, so `load_state_dict(strict=True)` of a reference checkpoint works and
a whole-module checkpoint (`torch.save(model_G)`, sr/1_train_model.py) unpickles onto these classes through
`reference_checkpoint_aliases()`.
"""
import contextlib
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

# (row, col) of pixels a, b, c, d inside the K x K patch, per pattern letter (common/network.py:150-227)
TAPS = {
    "S": (2, ((0, 0), (0, 1), (1, 0), (1, 1))),
    "D": (3, ((0, 0), (0, 2), (2, 0), (2, 2))),
    "Y": (3, ((0, 0), (1, 1), (1, 2), (2, 1))),
    "E": (4, ((0, 0), (0, 3), (3, 0), (3, 3))),
    "H": (4, ((0, 0), (2, 2), (2, 3), (3, 2))),
    "O": (4, ((0, 0), (2, 2), (1, 3), (3, 1))),
}


class Conv(nn.Module):
    """common/network.py:16-28 -- a Conv2d with MSRA init; used here as the holder of a layer's weight matrix."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, bias=True):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                              dilation=dilation, bias=bias)
        nn.init.kaiming_normal_(self.conv.weight)
        if bias:
            nn.init.constant_(self.conv.bias, 0)

    def matrix(self):
        return self.conv.weight.reshape(self.conv.weight.shape[0], -1), self.conv.bias

    def forward(self, x):
        return self.conv(x)


class ActConv(nn.Module):
    """common/network.py:31-44"""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, bias=True):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                              dilation=dilation, bias=bias)
        self.act = nn.ReLU()
        nn.init.kaiming_normal_(self.conv.weight)
        if bias:
            nn.init.constant_(self.conv.bias, 0)

    def forward(self, x):
        return self.act(self.conv(x))


class DenseConv(nn.Module):
    """common/network.py:47-59"""

    def __init__(self, in_nf, nf=64):
        super().__init__()
        self.act = nn.ReLU()
        self.conv1 = Conv(in_nf, nf, 1)

    def forward(self, x):
        return torch.cat([x, self.act(self.conv1(x))], dim=1)


class MuLUTUnit(nn.Module):
    """common/network.py:62-105.  `features(t)` is the whole block on [sites, 4] tap values."""

    def __init__(self, mode, nf, upscale=1, out_c=1, dense=True):
        super().__init__()
        self.act = nn.ReLU()
        self.upscale = upscale
        if mode == "2x2":
            self.conv1 = Conv(1, nf, 2)
        elif mode == "2x2d":
            self.conv1 = Conv(1, nf, 2, dilation=2)
        elif mode == "2x2d3":
            self.conv1 = Conv(1, nf, 2, dilation=3)
        elif mode == "1x4":
            self.conv1 = Conv(1, nf, (1, 4))
        else:
            raise AttributeError
        if dense:
            self.conv2 = DenseConv(nf, nf)
            self.conv3 = DenseConv(nf + nf * 1, nf)
            self.conv4 = DenseConv(nf + nf * 2, nf)
            self.conv5 = DenseConv(nf + nf * 3, nf)
            self.conv6 = Conv(nf * 5, 1 * upscale * upscale, 1)
        else:
            self.conv2 = ActConv(nf, nf, 1)
            self.conv3 = ActConv(nf, nf, 1)
            self.conv4 = ActConv(nf, nf, 1)
            self.conv5 = ActConv(nf, nf, 1)
            self.conv6 = Conv(nf, upscale * upscale, 1)
        if self.upscale > 1:
            self.pixel_shuffle = nn.PixelShuffle(upscale)

    def features(self, t):
        """t: [sites, 4] (a, b, c, d in 0..1) -> [sites, upscale^2], element si*u+sj (pixel-shuffle order)."""
        w, b = self.conv1.matrix()
        x = F.relu(F.linear(t, w, b))
        for layer in (self.conv2, self.conv3, self.conv4, self.conv5):
            if isinstance(layer, DenseConv):
                w, b = layer.conv1.matrix()
                x = torch.cat([x, F.relu(F.linear(x, w, b))], dim=1)
            else:
                w, b = layer.conv.weight.reshape(layer.conv.weight.shape[0], -1), layer.conv.bias
                x = F.relu(F.linear(x, w, b))
        w, b = self.conv6.matrix()
        return torch.tanh(F.linear(x, w, b))

    def forward(self, x):
        """[N, 1, kh, kw] patches whose conv1 window fits exactly once -> [N, 1, u, u]."""
        n = x.shape[0]
        d = self.conv1.conv.dilation
        kh, kw = self.conv1.conv.kernel_size
        taps = [x[:, 0, i * d[0], j * d[1]] for i in range(kh) for j in range(kw)]
        u = self.upscale
        return self.features(torch.stack(taps, dim=1)).reshape(n, 1, u, u)


class SRNet(nn.Module):
    """common/network.py:137-226: one pattern (S, D, Y, E, H, O) x ('1' = same size, 'N' = upscaling)."""

    def __init__(self, mode, nf=64, upscale=None, dense=True):
        super().__init__()
        self.mode = mode
        letter, kind = mode[0], mode[1:]
        if letter not in TAPS or kind not in ("x1", "xN"):
            raise AttributeError
        if kind == "x1":
            assert upscale is None
        u = 1 if kind == "x1" else upscale
        unit = {"S": "2x2", "D": "2x2d", "E": "2x2d3"}.get(letter, "1x4")
        self.model = MuLUTUnit(unit, nf, upscale=u, dense=dense)
        self.K = TAPS[letter][0]
        self.S = u
        self.P = self.K - 1

    def forward(self, x):
        """[B, C, H, W] -> [B, C, (H-P)*S, (W-P)*S]; every channel goes through the same block."""
        B, C, H, W = x.shape
        h, w = H - self.P, W - self.P
        taps = [x[:, :, i:i + h, j:j + w] for (i, j) in TAPS[self.mode[0]][1]]
        t = torch.stack(taps, dim=-1).reshape(-1, 4)
        u = self.S
        y = self.model.features(t).reshape(B, C, h, w, u, u)
        return y.permute(0, 1, 2, 4, 3, 5).reshape(B, C, h * u, w * u)


class SRNets(nn.Module):
    """sr/model.py:15-36: modules `s{stage}_{mode}`; non-final stages keep the size, the last one upscales."""

    def __init__(self, nf=64, scale=4, modes=("s", "d", "y"), stages=2):
        super().__init__()
        for s in range(stages):
            last = (s + 1) == stages
            for mode in modes:
                self.add_module("s{}_{}".format(s + 1, mode),
                                SRNet("{}x{}".format(mode.upper(), "N" if last else "1"), nf=nf,
                                      upscale=scale if last else None))

    def forward(self, x, stage, mode):
        return getattr(self, "s{}_{}".format(stage, mode))(x)


@contextlib.contextmanager
def reference_checkpoint_aliases():
    """While active, the module paths a reference whole-module pickle names (`model.SRNets`, `common.network.*`)
    resolve to the classes above, so `torch.load(path, weights_only=False)` rebuilds the checkpoint on them."""
    me = sys.modules[__name__]
    shim_model = types.ModuleType("model")
    shim_common = types.ModuleType("common")
    shim_net = types.ModuleType("common.network")
    for name in ("Conv", "ActConv", "DenseConv", "MuLUTUnit", "SRNet", "SRNets"):
        setattr(shim_model, name, getattr(me, name))
        setattr(shim_net, name, getattr(me, name))
    shim_common.network = shim_net
    saved = {k: sys.modules.get(k) for k in ("model", "common", "common.network")}
    sys.modules.update({"model": shim_model, "common": shim_common, "common.network": shim_net})
    try:
        yield
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def load_checkpoint(path, map_location="cpu"):
    """A reference `Model_{iter:06d}.pth` (whole module, sr/1_train_model.py) -> module built from the classes here."""
    with reference_checkpoint_aliases():
        return torch.load(path, map_location=map_location, weights_only=False)
