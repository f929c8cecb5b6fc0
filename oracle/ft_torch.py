"""CPU oracle for the LUT fine-tune path -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

A compact restatement, in vectorised torch ops with autograd, of the reference's differentiable module
(MuLUT.InterpTorchBatch + MuLUT.forward, sr/model.py:69-312): weight quantisation with BPDA rounding (:74-76),
MSB/LSB split with a differentiable ``%`` (:85-121), the 24-branch strict-'>' case cascade (:191-282) expressed as
a rank table, float32 left-to-right weighted sum, /q, rot90 back, ``pred = round_func(pred)`` after every pass
(:308) and the stage clamp/round (:309).  Pinned against tests/golden/ft_fixtures.npz (outputs AND gradients of the
reference itself) by tests/test_oracle_ft.py.
"""
import torch
import torch.nn.functional as F

PATTERNS = {
    "s": ((0, 0), (0, 1), (1, 0), (1, 1)),
    "d": ((0, 0), (0, 2), (2, 0), (2, 2)),
    "y": ((0, 0), (1, 1), (1, 2), (2, 1)),
}
PAD = {"s": 1, "d": 2, "y": 2}


def round_bpda(x):
    """sr/model.py:59-67: forward = round, backward = identity."""
    return x + (torch.round(x) - x).detach()


def _case_order(fa, fb, fc, fd):
    """[..., 4] long: key ids (0=a..3=d) in the order the reference's if-cascade ranks them."""
    fab, fac, fad, fbc, fbd, fcd = fa > fb, fa > fc, fa > fd, fb > fc, fb > fd, fc > fd
    T = lambda *o: torch.tensor(o, dtype=torch.long)  # noqa: E731

    def pick(c1, o1, c2, o2, c3, o3, o4):
        return torch.where(c1[..., None], T(*o1), torch.where(c2[..., None], T(*o2), torch.where(c3[..., None], T(*o3), T(*o4))))

    g1 = pick(fcd, (0, 1, 2, 3), fbd, (0, 1, 3, 2), fad, (0, 3, 1, 2), (3, 0, 1, 2))     # fab & fbc
    g2 = pick(fbd, (0, 2, 1, 3), fcd, (0, 2, 3, 1), fad, (0, 3, 2, 1), (3, 0, 2, 1))     # fab & ~fbc & fac
    g3 = pick(fbd, (2, 0, 1, 3), fad, (2, 0, 3, 1), fcd, (2, 3, 0, 1), (3, 2, 0, 1))     # fab & ~fbc & ~fac
    g4 = pick(fcd, (1, 0, 2, 3), fad, (1, 0, 3, 2), fbd, (1, 3, 0, 2), (3, 1, 0, 2))     # ~fab & fac
    g5 = pick(fad, (1, 2, 0, 3), fcd, (1, 2, 3, 0), fbd, (1, 3, 2, 0), (3, 1, 2, 0))     # ~fab & ~fac & fbc
    g6 = pick(fad, (2, 1, 0, 3), fbd, (2, 1, 3, 0), fcd, (2, 3, 1, 0), (3, 2, 1, 0))     # ~fab & ~fac & ~fbc
    w = lambda c, a, b: torch.where(c[..., None], a, b)  # noqa: E731
    return w(fab & fbc, g1, w(fab & fac, g2, w(fab, g3, w(fac, g4, w(fbc, g5, g6)))))


def interp_batch(weight, upscale, mode, img_in, bd, interval=4):
    """InterpTorchBatch (sr/model.py:69-287): weight [L^4, u*u] (learnable, int8/127 scale), img_in [B,C,h+bd,w+bd]."""
    if mode not in PATTERNS:
        raise ValueError("Mode {} not implemented.".format(mode))
    q, L = 2 ** interval, 2 ** (8 - interval) + 1
    wq = torch.clamp(round_bpda(weight * 127), -127, 127)
    B, C, Hp, Wp = img_in.shape
    h, w = Hp - bd, Wp - bd
    crops = [img_in[:, :, di:di + h, dj:dj + w] for di, dj in PATTERNS[mode]]
    msb = torch.stack([torch.floor_divide(c, q).long() for c in crops], -1)      # [B,C,h,w,4]
    lsb = torch.stack([c % q for c in crops], -1)
    order = _case_order(lsb[..., 0], lsb[..., 1], lsb[..., 2], lsb[..., 3])
    strides = torch.tensor([L ** 3, L ** 2, L, 1], dtype=torch.long)
    fs = torch.gather(lsb, -1, order)                                             # sorted fractional parts
    ss = strides[order]
    base = (msb * strides).sum(-1)
    idx = torch.cat([base[..., None], base[..., None] + torch.cumsum(ss, -1)], -1)   # five vertices
    wt = torch.cat([q - fs[..., :1], fs[..., :-1] - fs[..., 1:], fs[..., 3:]], -1)   # [.., 5]
    rows = wq[idx]                                                                # [B,C,h,w,5,u*u]
    out = wt[..., 0, None] * rows[..., 0, :]
    for j in range(1, 5):                                                         # same left-to-right float32 sum
        out = out + wt[..., j, None] * rows[..., j, :]
    out = out.reshape(B, C, h, w, upscale, upscale).permute(0, 1, 2, 4, 3, 5).reshape(B, C, h * upscale, w * upscale)
    return out / q


def forward(weights, x, stages, modes, upscale, interval=4):
    """MuLUT.forward (sr/model.py:289-312).  weights: dict 's{stage}_{mode}' -> tensor [L^4, u*u] (requires_grad ok)."""
    x = x * 255.0
    for s in range(stages):
        stage = s + 1
        last = stage == stages
        avg, bias, scale = (len(modes), 0, upscale) if last else (len(modes) * 4, 127, 1)
        pred = 0
        for mode in modes:
            pad = PAD[mode]
            wgt = weights["s{}_{}".format(stage, mode)]
            for r in range(4):
                t = F.pad(torch.rot90(x, r, [2, 3]), (0, pad, 0, pad), mode="replicate")
                pred = pred + torch.rot90(interp_batch(wgt, scale, mode, t, pad, interval), (4 - r) % 4, [2, 3])
                pred = round_bpda(pred)
        x = round_bpda(torch.clamp(pred / avg + bias, 0, 255))
    return x / 255.0
