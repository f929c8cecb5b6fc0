"""NumPy port of the reference's CPU inference path -- TEST INFRASTRUCTURE ONLY (oracle/__init__.py).

Written from the algorithm description, keeping the reference's OPERATION STRUCTURE
so that timing it on the GPU box's host cores stands in for "the reference's own CPU path":
float32 tables, 16 materialised corner gathers per pass (sr/4_test_lut.py:56-109), 24 boolean-masked
select + multiply-add passes (:140-230), block->image reshuffle + rot90 + /q (:232-236), and the
driver's float64 accumulate / divide / np.round / clip (:279-306).  Pinned bit-exact against the
fixtures in tests/golden (tests/test_oracle.py).
"""
import itertools

import numpy as np

# (row, col) offsets of the keys a,b,c,d: 's' :20-23, 'd' :32-35, 'y' :43-46
PATTERNS = {
    "s": ((0, 0), (0, 1), (1, 0), (1, 1)),
    "d": ((0, 0), (0, 2), (2, 0), (2, 2)),
    "y": ((0, 0), (1, 1), (1, 2), (2, 1)),
}
PAD = {"s": 1, "d": 2, "y": 2}   # :289-292


def four_simplex_interp(weight, img_in, h, w, interval, rot, upscale=4, mode="s"):
    """Same signature and result as FourSimplexInterpFaster (sr/4_test_lut.py:14): weight f32
    [L^4, u*u], img_in f32 [C, h+pad, w+pad] (already rotated and edge-padded) -> f64 [C, ., .]."""
    if mode not in PATTERNS:
        raise ValueError("Mode {} not implemented.".format(mode))
    q = 2 ** interval
    L = 2 ** (8 - interval) + 1
    crops = [img_in[:, di:di + h, dj:dj + w] for di, dj in PATTERNS[mode]]
    msb = [(c // q).reshape(-1).astype(np.int_) for c in crops]
    lsb = [(c % q).reshape(-1, 1) for c in crops]
    sz = msb[0].size
    stride = (L * L * L, L * L, L, 1)
    # all 16 corners are gathered and kept, as the reference does
    corner = {}
    for bits in itertools.product((0, 1), repeat=4):
        idx = sum((msb[k] + bits[k]) * stride[k] for k in range(4))
        corner[bits] = weight[idx].reshape(sz, -1)
    out = np.zeros((sz, upscale * upscale))
    # one masked pass per ordering of the four fractional parts (ties broken by key index, which
    # only ever moves a zero-weight vertex)
    def before(i, j):
        return (lsb[i] >= lsb[j]) if i < j else (lsb[i] > lsb[j])
    for order in itertools.permutations(range(4)):
        m = np.logical_and.reduce([before(order[i], order[i + 1]) for i in range(3)]).squeeze(1)
        if not m.any():
            continue
        f = [lsb[k][m] for k in order]
        bits = [0, 0, 0, 0]
        acc = (q - f[0]) * corner[tuple(bits)][m]
        for j in range(4):
            bits[order[j]] = 1
            acc = acc + (f[j] - (f[j + 1] if j < 3 else 0)) * corner[tuple(bits)][m]
        out[m] = acc
    C = img_in.shape[0]
    out = out.reshape(C, h, w, upscale, upscale).transpose(0, 1, 3, 2, 4).reshape(C, h * upscale, w * upscale)
    out = np.rot90(out, rot, [1, 2])
    return out / q


def run_stages(lut_dict, stages, modes, scale, img_hwc, interval=4, return_all=False):
    """Stage / mode / rotation loop of sr/4_test_lut.py:279-306. lut_dict['s{stage}_{mode}'] is a
    float32 [L^4, v_num] table as built at :333. Returns the final uint8 HWC image."""
    img = np.asarray(img_hwc).astype(np.float32)
    outs = []
    for s in range(stages):
        last = (s + 1) == stages
        upscale = scale if last else 1
        avg, bias = (len(modes), 0) if last else (len(modes) * 4, 127)
        pred = 0
        for mode in modes:
            p = PAD[mode]
            for r in range(4):
                rimg = np.rot90(img, r)
                h, w, _ = rimg.shape
                img_in = np.pad(rimg, ((0, p), (0, p), (0, 0)), mode="edge").transpose(2, 0, 1)
                pred = pred + four_simplex_interp(lut_dict["s%d_%s" % (s + 1, mode)], img_in, h, w, interval,
                                                  4 - r, upscale=upscale, mode=mode)
        img = np.round(np.clip(np.clip(pred / avg + bias, 0, 255).transpose(1, 2, 0), 0, 255))
        img = img.astype(np.uint8) if last else img.astype(np.float32)
        outs.append(img.astype(np.uint8))
    return outs if return_all else outs[-1]
