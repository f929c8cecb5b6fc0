"""Seeded synthetic frames for bench.py and the tests (SURVEY.md 8d input distributions)."""
import numpy as np


def noise_frames(n, h, w, c=3, seed=0):
    """D-noise: uniform random bytes -- worst case for LUT-key locality."""
    return np.random.default_rng(seed).integers(0, 256, (n, h, w, c), dtype=np.uint8)


def natural_frames(n, h, w, c=3, seed=0):
    """D-natural: per channel a sum of 6 low-frequency 2-D sinusoids spanning 0..255 plus sigma=2
    Gaussian noise -- neighbouring pixels share MSB keys the way photographs do."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    out = np.empty((n, h, w, c), dtype=np.uint8)
    for i in range(n):
        for ch in range(c):
            acc = np.zeros((h, w), dtype=np.float32)
            for _ in range(6):
                fy, fx = rng.uniform(0.5, 6.0, 2) * 2 * np.pi / max(h, w)
                acc += np.float32(rng.uniform(0.3, 1.0)) * np.sin(np.float32(fy) * yy + np.float32(fx) * xx +
                                                                  np.float32(rng.uniform(0, 2 * np.pi)))
            acc = (acc - acc.min()) / (acc.max() - acc.min()) * 255.0
            acc += rng.normal(0, 2, (h, w)).astype(np.float32)
            out[i, :, :, ch] = np.clip(np.rint(acc), 0, 255).astype(np.uint8)
    return out


def real_frames(n, h, w, png_path, seed=0):
    """D-real: a photograph (the DIV2K LR sample the reference ships) mirrored/tiled to h x w, shifted per frame."""
    from PIL import Image
    img = np.array(Image.open(png_path).convert("RGB"))
    # mirror-tile so that seams are continuous, then crop
    row = np.concatenate([img, img[:, ::-1]], axis=1)
    full = np.concatenate([row, row[::-1]], axis=0)
    reps = (-(-h // full.shape[0]) + 1, -(-w // full.shape[1]) + 1)
    big = np.tile(full, (reps[0], reps[1], 1))
    rng = np.random.default_rng(seed)
    out = np.empty((n, h, w, 3), dtype=np.uint8)
    for i in range(n):
        oy = int(rng.integers(0, big.shape[0] - h + 1))
        ox = int(rng.integers(0, big.shape[1] - w + 1))
        out[i] = big[oy:oy + h, ox:ox + w]
    return out
