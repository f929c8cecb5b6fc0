"""world_size-2 (and 3) gloo tests of the multi-GPU bookkeeping on CPU: strip + halo sharding is
bit-exact across seams, the gather assembles the frame, frame slices are balanced.  The oracle stands
in for the GPU engine as a test double (mulut_amd.dist takes the compute step as a callable)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN
from mulut_amd.dist import frame_slice, sr_frames, sr_strips, strip_band, strip_bounds
from oracle import c_oracle

STAGES, MODES, SCALE, HALO = 2, "sdy", 4, 4


def _luts():
    d = {}
    for s in (1, 2):
        for m in "sdy":
            a = np.load(os.path.join(GOLDEN, "luts", "LUT_ft_x4_4bit_int8_s%d_%s.npy" % (s, m)))
            d["s%d_%s" % (s, m)] = a.reshape(-1, 16 if s == 2 else 1)
    return d


def _oracle_rows(luts):
    def compute(band, band_row0, y0, y1, H):
        arr = band.numpy()
        outs = []
        for img in (arr if arr.ndim == 4 else arr[None]):
            full = c_oracle.pipeline(luts, STAGES, MODES, SCALE, img)       # replicates at the band's borders ...
            outs.append(full[(y0 - band_row0) * SCALE:(y1 - band_row0) * SCALE])   # ... which the halo keeps out
        out = np.stack(outs)
        return torch.from_numpy(out if arr.ndim == 4 else out[0])
    return compute


def _worker(rank, world, port, dst, shape, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        luts = _luts()
        img = torch.from_numpy(np.random.default_rng(7).integers(0, 256, shape, dtype=np.uint8))
        if dst == "rotate":
            # batched, frame n assembled on rank n % world, asynchronous handle; preallocated output on rank 0 only
            owned = [n for n in range(shape[0]) if n % world == rank]
            pre = torch.full((len(owned), shape[1] * SCALE, shape[2] * SCALE, shape[3]), 9, dtype=torch.uint8) if (rank == 0 and owned) else None
            pend = sr_strips(img, _oracle_rows(luts), SCALE, HALO, dst="rotate", out=pre, wait=False)
            out = pend.wait()
            assert pend.frames == owned and ((out is pre) if pre is not None else True) and ((out is None) == (not owned))
            want = [c_oracle.pipeline(luts, STAGES, MODES, SCALE, img.numpy()[n]) for n in owned]
            q.put((rank, all(np.array_equal(out[k].numpy(), w) for k, w in enumerate(want)), None if out is None else tuple(out.shape)))
            # sharded mode: nothing is exchanged, every rank keeps its rows
            rows = sr_strips(img, _oracle_rows(luts), SCALE, HALO, dst="none")
            a, b = strip_bounds(shape[1], world)[rank]
            assert tuple(rows.shape) == (shape[0], (b - a) * SCALE, shape[2] * SCALE, shape[3])
        elif img.dim() == 4:
            # batched case: preallocated output + asynchronous handle (what bench.py's config-3 leg does)
            shape_out = (shape[0], shape[1] * SCALE, shape[2] * SCALE, shape[3])
            pre = torch.full(shape_out, 7, dtype=torch.uint8) if (dst is None or rank == dst) else None
            pend = sr_strips(img, _oracle_rows(luts), SCALE, HALO, dst=dst, out=pre, wait=False)
            out = pend.wait()
            assert (out is pre) if pre is not None else (out is None)
        else:
            out = sr_strips(img, _oracle_rows(luts), SCALE, HALO, dst=dst)
        if dst == "rotate":
            pass
        elif dst is None or rank == dst:
            arr = img.numpy()
            want = np.stack([c_oracle.pipeline(luts, STAGES, MODES, SCALE, a) for a in (arr if arr.ndim == 4 else arr[None])])
            want = want if arr.ndim == 4 else want[0]
            q.put((rank, bool(np.array_equal(out.numpy(), want)), tuple(out.shape)))
        else:
            q.put((rank, out is None, None))
        # frame sharding: slices partition the batch
        frames = torch.arange(5 * 2 * 2 * 1, dtype=torch.uint8).reshape(5, 2, 2, 1)
        lo, hi, res = sr_frames(frames, lambda x: x + 1)
        q.put((rank, lo, hi, None if res is None else int(res.sum())))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,dst,shape", [(2, 0, (37, 29, 3)), (2, None, (2, 23, 17, 3)), (3, 1, (31, 16, 1)),
                                             (2, "rotate", (3, 23, 17, 3)), (3, "rotate", (2, 19, 16, 1))])
def test_strips_over_gloo(world, dst, shape):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, dst, shape, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2 * world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    strips = [g for g in got if len(g) == 3]
    frames = sorted(g for g in got if len(g) == 4)
    assert len(strips) == world and all(ok for _, ok, _ in strips)
    H = shape[-3]
    for r, ok, shp in strips:
        if shp is not None:
            assert shp[-3] == H * SCALE
    if dst == "rotate":      # every frame of the batch is held by exactly one rank
        assert sum(shp[0] for _, _, shp in strips if shp is not None) == shape[0]
    # frame slices are contiguous, disjoint and cover the batch of 5
    assert frames[0][1] == 0 and frames[-1][2] == 5
    for a, b in zip(frames, frames[1:]):
        assert a[2] == b[1]


def test_single_process_fills_out():
    """world == 1: a caller-supplied `out` is filled and returned."""
    img = torch.from_numpy(np.random.default_rng(3).integers(0, 256, (9, 8, 1), dtype=np.uint8))
    out = torch.zeros((36, 32, 1), dtype=torch.uint8)
    got = sr_strips(img, lambda band, r0, y0, y1, H: band.repeat_interleave(SCALE, 0).repeat_interleave(SCALE, 1), SCALE, HALO, out=out)
    assert got is out and torch.equal(out, img.repeat_interleave(SCALE, 0).repeat_interleave(SCALE, 1))


def test_slicing_helpers():
    assert [frame_slice(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert strip_bounds(1080, 8) == [(135 * r, 135 * (r + 1)) for r in range(8)]
    assert strip_band(1080, 8, 0, 4) == (0, 139, 0, 135)
    assert strip_band(1080, 8, 7, 4) == (941, 1080, 945, 1080)
    assert strip_band(1080, 8, 3, 4) == (401, 544, 405, 540)
    cover = []
    for w in (1, 2, 3, 5, 8):
        for h in (8, 9, 1080, 2160):
            b = strip_bounds(h, w)
            assert b[0][0] == 0 and b[-1][1] == h and all(x[1] == y[0] for x, y in zip(b, b[1:]))
            assert max(y - x for x, y in b) - min(y - x for x, y in b) <= 1
            cover.append(b)
    assert cover
