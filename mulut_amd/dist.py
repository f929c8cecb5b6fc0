"""Multi-GPU sharding of the LUT-inference path: one process per GPU, torch.distributed (RCCL over xGMI).

The reference parallelises over images only (``Pool(24).map`` over files, sr/4_test_lut.py:257-259).
Two shardings are offered:

* **frames** (default, what bench.py scales with): images / frames of a batch are independent units ->
  each rank takes a contiguous slice, no data-path collective at all.
* **strips** (one big frame, latency mode): the LR image is cut into ``world`` horizontal strips; a rank
  needs its strip plus a ``halo`` of LR rows (2 per stage) above and below, clamped to the image, and
  runs the whole cascade on it with ``mulut_pipeline_rows`` (edge replication only at true image
  borders, so seams are bit-exact).  The only exchange step is the final gather of the uint8 HR strips
  (RCCL ``gather`` / ``all_gather`` -- backend "nccl" on ROCm).

``compute`` callables keep this module free of any engine dependency, so the CPU (gloo) tests can drive
the same bookkeeping with a CPU test double.
"""
import torch
import torch.distributed as dist


def frame_slice(n_frames, world, rank):
    """Contiguous, balanced slice [lo, hi) of n_frames for `rank` (first n % world ranks get one more)."""
    base, extra = divmod(n_frames, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def strip_bounds(height, world):
    """LR row ranges [y0, y1) of the `world` strips (balanced like frame_slice)."""
    return [frame_slice(height, world, r) for r in range(world)]


def strip_band(height, world, rank, halo):
    """(band_row0, band_row1, y0, y1): the rows a rank must hold to produce LR rows [y0, y1)."""
    y0, y1 = frame_slice(height, world, rank)
    return max(0, y0 - halo), min(height, y1 + halo), y0, y1


def sr_strips(lr, compute, scale, halo, group=None, dst=0):
    """Super-resolve one (batch of) frame(s) split into horizontal strips across the process group.

    lr      : uint8 tensor [H,W,C] or [N,H,W,C] present on EVERY rank (the input is 1/scale^2 of the
              output, so it is simply replicated / read by each rank)
    compute : callable(band, band_row0, y0, y1, H) -> uint8 tensor with the output rows of LR rows
              [y0,y1) (``MuLUTEngine.pipeline_rows`` on the GPU)
    dst     : rank that receives the assembled frame (RCCL gather); None = every rank (all_gather)
    returns the full [.., H*scale, W*scale, C] tensor on `dst` (or everywhere), else None.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    batched = lr.dim() == 4
    H = lr.shape[1] if batched else lr.shape[0]
    if H < world:
        raise ValueError("fewer image rows than ranks")
    r0, r1, y0, y1 = strip_band(H, world, rank, halo)
    band = (lr[:, r0:r1] if batched else lr[r0:r1]).contiguous()
    mine = compute(band, r0, y0, y1, H)
    if world == 1:
        return mine
    # strips differ by at most one LR row: pad to the tallest so that one collective moves everything
    rows_max = max(b - a for a, b in strip_bounds(H, world)) * scale
    row_dim = 1 if batched else 0
    pad_shape = list(mine.shape)
    pad_shape[row_dim] = rows_max
    send = torch.zeros(pad_shape, dtype=mine.dtype, device=mine.device)
    send.narrow(row_dim, 0, mine.shape[row_dim]).copy_(mine)
    if dst is None:
        recv = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(recv, send, group=group)
    else:
        recv = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
        dist.gather(send, recv, dst=dst, group=group)
        if rank != dst:
            return None
    parts = [recv[r].narrow(row_dim, 0, (b - a) * scale) for r, (a, b) in enumerate(strip_bounds(H, world))]
    return torch.cat(parts, dim=row_dim)


def sr_frames(frames, compute, group=None):
    """Frame sharding: this rank's slice of a replicated batch, no collective. Returns (lo, hi, output)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = frame_slice(frames.shape[0], world, rank)
    return lo, hi, (compute(frames[lo:hi].contiguous()) if hi > lo else None)
