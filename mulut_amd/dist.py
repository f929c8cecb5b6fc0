"""Multi-GPU sharding of the LUT-inference path: one process per GPU, torch.distributed (RCCL over xGMI).

The reference parallelises over images only (``Pool(24).map`` over files, sr/4_test_lut.py:257-259).
Two shardings are offered:

* **frames** (default, what bench.py scales with): images / frames of a batch are independent units ->
  each rank takes a contiguous slice, no data-path collective at all.
* **strips** (one big frame, latency mode): the LR image is cut into ``world`` horizontal strips; a rank
  needs its strip plus a ``halo`` of LR rows (2 per stage) above and below, clamped to the image, and
  runs the whole cascade on it with ``mulut_pipeline_rows`` (edge replication only at true image
  borders, so seams are bit-exact).  The only exchange step is the final gather of the uint8 HR strips:
  every strip travels point to point (one batched RCCL group of send / recv, backend "nccl" on ROCm)
  straight into its rows of ONE preallocated output -- strips may differ by a row, nothing is padded,
  concatenated or copied twice.  The input is sliced before it goes to the device: a rank uploads only its
  band.  ``wait=False`` returns the pending requests, so the next frame's compute overlaps this gather.

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


class PendingGather:
    """Handle of an asynchronous strip gather: ``wait()`` returns the assembled tensor (or None off-root)."""

    def __init__(self, out, reqs, stage=None, stage_dst=None):
        self.out, self._reqs, self._stage, self._stage_dst = out, reqs, stage, stage_dst

    def wait(self):
        for r in self._reqs:
            r.wait()
        self._reqs = []
        if self._stage is not None:                 # host-staged receive (gloo rehearsal): back to the device
            self._stage_dst.copy_(self._stage, non_blocking=False)
            self._stage = None
        return self.out


def sr_strips(lr, compute, scale, halo, group=None, dst=0, out=None, device=None, via_host=False, wait=True):
    """Super-resolve one (batch of) frame(s) split into horizontal strips across the process group.

    lr      : uint8 tensor [H,W,C] or [N,H,W,C] available on EVERY rank -- typically a HOST tensor (the input is
              1/scale^2 of the output): only this rank's band (strip + halo) is sliced out and sent to `device`
    compute : callable(band, band_row0, y0, y1, H) -> uint8 tensor with the output rows of LR rows
              [y0,y1) (``MuLUTEngine.pipeline_rows`` on the GPU)
    dst     : rank that receives the assembled frame; None = every rank
    out     : optional preallocated [.., H*scale, W*scale, C] tensor on the receiving rank(s)
    via_host: exchange through host memory (gloo rehearsals with device tensors; RCCL moves device memory directly)
    wait    : False -> return a PendingGather at once (overlap the exchange with the next frame's compute)
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
    if device is not None and band.device != torch.device(device):
        band = band.to(device, non_blocking=True)
    mine = compute(band, r0, y0, y1, H)
    if world == 1:
        return mine if wait else PendingGather(mine, [])
    bounds = strip_bounds(H, world)
    receiver = dst is None or rank == dst
    row_dim = 1 if batched else 0
    if receiver:
        shape = list(mine.shape)
        shape[row_dim] = H * scale
        if out is None:
            out = torch.empty(shape, dtype=mine.dtype, device=mine.device)
        elif list(out.shape) != shape:
            raise ValueError("out has shape %s, expected %s" % (tuple(out.shape), tuple(shape)))
        out.narrow(row_dim, y0 * scale, (y1 - y0) * scale).copy_(mine)        # own strip: one device copy
    # every frame of every strip is one contiguous block of rows of the output: point-to-point straight into place
    frames = mine.shape[0] if batched else 1
    send_buf = mine.cpu() if via_host else mine
    stage = torch.empty(out.shape, dtype=out.dtype, device="cpu") if (via_host and receiver) else None
    recv_into = stage if stage is not None else out
    ops = []
    peers = range(world) if dst is None else [dst]
    for p in peers:
        if p == rank:
            continue
        for n in range(frames):
            ops.append(dist.P2POp(dist.isend, send_buf[n] if batched else send_buf, p, group))
    if receiver:
        for src in range(world):
            if src == rank:
                continue
            a, b = bounds[src]
            for n in range(frames):
                view = (recv_into[n] if batched else recv_into).narrow(0, a * scale, (b - a) * scale)
                ops.append(dist.P2POp(dist.irecv, view, src, group))
    reqs = dist.batch_isend_irecv(ops) if ops else []
    if stage is not None:
        # rows received on the host go to the device at wait(); this rank's own rows are already there
        class _Back:
            def __init__(self, o, st, bnds, me):
                self.o, self.st, self.b, self.me = o, st, bnds, me

            def copy_(self, _src, non_blocking=False):
                for src, (a, b) in enumerate(self.b):
                    if src != self.me:
                        self.o.narrow(row_dim, a * scale, (b - a) * scale).copy_(self.st.narrow(row_dim, a * scale, (b - a) * scale))
        pending = PendingGather(out, reqs, stage, _Back(out, stage, bounds, rank))
    else:
        pending = PendingGather(out if receiver else None, reqs)
    if not wait:
        return pending
    return pending.wait()


def sr_frames(frames, compute, group=None):
    """Frame sharding: this rank's slice of a replicated batch, no collective. Returns (lo, hi, output)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = frame_slice(frames.shape[0], world, rank)
    return lo, hi, (compute(frames[lo:hi].contiguous()) if hi > lo else None)
