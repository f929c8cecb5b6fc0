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
  Who receives is a choice: one root (its ingress -- 7 xGMI links -- bounds the exchange: fine for one frame, ~1.7x at
  most on 8 GPUs for a stream of them), every rank, or -- for batches -- ``dst="rotate"``: frame n is assembled on rank
  n % world, an all-to-all in which every link carries 1/world^2 of the batch; ``dst="none"`` leaves the strips sharded.

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
    """Handle of an asynchronous strip exchange: ``wait()`` returns the assembled tensor (None on a rank that receives
    nothing).  ``frames`` lists the batch indices of the frames held in ``out`` (all of them, except in rotate mode)."""

    def __init__(self, out, reqs, copies=(), frames=None):
        self.out, self._reqs, self._copies, self.frames = out, reqs, list(copies), frames

    def wait(self):
        for r in self._reqs:
            r.wait()
        self._reqs = []
        for dst_view, src_view in self._copies:      # host-staged receives (gloo rehearsal): back to the device
            dst_view.copy_(src_view)
        self._copies = []
        return self.out


def rotate_owner(frame, world):
    """Rank that assembles `frame` of a batch in rotate mode."""
    return frame % world


def sr_strips(lr, compute, scale, halo, group=None, dst=0, out=None, device=None, via_host=False, wait=True):
    """Super-resolve one (batch of) frame(s) split into horizontal strips across the process group.

    lr      : uint8 tensor [H,W,C] or [N,H,W,C] available on EVERY rank -- typically a HOST tensor (the input is
              1/scale^2 of the output): only this rank's band (strip + halo) is sliced out and sent to `device`
    compute : callable(band, band_row0, y0, y1, H) -> uint8 tensor with the output rows of LR rows
              [y0,y1) (``MuLUTEngine.pipeline_rows`` on the GPU)
    dst     : who ends up with whole frames --
              an int   : that rank receives every frame (a single root: its ingress bounds the exchange);
              None     : every rank receives every frame (all-gather);
              "rotate" : batched input only: frame n is assembled on rank n % world -- an all-to-all of HR strips in which
                         every rank receives (world-1)/world of the frames it owns and every link carries 1/world^2 of the
                         batch: the mode that scales;
              "none"   : no exchange at all: the strips stay where they were computed (returns this rank's rows).
    out     : optional preallocated result on the receiving rank(s): [.., H*scale, W*scale, C]; in rotate mode
              [frames this rank owns, H*scale, W*scale, C]
    via_host: exchange through host memory (gloo rehearsals with device tensors; RCCL moves device memory directly)
    wait    : False -> return a PendingGather at once (overlap the exchange with the next batch's compute)
    returns the assembled tensor (or the PendingGather), None on a rank that receives nothing.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    batched = lr.dim() == 4
    H = lr.shape[1] if batched else lr.shape[0]
    if H < world:
        raise ValueError("fewer image rows than ranks")
    if dst == "rotate" and not batched:
        raise ValueError("rotate mode needs a batch of frames [N,H,W,C]")
    r0, r1, y0, y1 = strip_band(H, world, rank, halo)
    band = (lr[:, r0:r1] if batched else lr[r0:r1]).contiguous()
    if device is not None and band.device != torch.device(device):
        band = band.to(device, non_blocking=True)
    mine = compute(band, r0, y0, y1, H)
    nframes = mine.shape[0] if batched else 1
    if dst == "none":
        return mine if wait else PendingGather(mine, [], frames=list(range(nframes)))
    # frames this rank assembles, and who assembles frame n
    if dst == "rotate":
        owners = [[rotate_owner(n, world)] for n in range(nframes)]
    elif dst is None:
        owners = [list(range(world))] * nframes
    else:
        owners = [[int(dst)]] * nframes
    held = [n for n in range(nframes) if rank in owners[n]]
    if world == 1:
        if out is not None:
            out.copy_(mine)
            mine = out
        return mine if wait else PendingGather(mine, [], frames=held)
    bounds = strip_bounds(H, world)
    row_dim = 1 if batched else 0
    if held:
        shape = list(mine.shape)
        shape[row_dim] = H * scale
        if dst == "rotate":
            shape[0] = len(held)
        if out is None:
            out = torch.empty(shape, dtype=mine.dtype, device=mine.device)
        elif list(out.shape) != shape:
            raise ValueError("out has shape %s, expected %s" % (tuple(out.shape), tuple(shape)))
    else:
        out = None
    slot = {n: k for k, n in enumerate(held)} if dst == "rotate" else {n: n for n in held}

    def frame_view(t, n):      # frame n of a result tensor (the whole tensor when the input was one image)
        return t[slot[n]] if batched else t

    # every strip of every frame is one contiguous block of rows of its destination: point-to-point straight into place
    send_buf = mine.cpu() if via_host else mine
    stage = torch.empty(out.shape, dtype=out.dtype, device="cpu") if (via_host and out is not None) else None
    recv_into = stage if stage is not None else out
    ops, copies = [], []
    for n in range(nframes):
        for p in owners[n]:
            if p != rank:
                ops.append(dist.P2POp(dist.isend, send_buf[n] if batched else send_buf, p, group))
    for n in held:
        frame_view(out, n).narrow(0, y0 * scale, (y1 - y0) * scale).copy_(mine[n] if batched else mine)      # own strip: one device copy
        for src in range(world):
            if src == rank:
                continue
            a, b = bounds[src]
            view = frame_view(recv_into, n).narrow(0, a * scale, (b - a) * scale)
            ops.append(dist.P2POp(dist.irecv, view, src, group))
            if stage is not None:      # rows received on the host go to the device at wait()
                copies.append((frame_view(out, n).narrow(0, a * scale, (b - a) * scale), view))
    reqs = dist.batch_isend_irecv(ops) if ops else []
    pending = PendingGather(out, reqs, copies, held)
    if not wait:
        return pending
    return pending.wait()


def sr_frames(frames, compute, group=None):
    """Frame sharding: this rank's slice of a replicated batch, no collective. Returns (lo, hi, output)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = frame_slice(frames.shape[0], world, rank)
    return lo, hi, (compute(frames[lo:hi].contiguous()) if hi > lo else None)
