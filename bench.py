#!/usr/bin/env python3
"""bench.py -- MuLUT LUT-inference throughput on MI355X (BASELINE.json config 2).

One "step" = one pass of the hot path (2-stage sdy x4 cascade, fused per stage) over one batch of
synthetic 1080p LR frames already resident in HBM: LR 1080x1920x3 -> HR 4320x7680x3 per frame.
`value` = HR output Mpixels/s of the whole job (LR-input Mpix/s = value / 16).

  python bench.py [--gpus N --steps K --warmup W]          (N > 1: launched by torch.distributed.run)

Multi-GPU: frames are independent units (the reference fans images out with Pool(24).map,
sr/4_test_lut.py:257-259), so each rank processes its own batch with no data-path collective: weak
scaling.  The oracle is imported only by the cpu_baseline leg.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mulut_amd import MuLUTEngine, load_lut_dict  # noqa: E402
from mulut_amd.synth import natural_frames, noise_frames, real_frames  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
STAGES, MODES, SCALE = 2, "sdy", 4


REAL_PNG = os.path.join(ROOT, "tests", "golden", "DIV2K_LR_X4", "0001x4.png")


def make_batch(dist, frames, h, w, seed):
    if dist == "real":
        return np.ascontiguousarray(real_frames(frames, h, w, REAL_PNG, seed))
    gen = natural_frames if dist == "natural" else noise_frames
    base = gen(min(frames, 2), h, w, 3, seed)
    out = [np.roll(base[i % len(base)], (37 * (i // len(base)), 91 * (i // len(base))), axis=(0, 1))
           for i in range(frames)]
    return np.ascontiguousarray(np.stack(out))


def timed_steps(eng, x, out, steps, dist_on, step_fn=None):
    if dist_on:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        if step_fn is None:
            eng.pipeline(x, out=out)
        else:
            step_fn()
    torch.cuda.synchronize()
    if dist_on:
        torch.distributed.barrier()
    return time.perf_counter() - t0


def cpu_baseline(eng, luts, frame_u8, crop):
    """The reference's CPU path, as ported in oracle/np_port.py (same operation structure: 16 corner
    gathers + 24 masked cases in float), timed single-process on a crop x crop window of the same
    frame; its output doubles as a parity check of the GPU result for that window."""
    from oracle import c_oracle, np_port
    win = np.ascontiguousarray(frame_u8[:crop, :crop])
    f32 = {k: v.astype(np.float32) for k, v in luts.items()}
    t0 = time.perf_counter()
    ref = np_port.run_stages(f32, STAGES, MODES, SCALE, win)
    dt = time.perf_counter() - t0
    gpu = eng.pipeline(torch.from_numpy(win).cuda()).cpu().numpy()
    t1 = time.perf_counter()
    ref_c = c_oracle.pipeline(luts, STAGES, MODES, SCALE, win)
    dt_c = time.perf_counter() - t1
    return {
        "value": round(crop * crop * SCALE * SCALE / dt / 1e6, 5), "unit": "Mpix/s", "cores": 1, "kind": "port",
        "sample": "oracle/np_port.py (NumPy port of sr/4_test_lut.py, float, 16 gathers + 24 masked cases) on the "
                  "top-left %dx%dx3 window of frame 0, 2-stage sdy x4, %.1f s" % (crop, crop, dt),
        "matches_gpu": bool(np.array_equal(ref, gpu) and np.array_equal(ref_c, gpu)),
        "c_oracle_value": round(crop * crop * SCALE * SCALE / dt_c / 1e6, 4),
        "cpu_model": _cpu_model(), "host_cores": os.cpu_count(),
    }


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def load_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        rec = json.load(open(path))
        if rec.get("workload") == workload:
            return rec.get("final_stage_bytes_per_launch")
    except (OSError, ValueError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=8, help="LR frames per GPU per step")
    ap.add_argument("--lr-h", type=int, default=1080)
    ap.add_argument("--lr-w", type=int, default=1920)
    ap.add_argument("--dist", choices=["natural", "noise", "real"], default="natural",
                    help="natural: smooth synthetic field (headline); noise: uniform random bytes (worst case); "
                         "real: the DIV2K LR photo the reference ships, mirror-tiled")
    ap.add_argument("--cpu-crop", type=int, default=512, help="window edge for the CPU baseline (0 = skip)")
    ap.add_argument("--skip-other", action="store_true", help="do not also time the other input distribution")
    ap.add_argument("--shard", choices=["frames", "strips"], default="frames",
                    help="frames: each GPU owns whole frames, no collective (default); strips: every frame is cut "
                         "into one strip per GPU (+4-row halo) and the HR strips are gathered on rank 0 over RCCL")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    # one rank per GPU; on a box with fewer GPUs than ranks (1-GPU rehearsal of the N > 1 path with
    # MULUT_BENCH_BACKEND=gloo) ranks share devices
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("MULUT_BENCH_BACKEND", "nccl")      # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            torch.distributed.init_process_group(backend)

    luts = load_lut_dict(os.path.join(ROOT, "tests", "golden", "luts"), STAGES, MODES, SCALE, 4, "LUT_ft")
    eng = MuLUTEngine(local).configure(STAGES, MODES, SCALE, 4).set_lut_dict(luts)
    F, H, W = args.frames, args.lr_h, args.lr_w
    eng.reserve(F, H, W, 3)
    host = make_batch(args.dist, F, H, W, seed=rank)
    x = torch.from_numpy(host).cuda()
    out = torch.empty((F, H * SCALE, W * SCALE, 3), dtype=torch.uint8, device=x.device)

    step_fn = None
    if args.shard == "strips" and dist_on:
        from mulut_amd.dist import sr_strips
        # strong scaling over one batch: every rank holds the same F frames and produces 1/world of the rows
        torch.manual_seed(0)
        x = torch.from_numpy(make_batch(args.dist, F, H, W, seed=0)).cuda()

        def step_fn():
            return sr_strips(x, lambda band, r0, y0, y1, hh: eng.pipeline_rows(band, r0, y0, y1, hh), SCALE, eng.halo,
                             dst=0)
    for _ in range(args.warmup):
        step_fn() if step_fn else eng.pipeline(x, out=out)
    elapsed = timed_steps(eng, x, out, args.steps, dist_on, step_fn)
    t = torch.tensor([elapsed], dtype=torch.float64, device=x.device)
    if dist_on:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(t.item())

    strips = step_fn is not None
    hr_pix = (1 if strips else world) * F * H * SCALE * W * SCALE * args.steps
    value = hr_pix / elapsed / 1e6

    # per-kernel device time: HIP events recorded around each stage launch on the launch stream
    eng.set_stage_timing(True)
    per_stage = []
    for _ in range(args.steps):
        eng.pipeline(x, out=out)
        per_stage.append(eng.last_stage_ms())
    eng.set_stage_timing(False)
    ms_stage = np.mean(np.asarray(per_stage), axis=0)

    # the other input distribution, same number of steps (reported beside the headline)
    others = {}
    if not args.skip_other:
        for other in ("natural", "noise", "real"):
            if other == args.dist or (other == "real" and not os.path.exists(REAL_PNG)):
                continue
            x2 = torch.from_numpy(make_batch(other, F, H, W, seed=rank)).cuda()
            for _ in range(2):
                eng.pipeline(x2, out=out)
            el2 = timed_steps(eng, x2, out, args.steps, dist_on)
            t2 = torch.tensor([el2], dtype=torch.float64, device=x.device)
            if dist_on:
                torch.distributed.all_reduce(t2, op=torch.distributed.ReduceOp.MAX)
            others[other] = world * F * H * SCALE * W * SCALE * args.steps / float(t2.item()) / 1e6
            del x2

    if rank == 0:
        sites = F * H * W * 3                                   # LR samples per launch
        lut_final = 3 * 83521 * 16
        alg_k2 = sites * (1 + SCALE * SCALE) + lut_final        # 17 B per LR sample + tables once
        achieved = alg_k2 / (ms_stage[-1] * 1e-3) / 1e9
        workload = "2-stage sdy x4, %d x LR %dx%dx3 -> HR %dx%dx3 per GPU per step, D-%s" % (
            F, H, W, H * SCALE, W * SCALE, args.dist)
        rec = {
            "metric": "Mpixels/sec SR-x4 2-stage sdy LUT inference (HR output pixels; LR-in = value/16)",
            "value": round(value, 2), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong" if strips else "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload, "frames_per_gpu": F, "lr": [H, W, 3], "stages": STAGES, "modes": MODES,
                       "scale": SCALE, "luts": "shipped sr_x2sdy fine-tuned int8 tables", "parallelism":
                       ("each frame cut into %d strips (+4-row halo), RCCL gather of the HR strips on rank 0" % world)
                       if strips else ("frames sharded over %d GPU(s), no collective" % world),
                       **{"value_D-%s" % k: round(v, 2) for k, v in others.items()}},
            "roofline": {"bound": "hbm", "kernel": eng.kernel_name(True), "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": load_traffic(workload), "algorithmic_bytes_per_launch": alg_k2,
                         "kernel_ms": round(float(ms_stage[-1]), 4),
                         "stage_ms": [round(float(v), 4) for v in ms_stage],
                         "pipeline_frac": round((alg_k2 + 3 * 83536) / (float(ms_stage.sum()) * 1e-3) / 1e9
                                                / HBM_PEAK_GBS, 5)},
        }
        if world == 1 and args.cpu_crop > 0:
            rec["cpu_baseline"] = cpu_baseline(eng, luts, host[0], min(args.cpu_crop, H, W))
        print(json.dumps(rec))
    if dist_on:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
