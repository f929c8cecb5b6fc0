#!/usr/bin/env python3
"""bench.py -- MuLUT LUT-inference throughput on MI355X (BASELINE.json config 2, + configs 3 and 5 on request).

One "step" = one pass of the hot path (2-stage sdy x4 cascade, one fused kernel per stage) over one batch of
synthetic 1080p LR frames already resident in HBM: LR 1080x1920x3 -> HR 4320x7680x3 per frame.
`value` = HR output Mpixels/s of the whole job (LR-input Mpix/s = value / 16).

  python bench.py [--gpus N --steps K --warmup W]

N > 1 launches N ranks itself (python -m torch.distributed.run ... bench.py, before this process touches the GPU) unless
it already runs under torchrun (WORLD_SIZE set).  Multi-GPU: frames are independent units (the reference fans images out
with Pool(24).map, sr/4_test_lut.py:257-259), so each rank processes its own batch with no data-path collective: weak
scaling -- that is `value`.  The north star's other sharding -- each frame cut into one strip per GPU (+halo), the HR
strips gathered on rank 0 over RCCL -- is timed in the same run on LR 2160x3840 batches (config 3) and reported under
config["strips_gather"].  The oracle is imported only by the cpu_baseline leg.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s
SIMDS, CLOCK_GHZ = 1024, 2.4  # 256 CUs x 4 SIMDs; one VALU wave-instruction per SIMD per 4 cycles at best
LDS_PEAK_GBS = 256 * 256 * CLOCK_GHZ   # 256 B/clk/CU for ds_read_b64/b128 (MI355X_MICROARCH.md, LDS)
STAGES, MODES, SCALE = 2, "sdy", 4
REAL_PNG = os.path.join(ROOT, "tests", "golden", "DIV2K_LR_X4", "0001x4.png")
LUT_DIR = os.path.join(ROOT, "tests", "golden", "luts")


# ---------------------------------------------------------------------------------------------------------------
# self-launch (N > 1 without torchrun): the parent never initialises the GPU
# ---------------------------------------------------------------------------------------------------------------
def self_launch(args):
    from mulut_amd import _native
    _native.build()                      # once, here: ranks must not race on the library
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MULUT_NO_BUILD="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1])
    else:
        sys.stderr.write(r.stdout[-4000:])
    sys.exit(r.returncode if r.returncode else (0 if lines else 1))


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1): the oracle, timed -- never the thing measured as `value`
# ---------------------------------------------------------------------------------------------------------------
_POOL_LUTS = None


def _pool_tile(job):
    from oracle import np_port
    img, y0, y1, x0, x1, r0, c0 = job          # img = tile + halo; [y0:y1, x0:x1] = the tile inside it
    out = np_port.run_stages(_POOL_LUTS, STAGES, MODES, SCALE, img)
    return r0, c0, out[y0 * SCALE:y1 * SCALE, x0 * SCALE:x1 * SCALE]


def cpu_baseline_host(luts, frame_u8, crop):
    """Timed before the GPU is touched (the Pool forks).  (i) oracle/np_port.py -- the reference's operation structure in
    NumPy -- single process on a crop x crop window; (ii) the same port with a process Pool over 128x128 tiles (+4-pixel
    halo, the reference fans out with Pool(24).map) of the whole frame on every host core; (iii) the C oracle, one core."""
    global _POOL_LUTS
    from multiprocessing import get_context
    from oracle import c_oracle, np_port
    f32 = {k: v.astype(np.float32) for k, v in luts.items()}
    win = np.ascontiguousarray(frame_u8[:crop, :crop])
    t0 = time.perf_counter()
    ref_win = np_port.run_stages(f32, STAGES, MODES, SCALE, win)
    dt1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    ref_c = c_oracle.pipeline(luts, STAGES, MODES, SCALE, win)
    dtc = time.perf_counter() - t0
    H, W = frame_u8.shape[:2]
    T, halo = 64, 2 * STAGES      # 17 x 30 = 510 tiles of a 1080p frame: every host core gets work
    jobs = []
    for r0 in range(0, H, T):
        for c0 in range(0, W, T):
            ya, yb = max(0, r0 - halo), min(H, r0 + T + halo)
            xa, xb = max(0, c0 - halo), min(W, c0 + T + halo)
            jobs.append((np.ascontiguousarray(frame_u8[ya:yb, xa:xb]), r0 - ya, min(H, r0 + T) - ya, c0 - xa, min(W, c0 + T) - xa, r0, c0))
    cores = max(1, min(os.cpu_count() or 1, len(jobs)))
    _POOL_LUTS = f32
    full = np.empty((H * SCALE, W * SCALE, frame_u8.shape[2]), np.uint8)
    t0 = time.perf_counter()
    with get_context("fork").Pool(cores) as pool:
        for r0, c0, tile in pool.imap_unordered(_pool_tile, jobs, chunksize=1):
            full[r0 * SCALE:r0 * SCALE + tile.shape[0], c0 * SCALE:c0 * SCALE + tile.shape[1]] = tile
    dtp = time.perf_counter() - t0
    rec = {
        "value": round(H * W * SCALE * SCALE / dtp / 1e6, 4), "unit": "Mpix/s", "cores": cores, "kind": "port",
        "sample": "oracle/np_port.py (NumPy port of sr/4_test_lut.py: float, 16 corner gathers + 24 masked cases per pass) on "
                  "frame 0 (LR %dx%dx3), multiprocessing Pool over %d tiles of 64x64 (+%d-pixel halo) on %d processes, %.1f s"
                  % (H, W, len(jobs), halo, cores, dtp),
        "single_core_value": round(crop * crop * SCALE * SCALE / dt1 / 1e6, 5),
        "single_core_sample": "the same port, one process, top-left %dx%dx3 window of frame 0, %.1f s" % (crop, crop, dt1),
        "c_oracle_single_core_value": round(crop * crop * SCALE * SCALE / dtc / 1e6, 4),
        "cpu_model": _cpu_model(), "host_cores": os.cpu_count(),
        # (the window is a stand-alone image: its last 2 x stages columns / rows see its own edge replication, the frame's do not)
        "tiles_match_window": bool(np.array_equal(full[:(crop - halo) * SCALE, :(crop - halo) * SCALE], ref_win[:(crop - halo) * SCALE, :(crop - halo) * SCALE])
                                   and np.array_equal(ref_c, ref_win)),
    }
    return rec, full


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def load_profile_json(name, workload=None):
    """Profiler-derived per-launch figures (profiles/<name>): valid only for the build they were measured on."""
    from mulut_amd import _native
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", name)))
    except (OSError, ValueError):
        return None
    if rec.get("source_hash") != _native.source_hash():
        return None
    if workload is not None and rec.get("workload") != workload:
        return None
    return rec


ROLLED_BATCH = False      # --batch rolled: rounds 1-2's batch (two draws, the other frames rolled copies of them)
MAX_DRAWS = 32            # distinct draws of a distribution per batch


def make_batch(dist, frames, h, w, seed):
    """`frames` frames of the distribution.  D-natural / D-noise: every frame its own draw (SURVEY section 8d defines a frame as one
    draw of the field).  Rounds 1 and 2 drew two frames and rolled them by (37 k, 91 k) pixels for the rest: the wrap-around
    seams of a rolled smooth field are edges the field does not have (they cost the routed kernels ~7 % on D-natural);
    `--batch rolled` still makes that batch."""
    from mulut_amd.synth import natural_frames, noise_frames, real_frames
    if dist == "real":
        return np.ascontiguousarray(real_frames(frames, h, w, REAL_PNG, seed))
    gen = natural_frames if dist == "natural" else noise_frames
    if not ROLLED_BATCH:
        # at most MAX_DRAWS draws are generated (half a second of host time each at 1080p); a larger batch repeats them in order
        base = gen(min(frames, MAX_DRAWS), h, w, 3, seed)
        return np.ascontiguousarray(np.concatenate([base] * (-(-frames // len(base))))[:frames])
    base = gen(min(frames, 2), h, w, 3, seed)
    out = [np.roll(base[i % len(base)], (37 * (i // len(base)), 91 * (i // len(base))), axis=(0, 1))
           for i in range(frames)]
    return np.ascontiguousarray(np.stack(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=128, help="LR frames per GPU per step (128 = 4 x 32 draws: a 20-step timed region lasts ~0.6 s, long enough for an external busy sampler to see it)")
    ap.add_argument("--launch-frames", type=int, default=32, help="frames per mulut_pipeline call (a step makes frames / launch-frames calls)")
    ap.add_argument("--batch", choices=("draws", "rolled"), default="draws", help="draws: every frame its own draw of the distribution; rolled: rounds 1-2's batch")
    ap.add_argument("--lr-h", type=int, default=1080)
    ap.add_argument("--lr-w", type=int, default=1920)
    ap.add_argument("--dist", choices=["natural", "noise", "real"], default="natural",
                    help="natural: smooth synthetic field (headline); noise: uniform random bytes (worst case); "
                         "real: the DIV2K LR photo the reference ships, mirror-tiled")
    ap.add_argument("--cpu-crop", type=int, default=512, help="window edge for the single-core CPU baseline (0 = skip the CPU legs)")
    ap.add_argument("--skip-other", action="store_true", help="do not also time the other input distributions")
    ap.add_argument("--skip-strips", action="store_true", help="N > 1: do not also time the strips + gather sharding (config 3)")
    ap.add_argument("--strip-frames", type=int, default=4, help="LR 2160x3840 frames per step of the strips + gather leg")
    ap.add_argument("--strips-timeout", type=int, default=240, help="seconds the strips + gather leg may take before the headline is printed without it")
    ap.add_argument("--config", type=int, default=2, choices=[2, 4, 5],
                    help="2: headline (2-stage sdy x4); 4: LUT fine-tune step (fwd + bwd + Adam, bs 256 x 1x48x48); "
                         "5: deep cascade (4-stage sdy x2, seeded synthetic tables), eager vs hipGraph")
    args = ap.parse_args()
    global ROLLED_BATCH
    ROLLED_BATCH = args.batch == "rolled"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d: reporting n_gpus = %d" % (args.gpus, world, world), file=sys.stderr)

    from mulut_amd import load_lut_dict
    luts = load_lut_dict(LUT_DIR, STAGES, MODES, SCALE, 4, "LUT_ft")
    F, H, W = args.frames, args.lr_h, args.lr_w
    host = make_batch(args.dist, F, H, W, seed=rank)

    # CPU legs first: the Pool forks, which must happen before this process initialises the GPU
    cpu_rec = cpu_full = None
    if world == 1 and args.cpu_crop > 0 and args.config == 2:
        cpu_rec, cpu_full = cpu_baseline_host(luts, host[0], min(args.cpu_crop, H, W))

    import torch
    from mulut_amd import MuLUTEngine
    local = local % max(1, torch.cuda.device_count())     # 1-GPU rehearsals of the N > 1 path (MULUT_BENCH_BACKEND=gloo) share the device
    torch.cuda.set_device(local)
    backend = os.environ.get("MULUT_BENCH_BACKEND", "nccl")      # "nccl" IS RCCL on ROCm
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            torch.distributed.init_process_group(backend)

    def barrier_sync():
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps):
        barrier_sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        if dist_on:
            torch.distributed.barrier()
        el = time.perf_counter() - t0
        if dist_on:
            t = torch.tensor([el], dtype=torch.float64, device="cuda")
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            el = float(t.item())
        return el

    if args.config in (4, 5):
        rec = (config5 if args.config == 5 else config4)(args, world, rank, local, timed)
        if rank == 0:
            print(json.dumps(rec), flush=True)
        return

    # A step takes its F frames through the library in calls of LF frames: one mulut_pipeline call is one launch of every kernel, and the
    # device work lists of the detailed-tile path address at most 256 MB of stage input per launch (43 frames of 1080p: DESIGN section 6).
    LF = min(F, args.launch_frames)
    if F % LF:
        raise SystemExit("--frames must be a multiple of --launch-frames")
    calls = F // LF
    eng = MuLUTEngine(local).configure(STAGES, MODES, SCALE, 4).set_lut_dict(luts)
    eng.reserve(LF, H, W, 3)
    x = torch.from_numpy(host).cuda()
    out = torch.empty((F, H * SCALE, W * SCALE, 3), dtype=torch.uint8, device=x.device)

    def run_step(src, dst=out, each=None):
        for k in range(calls):
            eng.pipeline(src[k * LF:(k + 1) * LF], out=dst[k * LF:(k + 1) * LF])
            if each:
                each()

    for _ in range(args.warmup):
        run_step(x)
    elapsed = timed(lambda: run_step(x), args.steps)
    value = world * F * H * SCALE * W * SCALE * args.steps / elapsed / 1e6
    ms_step = elapsed / args.steps * 1e3

    if cpu_rec is not None:
        cpu_rec["matches_gpu"] = bool(np.array_equal(out[0].cpu().numpy(), cpu_full))
        del cpu_full

    # per-stage and per-dominant-kernel device time: HIP events the library records on the launch stream
    eng.set_stage_timing(True)
    per_stage, per_kernel = [], []
    for _ in range(max(2, args.steps // calls)):
        run_step(x, each=lambda: (per_stage.append(eng.last_stage_ms()), per_kernel.append(eng.last_kernel_ms())))      # per call of LF frames
    eng.set_stage_timing(False)
    ms_stage = np.mean(np.asarray(per_stage), axis=0)
    ms_kernel = np.mean(np.asarray(per_kernel), axis=0)

    others = {}
    if not args.skip_other:
        for other in ("natural", "noise", "real"):
            if other == args.dist or (other == "real" and not os.path.exists(REAL_PNG)):
                continue
            x2 = torch.from_numpy(make_batch(other, F, H, W, seed=rank)).cuda()
            for _ in range(2):
                run_step(x2)
            el2 = timed(lambda: run_step(x2), max(2, args.steps // 2))
            others[other] = world * F * H * SCALE * W * SCALE * max(2, args.steps // 2) / el2 / 1e6
            del x2
        if args.dist == "natural" and not ROLLED_BATCH:
            # the batch definition of rounds 1-2 (two draws + rolled copies: their wrap seams are edges the field does not have), so that the
            # default line carries the number that compares with those rounds
            ROLLED_BATCH = True
            x2 = torch.from_numpy(make_batch("natural", F, H, W, seed=rank)).cuda()
            ROLLED_BATCH = False
            for _ in range(2):
                run_step(x2)
            el2 = timed(lambda: run_step(x2), max(2, args.steps // 2))
            others["natural_rolled_batch_of_rounds_1_2"] = world * F * H * SCALE * W * SCALE * max(2, args.steps // 2) / el2 / 1e6
            del x2

    if rank == 0:
        sites = LF * H * W * 3                                  # LR samples per launch
        lut_bytes = 3 * 83521 * (1 + SCALE * SCALE)             # every table read once per launch: SURVEY 8(d) LUT_bytes
        alg = calls * (sites * (1 + SCALE * SCALE) + lut_bytes)           # 17 B per LR sample + tables
        alg_k2 = sites * (1 + SCALE * SCALE) + 3 * 83521 * SCALE * SCALE
        workload = "2-stage sdy x4, %d x LR %dx%dx3 -> HR %dx%dx3 per GPU per step%s, D-%s" % (
            F, H, W, H * SCALE, W * SCALE, "" if calls == 1 else " in %d launches of %d frames" % (calls, LF), args.dist)
        achieved = alg / (ms_step * 1e-3) / 1e9
        k2_ms = float(ms_kernel[-1]) if ms_kernel[-1] > 0 else float(ms_stage[-1])
        counters = load_profile_json("kernel_counters.json", workload if world == 1 else None) if world == 1 else None
        traffic = load_profile_json("hbm_traffic.json", workload) if world == 1 else None
        secondary = None
        issue = load_profile_json("valu_issue.json")
        if counters:
            k2 = counters["final_stage_kernel"]
            scale_f = LF / float(counters.get("frames", LF))      # the counters were collected at counters["frames"] frames per launch
            insts = k2["valu_wave_insts_per_launch"] * scale_f
            rate = insts / (k2_ms * 1e-3) / 1e9
            # peak: what the kernel's OWN instruction stream sustains per SIMD with nothing else in the way (tools/ubench/gen_stream_ubench.py:
            # the hot loop's VALU instructions, same registers and modifiers, 4 waves per SIMD, in-kernel clock) -- and, beside it, the
            # guide's 2 cycles per wave64 VALU instruction
            cpi = issue["stream_cycles_per_valu_inst"] if issue else None
            clk = issue["stream_clock_ghz"] if issue else None
            peak_stream = SIMDS * clk / cpi if issue else None
            secondary = {
                "bound": "valu_issue", "kernel": k2["name"], "achieved": round(rate, 2), "unit": "G wave-instructions/s",
                # peak / frac: the hardware figure (MI355X_MICROARCH.md: a wave64 VALU instruction issues over 2 cycles on a SIMD-32, 2.4 GHz)
                "peak": round(SIMDS * CLOCK_GHZ / 2, 1), "frac": round(rate / (SIMDS * CLOCK_GHZ / 2), 4),
                "peak_how": "1024 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md, Wave scheduling)",
                # beside it: what the kernel's OWN instruction stream sustains as a microbenchmark at the kernel's occupancy (packed 16-bit and
                # three-operand instructions issue at 3.0-3.24 cycles per SIMD at 4 waves, profiles/r01_ubench_valu_issue_cost.txt)
                "stream_peak": round(peak_stream, 2) if issue else None, "frac_of_stream_peak": round(rate / peak_stream, 4) if issue else None,
                "stream_peak_how": ("the kernel's own VALU stream as a microbenchmark: %.2f cycles per instruction per SIMD at %.2f GHz in-kernel clock, 4 waves per SIMD "
                                    "(profiles/valu_issue.json, same source hash)" % (cpi, clk)) if issue else "profiles/valu_issue.json missing or stale",
                "how": "SQ_INSTS_VALU per launch (rocprofv3 --pmc, profiles/kernel_counters.json, same source hash) / the kernel's event-timed duration in this run",
                "valu_insts_per_pass": round(insts / (sites / 64 * 12), 1),
                "lds_gather": {"achieved": round(k2.get("lds_bytes_per_launch", 0) * scale_f / (k2_ms * 1e-3) / 1e9, 1), "peak": round(LDS_PEAK_GBS, 1),
                               "unit": "GB/s", "bytes_per_lr_sample": 12 * 5 * 32,
                               "note": "row gathers of the final stage: 12 passes x 5 rows x 32 B (16-bit fields) per LR sample"},
            }
        rec = {
            "metric": "Mpixels/sec SR-x4 2-stage sdy LUT inference (HR output pixels; LR-in = value/16)",
            "value": round(value, 2), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload, "batch": "rolled copies of two draws (rounds 1-2)" if ROLLED_BATCH else ("every frame its own draw" if F <= MAX_DRAWS else "%d draws, repeated in order to %d frames" % (MAX_DRAWS, F)), "frames_per_gpu": F, "frames_per_launch": LF, "lr": [H, W, 3], "stages": STAGES, "modes": MODES,
                       "scale": SCALE, "luts": "shipped sr_x2sdy fine-tuned int8 tables",
                       "parallelism": "frames sharded over %d GPU(s), no collective" % world,
                       **{"value_D-%s" % k: round(v, 2) for k, v in others.items()}},
            "roofline": {"bound": "hbm", "kernel": "fused 2-stage pipeline: %s | %s" % (eng.kernel_name(False), eng.kernel_name(True)),
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": traffic.get("pipeline_bytes_per_step") * calls if traffic else None,      # (the JSON holds bytes per launch of every kernel)
                         "algorithmic_bytes_per_step": alg,
                         "stage_ms": [round(float(v), 4) for v in ms_stage],
                         "dominant_kernel": {"name": "stage_tube2_kernel<rgb>" if "tube2" in eng.kernel_name(True) else "stage_tube_kernel<rgb>" if "tube" in eng.kernel_name(True) else eng.kernel_name(True),
                                             "ms": round(k2_ms, 4), "algorithmic_bytes_per_launch": alg_k2,
                                             "achieved": round(alg_k2 / (k2_ms * 1e-3) / 1e9, 2),
                                             "kernel_frac": round(alg_k2 / (k2_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                             "traffic": traffic.get("final_stage_bytes_per_launch") if traffic else None},
                         "first_stage_kernel_ms": round(float(ms_kernel[0]), 4),
                         "secondary": secondary},
        }
        if cpu_rec is not None:
            rec["cpu_baseline"] = cpu_rec

    # config 3: LR 2160x3840 frames.  N = 1: whole frames on the one GPU.  N > 1: every frame cut into one strip per rank
    # (+halo), cascade per strip, HR strips sent point to point (RCCL) into the frame on rank 0; the gather of batch k
    # overlaps the compute of batch k + 1.  The headline record is complete before this leg starts and cannot be lost to
    # it: an exception is recorded, and a watchdog prints the record without the leg and ends every rank with exit code 3 if the exchange
    # has not come back in time.
    strips = None
    if not args.skip_strips:
        import threading
        lock, state = threading.Lock(), {"printed": False}

        def give_up():
            with lock:
                if state["printed"]:
                    return
                state["printed"] = True
                if rank == 0:
                    rec["config"]["strips_gather"] = {"error": "no result within %d s; headline printed without it" % args.strips_timeout}
                    print(json.dumps(rec), flush=True)
            os._exit(3)      # a hung exchange is a failure: the record is out, the run must not read as green
        timer = threading.Timer(args.strips_timeout, give_up)
        timer.daemon = True
        timer.start()
        try:
            strips = config3(args, eng, world, rank, dist_on, backend, timed)
        except Exception as exc:      # the headline line must not depend on this leg
            strips = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
        timer.cancel()
        with lock:
            if state["printed"]:      # the watchdog fired while the leg was finishing: it owns the output
                return
            state["printed"] = True
    extra = None
    if not args.skip_other and world == 1:
        del x, out
        torch.cuda.empty_cache()
        extra = other_configs(args, world, rank, local, timed)
    if rank == 0:
        if strips:
            rec["config"]["strips_gather"] = strips
        if extra:
            rec["config"]["other_configs"] = extra
        if dist_on:
            # what a scaling record must answer at the top level: did the collective backend see N ranks, and what crossed the links
            rec["backend"] = "rccl (torch.distributed 'nccl')" if backend == "nccl" else "%s (rehearsal: strips travel through host memory)" % backend
            rec["ranks_in_group"] = torch.distributed.get_world_size()
            rec["scaling_note"] = ("value = frames sharded over the ranks, no data-path collective (weak); strips_rotate_value = config 3, every LR 2160x3840 "
                                   "frame cut into one strip per rank (+halo), HR strips exchanged point to point, frame n assembled on rank n mod N (strong)")
            if strips and "rotate" in strips:
                rec["strips_rotate_value"] = strips["rotate"]["value"]
                rec["strips_single_root_value"] = strips["single_root"]["value"]
                rec["exchanged_bytes_per_step"] = strips["rotate"]["exchanged_bytes_per_step"]
                rec["max_bytes_into_one_rank_per_step"] = strips["rotate"]["max_bytes_into_one_rank_per_step"]
        print(json.dumps(rec), flush=True)
    if dist_on:
        torch.distributed.destroy_process_group()


def config3(args, eng, world, rank, dist_on, backend, timed):
    """LR 2160x3840 batches (BASELINE config 3)."""
    import torch
    from mulut_amd.dist import sr_strips
    Fs, Hs, Ws = args.strip_frames, 2160, 3840
    steps = max(2, args.steps // 4)
    frames = make_batch(args.dist, Fs, Hs, Ws, seed=0)                  # the same batch on every rank (host memory)
    if world == 1:
        x = torch.from_numpy(frames).cuda()
        out = torch.empty((Fs, Hs * SCALE, Ws * SCALE, 3), dtype=torch.uint8, device="cuda")
        eng.pipeline(x, out=out)
        el = timed(lambda: eng.pipeline(x, out=out), steps)
        return {"workload": "config 3 on one GPU: %d x LR %dx%dx3 whole frames per step" % (Fs, Hs, Ws), "n_gpus": 1,
                "value": round(Fs * Hs * SCALE * Ws * SCALE * steps / el / 1e6, 2), "unit": "Mpix/s", "ms_per_step": round(el / steps * 1e3, 3)}
    via_host = backend != "nccl"
    import time as _t

    def compute(band, r0, y0, y1, hh):
        return eng.pipeline_rows(band, r0, y0, y1, hh)

    def leg(mode, nfr):
        """nfr frames per step; mode "root": every HR strip goes to rank 0; "rotate": frame n is assembled on rank n % world."""
        lr = torch.from_numpy(make_batch(args.dist, nfr, Hs, Ws, seed=0) if nfr != Fs else frames)
        if mode == "root":
            held = nfr if rank == 0 else 0
        else:
            held = len([n for n in range(nfr) if n % world == rank])
        outs = [torch.empty((held, Hs * SCALE, Ws * SCALE, 3), dtype=torch.uint8, device="cuda") if held else None for _ in range(2)]
        pend = [None, None]
        k = [0]

        def step():
            b = k[0] & 1
            if pend[b] is not None:
                pend[b].wait()                       # the exchange issued two steps ago has landed: its buffer is free
            pend[b] = sr_strips(lr, compute, SCALE, eng.halo, dst=0 if mode == "root" else "rotate", out=outs[b], device="cuda", via_host=via_host, wait=False)
            k[0] += 1

        def drain():
            for b in (0, 1):
                if pend[b] is not None:
                    pend[b].wait()
                    pend[b] = None
        step(); drain()
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = _t.perf_counter()
        for _ in range(steps):
            step()
        drain()
        torch.cuda.synchronize()
        if dist_on:
            torch.distributed.barrier()
        el = _t.perf_counter() - t0
        t = torch.tensor([el], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        el = float(t.item())
        frame_bytes = Hs * SCALE * Ws * SCALE * 3
        del outs
        torch.cuda.empty_cache()
        return {"frames_per_step": nfr, "value": round(nfr * Hs * SCALE * Ws * SCALE * steps / el / 1e6, 2), "unit": "Mpix/s",
                "ms_per_step": round(el / steps * 1e3, 3), "exchanged_bytes_per_step": nfr * frame_bytes * (world - 1) // world,
                "max_bytes_into_one_rank_per_step": (nfr if mode == "root" else -(-nfr // world)) * frame_bytes * (world - 1) // world}
    rot_frames = world * max(1, Fs // world)
    rec = {"workload": "config 3: LR %dx%dx3 frames, each cut into %d strips (+%d-row halo) computed on %d GPUs; the HR strips travel point to point "
                       "(%s) straight into their rows, the exchange of step k overlapping the compute of step k+1"
                       % (Hs, Ws, world, eng.halo, world, "RCCL" if backend == "nccl" else backend + " via host memory"),
           "n_gpus": world, "scaling": "strong", "ranks_in_group": torch.distributed.get_world_size()}
    rec["single_root"] = leg("root", Fs)
    rec["rotate"] = leg("rotate", rot_frames)
    rec["value"], rec["unit"], rec["ms_per_step"] = rec["rotate"]["value"], "Mpix/s", rec["rotate"]["ms_per_step"]      # the mode that scales
    return rec


def config4(args, world, rank, local, timed, steps=None):
    """BASELINE config 4: one LUT fine-tune step (forward + backward HIP kernels + Adam) at bs 256 of 1x48x48 crops, 2-stage sdy x4."""
    import tempfile
    import torch
    from mulut_amd.finetune import MuLUT
    from mulut_amd.synth import natural_frames
    bs, crop = 256, 48
    with tempfile.TemporaryDirectory() as td:
        for s_ in (1, 2):
            for m in MODES:
                np.save(os.path.join(td, "LUT_x4_4bit_int8_s%d_%s.npy" % (s_, m)), np.load(os.path.join(LUT_DIR, "LUT_ft_x4_4bit_int8_s%d_%s.npy" % (s_, m))))
        net = MuLUT(td, 2, MODES, upscale=4, interval=4).cuda()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, fused=True)       # one launch for the six tables
    big = natural_frames(1, 1080, 1920, 1, rank)[0, :, :, 0]
    rng = np.random.default_rng(rank)
    ys, xs = rng.integers(0, 1080 - crop, bs), rng.integers(0, 1920 - crop, bs)
    x = torch.from_numpy(np.stack([big[a:a + crop, b:b + crop] for a, b in zip(ys, xs)])[:, None].astype(np.float32) / 255.0).cuda()
    y = torch.rand((bs, 1, crop * 4, crop * 4), device="cuda", generator=torch.Generator(device="cuda").manual_seed(rank))

    def step():
        opt.zero_grad()
        torch.nn.functional.mse_loss(net(x), y).backward()
        opt.step()
    for _ in range(max(args.warmup, 3)):
        step()
    steps = steps or args.steps
    el = timed(step, steps)
    ms = el / steps * 1e3
    # the backward kernel of the final stage dominates the step; two candidate bounds for it, both from MI355X_MICROARCH.md:
    # The table-gradient adds are LDS read + add + write of an address private to a 16-lane group (7.0 cycles of the CU's LDS
    # pipeline per 64-lane wave instruction, measured: profiles/r03_ubench_lds_atomic.txt; a ds_add_f32 takes 48 whatever its
    # addresses, which is why only evictions use it); rows outside the band and the flush use memory-side float atomics
    # (about 1.3 TB/s of added bytes chip-wide)
    sites = bs * crop * crop
    adds = sites * 12 * 5 * 16                     # (site, pass, row, element) float adds into the final-stage tables
    counters = load_profile_json("finetune_counters.json")
    return {
        "metric": "LR Mpixels/sec through one LUT fine-tune step (forward + backward + Adam), 2-stage sdy x4", "value": round(world * bs * crop * crop * steps / el / 1e6, 3),
        "unit": "Mpix/s", "n_gpus": world, "steps": steps, "warmup": max(args.warmup, 3), "ms_per_step": round(ms, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "config 4: bs 256 x 1x48x48 crops of the D-natural field per GPU per step, shipped tables as the start point",
                   "note": "replicas only: each rank trains its own copy (a data-parallel all-reduce of the six table gradients is not built)",
                   "reference_logged": "7.0 s/iter at batch 320 (models/sr_x2sdy/lutft.log), unspecified 2022 GPU"},
        "roofline": {"bound": "lds_rmw", "kernel": "ft_stage_bwd4 (final-stage backward: table gradients summed in per-group LDS caches of band rows -- a site's five vertices in flight together --, evicted into an LDS copy of the tube band, flushed once per workgroup and mode; clamp mask from the forward; input gradient in per-group LDS tiles)",
                     "table_gradient_adds_per_step": adds,
                     "achieved": round(adds / (ms * 1e-3) / 1e9, 2), "unit": "G float adds/s (whole step time as the denominator)",
                     "peak": round(256 * 64 / 7.0 * CLOCK_GHZ, 1), "peak_how": "256 CUs x 64 lanes per 7.0 LDS cycles (read + add + write of private addresses, measured: profiles/r03_ubench_lds_atomic.txt) x 2.4 GHz; as ds_add_f32 (48 cycles per wave instruction) the peak would be 819",
                     "frac": round(adds / (ms * 1e-3) / 1e9 / (256 * 64 / 7.0 * CLOCK_GHZ), 4),
                     "memory_side_float_atomics": {"peak": 1300.0, "unit": "GB/s of added bytes (MI355X_MICROARCH.md, Global float atomics)",
                                                   "if_every_add_went_to_memory": round(adds * 4 / (ms * 1e-3) / 1e9, 1)},
                     "counters": counters.get("kernels") if counters else None, "traffic": None},
    }


def config5(args, world, rank, local, timed, size=None, steps=None):
    """BASELINE config 5: 4-stage sdy x2 cascade on seeded synthetic tables, eager launches vs one hipGraph replay."""
    import torch
    from mulut_amd import MuLUTEngine
    stages, scale = 4, 2
    rng = np.random.default_rng(5)
    F, H, W = size or (args.frames, args.lr_h, args.lr_w)
    steps = steps or args.steps
    eng = MuLUTEngine(local).configure(stages, MODES, scale, 4)
    for s in range(1, stages + 1):
        for m in MODES:
            vn = scale * scale if s == stages else 1
            base = rng.integers(-20, 21, (17, 17, 17, 17, vn)).astype(np.float32)
            grid = np.indices((17, 17, 17, 17)).astype(np.float32).sum(0)[..., None] * (3.0 if s < stages else 4.0) - 96.0
            eng.set_lut(s, m, np.clip(np.rint(grid + base), -127, 127).astype(np.int8).reshape(-1, vn))
    eng.reserve(F, H, W, 3)
    x = torch.from_numpy(make_batch(args.dist, F, H, W, seed=rank)).cuda()
    out = torch.empty((F, H * scale, W * scale, 3), dtype=torch.uint8, device="cuda")
    for _ in range(args.warmup):
        eng.pipeline(x, out=out)
    el_eager = timed(lambda: eng.pipeline(x, out=out), steps)
    eng.set_stage_timing(True)
    eng.pipeline(x, out=out)
    ms_stage = eng.last_stage_ms()
    eng.set_stage_timing(False)
    want = out.clone()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        eng.pipeline(x, out=out)
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eng.pipeline(x, out=out)
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    same = bool(torch.equal(out, want))
    el_graph = timed(g.replay, steps)
    sites = F * H * W * 3
    alg = sites * (1 + scale * scale) + 3 * 83521 * ((stages - 1) + scale * scale)
    ms = el_graph / steps * 1e3
    return {
        "metric": "Mpixels/sec SR-x2 4-stage sdy LUT inference, hipGraph replay (HR output pixels)", "value": round(world * F * H * scale * W * scale * steps / el_graph / 1e6, 2),
        "unit": "Mpix/s", "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "config 5: 4-stage sdy x2, %d x LR %dx%dx3 -> HR %dx%dx3 per GPU per step, D-%s, seeded synthetic tables"
                               % (F, H, W, H * scale, W * scale, args.dist),
                   "eager_ms_per_step": round(el_eager / steps * 1e3, 4), "graph_ms_per_step": round(ms, 4),
                   "graph_replay_matches_eager": same, "stage_ms": [round(v, 4) for v in ms_stage],
                   "kernels": [eng.kernel_name(False), eng.kernel_name(True)]},
        "roofline": {"bound": "hbm", "kernel": "4-stage pipeline", "achieved": round(alg / (ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": None, "algorithmic_bytes_per_step": alg},
    }


def other_configs(args, world, rank, local, timed):
    """Configs 4 and 5 in brief, for the default line's config.other_configs (the driver only runs the default line)."""
    out = {}
    for name, fn in (("config5_p1", lambda: config5(args, world, rank, local, timed, size=(8, 1080, 1920), steps=10)),
                     ("config5_small_frames", lambda: config5(args, world, rank, local, timed, size=(8, 270, 480), steps=50)),
                     ("config4", lambda: config4(args, world, rank, local, timed, steps=10))):
        try:
            r = fn()
            keep = {"workload": r["config"]["workload"], "value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "roofline_frac": r["roofline"]["frac"],
                    "roofline_bound": r["roofline"]["bound"]}
            for k in ("eager_ms_per_step", "graph_ms_per_step", "graph_replay_matches_eager", "stage_ms"):
                if k in r["config"]:
                    keep[k] = r["config"][k]
            out[name] = keep
        except Exception as exc:      # the headline must not depend on these legs
            out[name] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
    return out


if __name__ == "__main__":
    main()
