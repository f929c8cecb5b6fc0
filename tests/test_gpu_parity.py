"""GPU parity tests (-m gpu): the HIP path, called through the C ABI (mulut_amd.engine -> libmulut_hip.so),
against (a) fixtures produced by running the reference itself, (b) the reference's own Set5 PNGs,
(c) the CPU oracle on seeded inputs, (d) size-independent properties at the full BASELINE size.
Bar: bit-exact (integer / byte work)."""
import os
import re

import numpy as np
import pytest
from PIL import Image

from conftest import GOLDEN

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from mulut_amd import MuLUTEngine, MuLUTError, load_lut_dict, synthetic_lut  # noqa: E402
from mulut_amd.engine import LAYOUT_CHW, LAYOUT_HWC  # noqa: E402
from oracle import c_oracle  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def eng(shipped_luts):
    e = MuLUTEngine(0)
    e.configure(2, "sdy", 4, 4).set_lut_dict(shipped_luts)
    yield e
    e.close()


def dev(a):
    return torch.from_numpy(np.array(a, order="C", copy=True)).cuda()


def natural_image(h, w, c=3, seed=0):
    """D-natural of SURVEY 8d: low-frequency sinusoids + sigma=2 noise."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    out = np.zeros((h, w, c))
    for ch in range(c):
        acc = np.zeros((h, w))
        for _ in range(6):
            fy, fx = rng.uniform(0.5, 6.0, 2) * 2 * np.pi / max(h, w)
            acc += rng.uniform(0.3, 1.0) * np.sin(fy * yy + fx * xx + rng.uniform(0, 2 * np.pi))
        acc = (acc - acc.min()) / (acc.max() - acc.min()) * 255.0
        out[:, :, ch] = acc + rng.normal(0, 2, (h, w))
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


# ---------------------------------------------------------------------------------------------
# (a) fixtures from the reference
# ---------------------------------------------------------------------------------------------
def test_pass_matches_reference_fixtures(eng, pass_fx):
    n = 0
    for key in sorted(k for k in pass_fx.files if k.startswith("out/")):
        _, name, u, mode, r = key.split("/")
        u, r = int(u[1:]), int(r[1:])
        img = pass_fx["in/" + name].transpose(2, 0, 1)
        got = eng.pass_q(2 if u == 4 else 1, mode, r, dev(img)).cpu().numpy()
        assert np.array_equal(got, pass_fx[key]), key
        n += 1
    assert n == 168


def test_interp_twin_has_reference_signature_and_values(pass_fx, shipped_luts):
    from mulut_amd import FourSimplexInterpFaster
    for name in ("rand_19x13x3", "ties_9x11x3", "one_1x1x1"):
        img = pass_fx["in/" + name].astype(np.float32)
        for u, st in ((1, 1), (4, 2)):
            for mode in "sdy":
                pad = 1 if mode == "s" else 2
                for r in range(4):
                    rimg = np.rot90(img, r)
                    h, w, _ = rimg.shape
                    img_in = np.pad(rimg, ((0, pad), (0, pad), (0, 0)), mode="edge").transpose(2, 0, 1)
                    weight = shipped_luts["s%d_%s" % (st, mode)].astype(np.float32)
                    out = FourSimplexInterpFaster(weight, img_in, h, w, 4, 4 - r, upscale=u, mode=mode)
                    assert out.dtype == np.float64
                    assert np.array_equal(out * 16, pass_fx["out/%s/u%d/%s/r%d" % (name, u, mode, r)])
    with pytest.raises(ValueError, match="Mode x not implemented."):
        FourSimplexInterpFaster(np.zeros((17 ** 4, 1), np.float32), np.zeros((1, 4, 4), np.float32), 3, 3, 4, 0, 1, "x")


def test_two_stage_sdy_matches_reference(eng, pipe_fx):
    for name in sorted({k.split("/")[1] for k in pipe_fx.files if k.startswith("s2sdy/")}):
        img = pipe_fx["in/" + name]
        st1 = eng.stage(1, dev(img)).cpu().numpy()
        assert np.array_equal(st1, pipe_fx["s2sdy/%s/stage1" % name]), name
        fin = eng.pipeline(dev(img)).cpu().numpy()
        assert np.array_equal(fin, pipe_fx["s2sdy/%s/final" % name]), name
        # planar layout gives the same bytes
        chw = eng.pipeline(dev(img.transpose(2, 0, 1)), layout=LAYOUT_CHW).cpu().numpy()
        assert np.array_equal(chw.transpose(1, 2, 0), fin), name


def test_other_configs_match_reference(pipe_fx):
    e = MuLUTEngine(0)
    keys = sorted({"/".join(k.split("/")[:2]) for k in pipe_fx.files if k.startswith("synth_")})
    assert len(keys) >= 20
    for key in keys:
        cfg, name = key.split("/")
        m = re.match(r"synth_S(\d)_([sdy]+)_x(\d)", cfg)
        stages, modes, scale = int(m.group(1)), m.group(2), int(m.group(3))
        e.configure(stages, modes, scale, 4)
        for s in range(stages):
            for mode in modes:
                seed = 1000 * stages + 100 * scale + 10 * s + "sdy".index(mode)
                e.set_lut(s + 1, mode, synthetic_lut(seed, scale * scale if s + 1 == stages else 1))
        img = pipe_fx["in/" + name]
        fin = e.pipeline(dev(img)).cpu().numpy()
        assert np.array_equal(fin, pipe_fx["%s/stage%d" % (key, stages)]), key
        cur = dev(img)
        for s in range(stages - 1):
            cur = e.stage(s + 1, cur)
            assert np.array_equal(cur.cpu().numpy(), pipe_fx["%s/stage%d" % (key, s + 1)]), (key, s)
    e.close()


def test_config1_single_stage_s(pipe_fx, shipped_luts):
    e = MuLUTEngine(0).configure(1, "s", 4, 4)
    e.set_lut(1, "s", shipped_luts["s2_s"])
    for name in ("rand_19x13x3", "smooth_24x20x3"):
        got = e.pipeline(dev(pipe_fx["in/" + name])).cpu().numpy()
        assert np.array_equal(got, pipe_fx["cfg1_s2s_as_s1/%s/final" % name])
    e.close()


# ---------------------------------------------------------------------------------------------
# (b) the reference's own golden outputs: Set5, through the CLI twin
# ---------------------------------------------------------------------------------------------
def test_cli_reproduces_set5_pngs_and_psnr(tmp_path, capsys):
    from mulut_amd import test_lut
    test_dir = tmp_path / "SRBenchmark"
    (test_dir / "Set5").mkdir(parents=True)
    os.symlink(os.path.join(GOLDEN, "Set5", "HR"), test_dir / "Set5" / "HR")
    os.symlink(os.path.join(GOLDEN, "Set5", "LR_bicubic"), test_dir / "Set5" / "LR_bicubic")
    exp = tmp_path / "models" / "sr_x2sdy"
    exp.mkdir(parents=True)
    for fn in os.listdir(os.path.join(GOLDEN, "luts")):
        os.symlink(os.path.join(GOLDEN, "luts", fn), exp / fn)
    res = test_lut.main(["--stages", "2", "--modes", "sdy", "-e", str(exp), "--testDir", str(test_dir),
                         "--resultRoot", str(tmp_path / "results")])
    line = capsys.readouterr().out.strip().splitlines()[-1]
    assert line == "Dataset Set5 | AVG LUT PSNR: 30.61 SSIM: 0.8656" or line.startswith(
        "Dataset Set5 | AVG LUT PSNR: 30.61 SSIM: 0.865")       # reference prints 30.61 / 0.8655 (:343)
    assert res["Set5"].shape == (5, 2)
    out_dir = tmp_path / "results" / "sr_x2sdy" / "Set5" / "X4"
    for fn in sorted(os.listdir(os.path.join(GOLDEN, "Set5", "ref_out"))):
        want = np.array(Image.open(os.path.join(GOLDEN, "Set5", "ref_out", fn)))
        got = np.array(Image.open(out_dir / fn))              # same file naming as the reference (:311)
        assert np.array_equal(got, want), fn


def test_gray_input_is_replicated_like_the_reference(eng, pipe_fx):
    from mulut_amd.test_lut import eltr
    gray = pipe_fx["in/rand_8x31x1"][:, :, 0]
    obj = eltr.__new__(eltr)
    obj.engine = eng
    out = obj.super_resolve(gray)
    want = np.repeat(pipe_fx["s2sdy/rand_8x31x1/final"], 3, axis=2)
    assert np.array_equal(out, want)


# ---------------------------------------------------------------------------------------------
# (c) seeded inputs vs the CPU oracle, edge shapes, batches, layouts
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(1, 1, 3), (1, 7, 1), (9, 1, 3), (2, 2, 3), (33, 65, 3), (64, 32, 3), (65, 129, 3),
                                   (31, 200, 1), (100, 37, 2)])
def test_shapes_vs_oracle(eng, shipped_luts, shape):
    img = np.random.default_rng(sum(shape)).integers(0, 256, shape, dtype=np.uint8)
    got = eng.pipeline(dev(img)).cpu().numpy()
    assert np.array_equal(got, c_oracle.pipeline(shipped_luts, 2, "sdy", 4, img))


@pytest.mark.parametrize("C", [4, 5, 7])
def test_more_than_three_channels(eng, shipped_luts, C):
    """The reference function is channel-count agnostic (sr/4_test_lut.py:14-237): the boundary runs groups of three planes."""
    img = np.concatenate([natural_image(45, 70, 3, seed=C), np.random.default_rng(C).integers(0, 256, (45, 70, C - 3), dtype=np.uint8)], axis=2)
    want = c_oracle.pipeline(shipped_luts, 2, "sdy", 4, img)
    assert want.shape == (180, 280, C)
    assert np.array_equal(eng.pipeline(dev(img)).cpu().numpy(), want)
    got_p = eng.pipeline(dev(img.transpose(2, 0, 1)), layout=LAYOUT_CHW).cpu().numpy()
    assert np.array_equal(got_p.transpose(1, 2, 0), want)


def test_batch_and_natural_vs_oracle(eng, shipped_luts):
    imgs = np.stack([natural_image(70, 150, 3, seed=s) for s in range(3)] +
                    [np.random.default_rng(9).integers(0, 256, (70, 150, 3), dtype=np.uint8)])
    got = eng.pipeline(dev(imgs)).cpu().numpy()
    assert got.shape == (4, 280, 600, 3)
    for k in range(4):
        assert np.array_equal(got[k], c_oracle.pipeline(shipped_luts, 2, "sdy", 4, imgs[k])), k
    # planar batch
    got_p = eng.pipeline(dev(imgs.transpose(0, 3, 1, 2)), layout=LAYOUT_CHW).cpu().numpy()
    assert np.array_equal(got_p.transpose(0, 2, 3, 1), got)


@pytest.mark.parametrize("scale,stages,modes", [(2, 4, "sdy"), (3, 2, "sd"), (4, 1, "y"), (1, 2, "sdy"), (2, 1, "ss"),
                                                (4, 3, "ydsd")])
def test_generic_configs_vs_oracle(scale, stages, modes):
    e = MuLUTEngine(0).configure(stages, modes, scale, 4)
    luts = {}
    for s in range(stages):
        for mode in set(modes):
            luts["s%d_%s" % (s + 1, mode)] = synthetic_lut(7 * s + ord(mode), scale * scale if s + 1 == stages else 1)
    e.set_lut_dict(luts)
    for C in (1, 2, 3):
        img = np.random.default_rng(C).integers(0, 256, (37, 53, C), dtype=np.uint8)
        want = c_oracle.pipeline(luts, stages, modes, scale, img)
        assert np.array_equal(e.pipeline(dev(img)).cpu().numpy(), want), (scale, stages, modes, C)
        got_p = e.pipeline(dev(img.transpose(2, 0, 1)), layout=LAYOUT_CHW).cpu().numpy()
        assert np.array_equal(got_p.transpose(1, 2, 0), want)
    e.close()


def test_extreme_tables_vs_oracle():
    img = np.random.default_rng(5).integers(0, 256, (40, 40, 3), dtype=np.uint8)
    e = MuLUTEngine(0).configure(2, "sdy", 4, 4)
    for val in (127, -128):
        luts = {"s%d_%s" % (s, m): np.full((17 ** 4, 16 if s == 2 else 1), val, np.int8) for s in (1, 2) for m in "sdy"}
        e.set_lut_dict(luts)
        assert np.array_equal(e.pipeline(dev(img)).cpu().numpy(), c_oracle.pipeline(luts, 2, "sdy", 4, img))
    e.close()


# ---------------------------------------------------------------------------------------------
# error behaviour of the boundary
# ---------------------------------------------------------------------------------------------
def test_error_behaviour(shipped_luts):
    e = MuLUTEngine(0)
    with pytest.raises(MuLUTError):                       # not configured
        e.pipeline(torch.zeros((4, 4, 3), dtype=torch.uint8, device="cuda"))
    with pytest.raises(ValueError, match="Mode"):         # reference: ValueError("Mode {} not implemented.")
        e.configure(2, "sxq", 4, 4)
    with pytest.raises(MuLUTError):                       # interval != 4
        e.configure(2, "sdy", 4, 3)
    e.configure(2, "sdy", 4, 4)
    with pytest.raises(MuLUTError, match="not set"):      # reference: FileNotFoundError at np.load
        e.pipeline(torch.zeros((4, 4, 3), dtype=torch.uint8, device="cuda"))
    e.set_lut_dict(shipped_luts)
    e.set_lut(2, "s", shipped_luts["s1_s"])               # wrong v_num for the last stage
    with pytest.raises(MuLUTError, match="shape"):
        e.pipeline(torch.zeros((4, 4, 3), dtype=torch.uint8, device="cuda"))
    with pytest.raises(MuLUTError):
        e.set_lut(1, "s", np.zeros((100, 1), np.int8))
    with pytest.raises(TypeError):
        e.pipeline(torch.zeros((4, 4, 3), dtype=torch.float32, device="cuda"))
    with pytest.raises(TypeError):
        e.pipeline(torch.zeros((4, 4, 3), dtype=torch.uint8))   # host tensor: no CPU path
    e.close()


# ---------------------------------------------------------------------------------------------
# strips (tile sharding): seams are bit-exact
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nstrips", [2, 3, 8])
def test_strips_tile_bit_exactly(eng, nstrips):
    H, W = 97, 61
    img = natural_image(H, W, 3, seed=4)
    full = eng.pipeline(dev(img))
    halo = eng.halo
    assert halo == 4
    bounds = np.linspace(0, H, nstrips + 1).astype(int)
    parts = []
    for k in range(nstrips):
        y0, y1 = int(bounds[k]), int(bounds[k + 1])
        r0, r1 = max(0, y0 - halo), min(H, y1 + halo)
        parts.append(eng.pipeline_rows(dev(img[r0:r1]), r0, y0, y1, H))
    assert torch.equal(torch.cat(parts, 0), full)
    with pytest.raises(MuLUTError, match="halo"):          # band too small
        eng.pipeline_rows(dev(img[10:20]), 10, 10, 20, H)


# ---------------------------------------------------------------------------------------------
# (d) BASELINE size (1080x1920x3 -> 4320x7680x3): whole frame vs oracle + size-independent properties
# ---------------------------------------------------------------------------------------------
def test_full_1080p_frame(eng, shipped_luts):
    H, W = 1080, 1920
    nat = natural_image(H, W, 3, seed=0)
    noise = np.random.default_rng(0).integers(0, 256, (H, W, 3), dtype=np.uint8)
    batch = dev(np.stack([nat, noise]))
    out = eng.pipeline(batch)
    assert out.shape == (2, 4320, 7680, 3)
    # whole natural frame against the oracle (about 13 s of CPU)
    assert np.array_equal(out[0].cpu().numpy(), c_oracle.pipeline(shipped_luts, 2, "sdy", 4, nat))
    # noise frame: three 96x96 windows recomputed by the oracle with a 4-px halo (receptive field)
    o1 = out[1].cpu().numpy()
    for (y, x) in ((0, 0), (500, 900), (1080 - 96, 1920 - 96)):
        y0, y1, x0, x1 = max(0, y - 4), min(H, y + 100), max(0, x - 4), min(W, x + 100)
        ref = c_oracle.pipeline(shipped_luts, 2, "sdy", 4, noise[y0:y1, x0:x1])
        ref = ref[(y - y0) * 4:(y - y0 + 96) * 4, (x - x0) * 4:(x - x0 + 96) * 4]
        assert np.array_equal(o1[y * 4:(y + 96) * 4, x * 4:(x + 96) * 4], ref), (y, x)
    # rotation covariance: the 4-rotation ensemble commutes with rot90 of the image
    rot = eng.pipeline(dev(np.ascontiguousarray(np.rot90(noise, 1))))
    assert torch.equal(rot, torch.rot90(out[1], 1, (0, 1)))
    # strips == whole frame (8 strips as on 8 GPUs)
    parts = []
    for k in range(8):
        y0, y1 = k * 135, (k + 1) * 135
        r0, r1 = max(0, y0 - 4), min(H, y1 + 4)
        parts.append(eng.pipeline_rows(dev(noise[r0:r1]), r0, y0, y1, H))
    assert torch.equal(torch.cat(parts, 0), out[1])
    # determinism
    assert torch.equal(eng.pipeline(batch), out)


# ---------------------------------------------------------------------------------------------
# hipGraph capture (deep cascades are launch-bound at small frames; config 5 of BASELINE.json)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("stages,modes,scale,w", [(4, "sdy", 2, 77), (2, "sdy", 3, 76), (1, "yd", 3, 76)])
def test_pipeline_replays_from_a_captured_graph(stages, modes, scale, w):
    """(width 76: a multiple of four, i.e. the routed launch of the x2 / x3 final stages -- tile marks, memsets and the gather kernel's
    launch on marked tiles are all inside the captured region)"""
    e = MuLUTEngine(0).configure(stages, modes, scale, 4)
    luts = {}
    for s in range(stages):
        for mode in modes:
            luts["s%d_%s" % (s + 1, mode)] = synthetic_lut(31 * s + ord(mode), scale * scale if s + 1 == stages else 1)
    e.set_lut_dict(luts)
    img = np.random.default_rng(3).integers(0, 256, (2, 45, w, 3), dtype=np.uint8)
    img[0, :, : w // 2] = natural_image(45, w // 2, 3, seed=4)         # smooth and noisy tiles in one frame
    x = dev(img)
    out = torch.empty((2, 45 * scale, w * scale, 3), dtype=torch.uint8, device="cuda")
    e.reserve(2, 45, w, 3)                        # no allocation inside the captured region
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        e.pipeline(x, out=out)                    # warm-up: kernel attributes are set on first launch
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        e.pipeline(x, out=out)
    want = np.stack([c_oracle.pipeline(luts, stages, modes, scale, im) for im in img])
    for trial in range(2):
        x.copy_(dev(np.roll(img, trial, axis=2)))
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        ref = want if trial == 0 else np.stack([c_oracle.pipeline(luts, stages, modes, scale, im)
                                                for im in np.roll(img, trial, axis=2)])
        assert np.array_equal(out.cpu().numpy(), ref), trial
    e.close()


def test_hybrid_final_stage_variants_agree_and_capture(eng, shipped_luts):
    """Every final-stage variant (full-table gathers, both tube kernels on every tile, hybrid with several thresholds) gives
    the same bytes on smooth, photographic and noisy content; the default (hybrid) path replays from a hipGraph."""
    from mulut_amd.synth import natural_frames, noise_frames, real_frames
    png = os.path.join(GOLDEN, "DIV2K_LR_X4", "0001x4.png")
    frames = np.concatenate([natural_frames(1, 150, 200, 3, 1), noise_frames(1, 150, 200, 3, 1),
                             real_frames(1, 150, 200, png, 1)])
    x = dev(frames)
    eng.set_tuning("final_stage_kernel", 1)
    want = eng.pipeline(x).clone()
    assert np.array_equal(want[2].cpu().numpy(), c_oracle.pipeline(shipped_luts, 2, "sdy", 4, frames[2]))
    for pipelined in (1, 0):          # stage_tube2_kernel (hand-scheduled, the default for sdy) and stage_tube_kernel (any mode list)
        eng.set_tuning("tube_pipelined", pipelined)
        for sel, thr in ((5, None), (6, 0), (6, 128), (6, 1024), (0, None)):
            eng.set_tuning("final_stage_kernel", sel)
            if thr is not None:
                eng.set_tuning("hybrid_oob_per_1024", thr)
            assert torch.equal(eng.pipeline(x), want), (pipelined, sel, thr)
    eng.set_tuning("tube_pipelined", 1).set_tuning("hybrid_oob_per_1024", 128).set_tuning("final_stage_kernel", 0)
    for key in ("stat_from_first_stage", "detail_kernel", "fix_kernel"):       # the routing / work-list options of the default path
        for val in ((0, 1, 2) if key == "fix_kernel" else (0, 1)):
            eng.set_tuning(key, val)
            assert torch.equal(eng.pipeline(x), want), (key, val)
    eng.set_tuning("stat_from_first_stage", 1).set_tuning("detail_kernel", 0).set_tuning("fix_kernel", 0)
    for first in (2, 3, 0):                 # first stage: window kernel everywhere, tube kernel everywhere, the routed default
        eng.set_tuning("first_stage_kernel", first)
        assert torch.equal(eng.pipeline(x), want), first
    for retired in (("final_stage_kernel", 3), ("first_stage_kernel", 1), ("tube_site_flags", 1)):      # generations moved out in round 3
        with pytest.raises(Exception):
            eng.set_tuning(*retired)
    # capture the default path
    out = torch.empty_like(want)
    eng.reserve(3, 150, 200, 3)
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
    assert torch.equal(out, want)


def test_randomised_configurations_vs_oracle():
    """60 seeded random configurations (stages, mode strings, scales, channels, layouts, strips, kernel variants,
    content and table kinds) through tools/fuzz_parity.py; the long runs are logged in profiles/."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "--cases", "60", "--seed", "123"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.parametrize("shape", [(3, 37, 129, 3), (1, 150, 200, 3), (2, 16, 64, 1), (1, 33, 70, 2), (1, 5, 9, 3)])
def test_detailed_tiles_take_the_anchor_slab_path(shape):
    """Noisy content goes through the anchor-slab kernels of the final stage (the work counters say so), gives the oracle's
    bytes for random and extreme tables in both layouts and in strips, the same bytes as the full-table gather kernel, and the
    same bytes again when repeated (the id lists are built without atomics)."""
    rng = np.random.default_rng(shape[1] * 1000 + shape[2])
    n, h, w, c = shape
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    img[:, :, : w // 3] = (img[:, :, : w // 3] // 64) + 100            # a smooth part too: both final-stage kernels run
    for val in (None, 127, -128):
        luts = {}
        for st in (1, 2):
            for m in "sdy":
                vn = 16 if st == 2 else 1
                luts["s%d_%s" % (st, m)] = (np.full((17 ** 4, vn), val, np.int8) if val is not None
                                             else rng.integers(-128, 128, (17 ** 4, vn), dtype=np.int8))
        want = np.stack([c_oracle.pipeline(luts, 2, "sdy", 4, im) for im in img])
        e = MuLUTEngine(0).configure(2, "sdy", 4, 4).set_lut_dict(luts)
        got = e.pipeline(dev(img))
        cnt = e.last_detail_counters()
        assert np.array_equal(got.cpu().numpy(), want), val
        if val is None and w >= 64:
            assert sum(cnt["samples_per_anchor"]) > 0 and cnt["items"] > 0, cnt        # the slab path did run
        assert torch.equal(e.pipeline(dev(img)), got)                                    # deterministic
        chw = e.pipeline(dev(np.ascontiguousarray(img.transpose(0, 3, 1, 2))), layout=LAYOUT_CHW)
        assert np.array_equal(chw.cpu().numpy().transpose(0, 2, 3, 1), want), val
        if h >= 8:                                                                         # two strips of the first image
            mid, halo = h // 2, e.halo
            top = e.pipeline_rows(dev(img[0][: min(h, mid + halo)]), 0, 0, mid, h, layout=LAYOUT_HWC)
            bot = e.pipeline_rows(dev(img[0][max(0, mid - halo):]), max(0, mid - halo), mid, h, h, layout=LAYOUT_HWC)
            assert np.array_equal(torch.cat([top, bot], 0).cpu().numpy(), want[0]), val
        e.set_tuning("detail_kernel", 1)
        assert torch.equal(e.pipeline(dev(img)), got)
        e.close()


@pytest.mark.parametrize("modes", ["sdysd", "sdysdysd"])
def test_more_than_four_modes_at_scale_4(modes):
    """scale 4 with 5 and 8 modes: per-rotation accumulators (the merged pairs hold 4 x 16 x 255 per field at most),
    extreme and random tables, vs the oracle"""
    from mulut_amd import MuLUTEngine
    rng = np.random.default_rng(len(modes))
    img = rng.integers(0, 256, (2, 21, 34, 3), dtype=np.uint8)
    for val in (127, -128, None):
        luts = {"s1_%s" % m: (np.full((17 ** 4, 16), val, np.int8) if val is not None else rng.integers(-128, 128, (17 ** 4, 16), dtype=np.int8))
                for m in set(modes)}
        e = MuLUTEngine(0).configure(1, modes, 4, 4).set_lut_dict(luts)
        got = e.pipeline(dev(img)).cpu().numpy()
        want = np.stack([c_oracle.pipeline(luts, 1, modes, 4, im) for im in img])
        assert np.array_equal(got, want), (modes, val)
        e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("modes", ["dys", "sdys", "yssd", "ddys", "sdyy"])
def test_mode_lists_as_pattern_multisets(modes):
    """A mode list is a multiset of patterns: stage_tube2_kernel (built for s, d, y) takes any list that uses all three, in any
    order and with repeats up to four modes, by staging k-fold bands.  Smooth content (so that the tube kernel does the work),
    extreme tables (field capacity with a doubled pattern) and random ones, every tile on the tube kernel and the hybrid."""
    from mulut_amd import MuLUTEngine
    rng = np.random.default_rng(len(modes) + ord(modes[0]))
    img = np.stack([natural_image(40, 72, 3, seed=s) for s in (1, 2)])
    for val in (127, -128, None):
        luts = {"s1_%s" % m: (np.full((17 ** 4, 16), val, np.int8) if val is not None else rng.integers(-128, 128, (17 ** 4, 16), dtype=np.int8))
                for m in set(modes)}
        e = MuLUTEngine(0).configure(1, modes, 4, 4).set_lut_dict(luts)
        want = np.stack([c_oracle.pipeline(luts, 1, modes, 4, im) for im in img])
        for sel in (5, 0, 1):
            e.set_tuning("final_stage_kernel", sel)
            assert np.array_equal(e.pipeline(dev(img)).cpu().numpy(), want), (modes, val, sel)
            if sel == 5:
                assert "tube2" in e.kernel_name(1), e.kernel_name(1)
        e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("stages,scale,modes", [(2, 4, "sdys"), (2, 2, "ydsd"), (3, 1, "sdysdysd"), (2, 4, "sdysd"), (2, 2, "ssdyy")])
def test_long_mode_lists_in_cascades(stages, scale, modes):
    """More than three modes through the LDS kernels of the non-final stages (bands per pattern; any list), the x2 final stage
    (up to four modes) and the x4 final stage (multisets up to four modes, per-rotation accumulators beyond), on smooth content."""
    from mulut_amd import MuLUTEngine
    img = np.stack([natural_image(52, 80, 3, seed=s) for s in (3, 4)])
    luts = {}
    for s in range(stages):
        for m in set(modes):
            luts["s%d_%s" % (s + 1, m)] = synthetic_lut(11 * s + ord(m), scale * scale if s + 1 == stages else 1)
    e = MuLUTEngine(0).configure(stages, modes, scale, 4).set_lut_dict(luts)
    want = np.stack([c_oracle.pipeline(luts, stages, modes, scale, im) for im in img])
    for first in (0, 3, 2):
        e.set_tuning("first_stage_kernel", first)
        assert np.array_equal(e.pipeline(dev(img)).cpu().numpy(), want), (stages, scale, modes, first)
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("modes", ["sdysd", "sdysdysd", "yyddsss"])
def test_five_to_eight_modes_on_the_lds_path(modes):
    """x4 final stages with five to eight modes run on the pipelined tube kernel (a mode list is a multiset of the patterns s, d, y; the pair sums
    still fit their unsigned 16-bit fields, block values are added in 32 bits; seven modes take the integer epilogue), their detailed tiles on the per-rotation gather
    kernel, their flagged samples on the fix-up kernel: extreme tables (every field at its bound) and random ones, smooth + noisy content,
    every final-stage variant, vs the oracle."""
    from mulut_amd import MuLUTEngine
    rng = np.random.default_rng(len(modes))
    img = np.stack([np.concatenate([natural_image(40, 70, 3, seed=s), rng.integers(0, 256, (40, 66, 3), dtype=np.uint8)], axis=1) for s in (1, 2)])
    for val in (127, -128, None):
        luts = {}
        for m in set(modes):
            luts["s1_%s" % m] = synthetic_lut(ord(m), 1)
            luts["s2_%s" % m] = np.full((17 ** 4, 16), val, np.int8) if val is not None else rng.integers(-128, 128, (17 ** 4, 16), dtype=np.int8)
        e = MuLUTEngine(0).configure(2, modes, 4, 4).set_lut_dict(luts)
        assert "stage_tube2_kernel" in e.kernel_name(True)
        want = np.stack([c_oracle.pipeline(luts, 2, modes, 4, im) for im in img])
        for sel in (0, 5, 1):
            e.set_tuning("final_stage_kernel", sel)
            for fixk in (0, 1, 2):        # (1 merges rotation pairs in 16-bit fields: the library takes the default kernel for these lists)
                e.set_tuning("fix_kernel", fixk)
                assert np.array_equal(e.pipeline(dev(img)).cpu().numpy(), want), (modes, val, sel, fixk)
            e.set_tuning("fix_kernel", 0)
            # two channels (the byte-wise output form) and a planar batch (the dword form) of the same images
            got2 = e.pipeline(dev(np.ascontiguousarray(img[..., :2]))).cpu().numpy()
            assert np.array_equal(got2, want[..., :2]), (modes, val, sel, "C=2")
            gotp = e.pipeline(dev(np.ascontiguousarray(img.transpose(0, 3, 1, 2))), layout=0).cpu().numpy()
            assert np.array_equal(gotp, want.transpose(0, 3, 1, 2)), (modes, val, sel, "planar")
        e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("scale,stages,modes", [(3, 2, "sdy"), (3, 1, "s"), (3, 2, "ddyy"), (3, 3, "ys"), (3, 2, "sdysdysd"), (3, 1, "yyyyy"),
                                                (2, 2, "sdysdysd"), (2, 1, "ddddd")])
def test_x3_final_stage_on_the_lds_path(scale, stages, modes):
    """x3 final stages (9-value rows) run on the tube-band kernel family (stage_u1t_kernel<3>: the nine values as ten 16-bit fields, centre
    twice, so that rotations r and r + 2 share accumulators), flagged sites on the site fix-up kernel: extreme and random tables (up to eight
    modes: a merged pair of rotations then fills its unsigned 16-bit field to 65280), smooth + noisy + ragged content, HWC / planar / two
    channels, against the gather kernel (final_stage_kernel 1) and the oracle.  The x2 instance with more than four modes rides along."""
    from mulut_amd import MuLUTEngine
    rng = np.random.default_rng(30 + stages + len(modes) + scale)
    # 136 columns (a multiple of four: the routed launch -- smooth 64 x 64 tiles on the tube kernel, detailed ones on the gather kernel) and 133
    img = np.stack([np.concatenate([natural_image(37, 72, 3, seed=s), rng.integers(0, 256, (37, 64, 3), dtype=np.uint8)], axis=1) for s in (1, 2)])
    for val in (127, -128, None):
        luts = {}
        for st in range(1, stages + 1):
            for m in set(modes):
                if st < stages:
                    luts["s%d_%s" % (st, m)] = synthetic_lut(ord(m) + st, 1)
                else:
                    luts["s%d_%s" % (st, m)] = np.full((17 ** 4, scale * scale), val, np.int8) if val is not None else rng.integers(-128, 128, (17 ** 4, scale * scale), dtype=np.int8)
        e = MuLUTEngine(0).configure(stages, modes, scale, 4).set_lut_dict(luts)
        assert "stage_u1t_kernel<%d>" % scale in e.kernel_name(True)
        want = np.stack([c_oracle.pipeline(luts, stages, modes, scale, im) for im in img])
        assert want.shape == (2, 37 * scale, 136 * scale, 3)
        for sel in (0, 5, 1):           # routed, tube kernel on every tile, gather kernel
            e.set_tuning("final_stage_kernel", sel)
            assert np.array_equal(e.pipeline(dev(img)).cpu().numpy(), want), (modes, val, sel)
            rag = np.ascontiguousarray(img[:, :, 3:])
            assert np.array_equal(e.pipeline(dev(rag)).cpu().numpy(), np.stack([c_oracle.pipeline(luts, stages, modes, scale, im) for im in rag])), (modes, val, sel, "W=133")
            got2 = e.pipeline(dev(np.ascontiguousarray(img[..., :2]))).cpu().numpy()
            assert np.array_equal(got2, want[..., :2]), (modes, val, sel, "C=2")
            gotp = e.pipeline(dev(np.ascontiguousarray(img.transpose(0, 3, 1, 2))), layout=0).cpu().numpy()
            assert np.array_equal(gotp, want.transpose(0, 3, 1, 2)), (modes, val, sel, "planar")
            if sel == 0:            # the routed launch at its extremes: every tile left to the gather kernel, every tile kept
                for thr in (0, 1024):
                    e.set_tuning("final_stage_detail_per_1024", thr)
                    assert np.array_equal(e.pipeline(dev(img)).cpu().numpy(), want), (modes, val, "threshold", thr)
                e.set_tuning("final_stage_detail_per_1024", 8)
            one = np.ascontiguousarray(img[0, :5, :3, :1])          # smaller than a window
            assert np.array_equal(e.pipeline(dev(one)).cpu().numpy(), c_oracle.pipeline(luts, stages, modes, scale, one)), (modes, val, sel, "tiny")
        e.close()


def _strip_rank(rank, world, port, q):
    """one rank of the config-3 rehearsal: real engine, strips + halo, gather on rank 0 (gloo moves host memory)"""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mulut_amd import MuLUTEngine, load_lut_dict
        from mulut_amd.dist import sr_strips
        from mulut_amd.synth import natural_frames
        luts = load_lut_dict(os.path.join(GOLDEN, "luts"), 2, "sdy", 4, 4, "LUT_ft")
        eng = MuLUTEngine(0).configure(2, "sdy", 4, 4).set_lut_dict(luts)
        frame = torch.from_numpy(natural_frames(1, 2160, 3840, 3, 3)[0])          # host memory, LR 2160 x 3840
        frame[700:1500, 1000:2600] = torch.from_numpy(np.random.default_rng(4).integers(0, 256, (800, 1600, 3), dtype=np.uint8))
        out = sr_strips(frame, lambda band, r0, y0, y1, hh: eng.pipeline_rows(band, r0, y0, y1, hh), 4, eng.halo, dst=0,
                        device="cuda:0", via_host=True)
        if rank == 0:
            want = eng.pipeline(frame.cuda())
            q.put((0, bool(torch.equal(out, want)), tuple(out.shape)))
        else:
            q.put((rank, out is None, None))
    finally:
        dist.destroy_process_group()


def test_config3_strips_on_4k_lr_frame_two_ranks():
    """BASELINE config 3 at its real size: an LR 2160x3840 frame (smooth field with a block of noise, so both kernel
    families of both stages run) cut into two strips (+4-row halo), one rank each (both on GPU 0, gloo), HR strips gathered
    into the 8640x15360 frame on rank 0 == the single-GPU result, bit for bit."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_strip_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert got[0] == (0, True, (8640, 15360, 3)) and got[1][1]


@pytest.mark.gpu
def test_bench_multi_rank_path_rehearsal(tmp_path):
    """bench.py's N > 1 path end to end on one device: two ranks under torchrun (gloo through host memory stands in for RCCL, which
    refuses two ranks on one GPU), frames per rank + the config-3 legs (single root and rotating roots), one JSON line from rank 0."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, MULUT_NO_BUILD="1", MULUT_BENCH_BACKEND="gloo")
    port = 29500 + (os.getpid() % 400)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--frames", "2", "--steps", "2", "--warmup", "1",
           "--cpu-crop", "0", "--skip-other", "--lr-h", "270", "--lr-w", "480", "--strip-frames", "2"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0
    sg = rec["config"]["strips_gather"]
    assert sg["ranks_in_group"] == 2 and sg["single_root"]["value"] > 0 and sg["rotate"]["value"] > 0
    # the same facts at the top level of the N > 1 record: the group size the backend saw, the backend, the bytes that crossed the links
    assert rec["ranks_in_group"] == 2 and rec["backend"].startswith("gloo") and rec["scaling"] == "weak"
    assert rec["strips_rotate_value"] == sg["rotate"]["value"] and rec["exchanged_bytes_per_step"] == sg["rotate"]["exchanged_bytes_per_step"] > 0
    assert sg["rotate"]["max_bytes_into_one_rank_per_step"] * 2 == sg["single_root"]["max_bytes_into_one_rank_per_step"]


@pytest.mark.gpu
def test_large_batch_runs_as_sub_batches(eng, shipped_luts):
    """A batch whose stage input passes 2^28 bytes is split by mulut_pipeline (the detailed-tile work lists index 28 bits):
    same bytes as frame-by-frame calls, on detailed content, and no slower path (the anchor-slab kernels still run)."""
    from mulut_amd.synth import noise_frames
    h, w = 1080, 1920
    n = (1 << 28) // (h * w * 3) + 2          # 45 frames: one more sub-batch than fits
    base = dev(noise_frames(3, h, w, 3, 7))
    x = base.repeat((n + 2) // 3, 1, 1, 1)[:n].contiguous()
    out = torch.empty((n, 4 * h, 4 * w, 3), dtype=torch.uint8, device="cuda")
    eng.pipeline(x, out=out)
    ref = eng.pipeline(base)
    for k in range(n):
        assert torch.equal(out[k], ref[k % 3]), k
    d = eng.last_detail_counters()
    assert sum(d["samples_per_anchor"]) > 0          # the last sub-batch went through the anchor slabs


@pytest.mark.gpu
def test_large_strip_and_stage_calls_run_as_sub_launches(eng, shipped_luts):
    """mulut_pipeline_rows (the multi-GPU entry point) and mulut_stage with more than 2^28 bytes of stage input: the final stage
    runs as sub-launches of whole images inside the library, so the detailed tiles still take the anchor-slab kernels (no batch size
    falls to the gather kernels) and every image equals its frame-by-frame result."""
    from mulut_amd.synth import noise_frames
    h, w, y1 = 1080, 1920, 544
    rows = y1 + eng.halo                                       # the strip [0, y1) and its halo below
    n = (1 << 28) // (rows * w * 3) + 2                        # more strips than one final-stage launch indexes
    base = dev(noise_frames(3, h, w, 3, 11))
    band = base[:, :rows].repeat((n + 2) // 3, 1, 1, 1)[:n].contiguous()
    assert band.numel() > (1 << 28)
    out = eng.pipeline_rows(band, 0, 0, y1, h)
    d = eng.last_detail_counters()
    assert sum(d["samples_per_anchor"]) > 0.9 * 3 * y1 * w     # the last sub-launch: (nearly) every sample through the anchor slabs
    ref = eng.pipeline_rows(base[:, :rows].contiguous(), 0, 0, y1, h)
    for k in range(n):
        assert torch.equal(out[k], ref[k % 3]), k
    del out, band
    # the final stage alone on a planar batch of the same size
    mid = eng.stage(1, base, out_layout=0)                     # [3][C][H][W]
    n2 = (1 << 28) // (h * w * 3) + 2
    big = mid.repeat((n2 + 2) // 3, 1, 1, 1)[:n2].contiguous()
    got = eng.stage(2, big, layout=0, out_layout=1)
    d = eng.last_detail_counters()
    assert sum(d["samples_per_anchor"]) > 0.9 * 3 * h * w
    want = eng.stage(2, mid, layout=0, out_layout=1)
    for k in range(n2):
        assert torch.equal(got[k], want[k % 3]), k
    assert torch.equal(want, eng.pipeline(base))


@pytest.mark.gpu
def test_large_batch_captures_into_a_graph_after_reserve(eng, shipped_luts):
    """mulut_reserve sizes every buffer for the sub-launches a batch beyond the 28-bit sample descriptors runs as, so such a call allocates nothing:
    it can be captured into a hipGraph (an allocation under capture would fail it) and the replay equals the eager result, on detailed content."""
    from mulut_amd.synth import noise_frames
    h, w = 1080, 1920
    n = (1 << 28) // (h * w * 3) + 2          # 45 frames: two final-stage sub-launches
    base = dev(noise_frames(3, h, w, 3, 9))
    x = base.repeat((n + 2) // 3, 1, 1, 1)[:n].contiguous()
    out = torch.empty((n, 4 * h, 4 * w, 3), dtype=torch.uint8, device="cuda")
    e = eng
    e.reserve(n, h, w, 3)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        e.pipeline(x[:3], out=out[:3])        # first launches of every kernel outside capture (they raise the kernels' LDS limits)
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        e.pipeline(x, out=out)
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    ref = e.pipeline(base)
    for k in range(n):
        assert torch.equal(out[k], ref[k % 3]), k
