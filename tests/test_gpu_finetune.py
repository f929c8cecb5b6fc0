"""GPU parity of the fine-tune kernels against fixtures produced by running the
reference's MuLUT module (sr/model.py) on CPU (tests/golden/gen_golden_ft.py): forward within 1e-5,
loss and all parameter / input gradients within float-accumulation tolerance."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def synthetic_lut(seed, vnum):
    rng = np.random.default_rng(seed)
    return rng.integers(-127, 128, size=(17 ** 4, vnum), dtype=np.int8)


def build_module(tmp_path, fx, name):
    from mulut_amd.finetune import MuLUT
    stages, scale = [int(v) for v in fx[name + "/cfg"]]
    modes = bytes(fx[name + "/modes"]).decode()
    src = bytes(fx[name + "/lutsrc"]).decode()
    d = tmp_path / name
    d.mkdir()
    for s in range(stages):
        vnum = scale * scale if s + 1 == stages else 1
        for m in modes:
            key = "s%d_%s" % (s + 1, m)
            if src == "shipped":
                t = np.load(os.path.join(GOLDEN, "luts", "LUT_ft_x4_4bit_int8_%s.npy" % key)).reshape(-1, vnum)
            elif src == "s2_s":
                t = np.load(os.path.join(GOLDEN, "luts", "LUT_ft_x4_4bit_int8_s2_s.npy")).reshape(-1, 16)
            else:
                t = synthetic_lut(17 * s + ord(m), vnum)
            np.save(d / ("LUT_x%d_4bit_int8_%s.npy" % (scale, key)), t.astype(np.int8))
    return MuLUT(str(d), stages, modes, upscale=scale, interval=4).cuda(), stages, modes


@pytest.mark.parametrize("name", ["A_s2sdy_x4_u8", "B_s1s_x4_float", "C_s2sd_x2_u8", "D_s2sdy_x4_float"])
def test_forward_and_gradients_match_reference(tmp_path, name):
    fx = np.load(os.path.join(GOLDEN, "ft_fixtures.npz"))
    net, stages, modes = build_module(tmp_path, fx, name)
    x = torch.from_numpy(fx[name + "/x"]).cuda().requires_grad_(True)
    y = net(x)
    want = fx[name + "/y"]
    got = y.detach().cpu().numpy()
    # values are k/255; a mismatch would be >= 1/255, so 1e-5 means "the same rounding decisions everywhere"
    assert np.abs(got - want).max() <= 1e-5, float(np.abs(got - want).max())
    loss = torch.nn.functional.mse_loss(y, torch.from_numpy(fx[name + "/target"]).cuda())
    assert abs(loss.item() - float(fx[name + "/loss"])) <= 1e-6
    loss.backward()
    gx = x.grad.cpu().numpy()
    assert np.allclose(gx, fx[name + "/grad_x"], rtol=2e-4, atol=1e-7), float(np.abs(gx - fx[name + "/grad_x"]).max())
    for s in range(stages):
        for m in modes:
            key = "s%d_%s" % (s + 1, m)
            g = getattr(net, "weight_" + key).grad.cpu().numpy()
            rows = fx[name + "/grad/" + key + "/rows"]
            vals = fx[name + "/grad/" + key + "/vals"]
            dense = np.zeros_like(g)
            dense[rows] = vals
            assert np.allclose(g, dense, rtol=2e-4, atol=1e-7), (key, float(np.abs(g - dense).max()))


def test_module_contract(tmp_path):
    fx = np.load(os.path.join(GOLDEN, "ft_fixtures.npz"))
    net, stages, modes = build_module(tmp_path, fx, "A_s2sdy_x4_u8")
    names = sorted(n for n, _ in net.named_parameters())
    assert names == sorted("weight_s%d_%s" % (s, m) for s in (1, 2) for m in "sdy")      # sr/model.py:57
    assert net.weight_s2_d.shape == (83521, 16) and net.weight_s1_s.shape == (83521, 1)
    exp = net.export_int8()                                                               # sr/3_finetune_lut.py:162-169
    ref = np.load(os.path.join(GOLDEN, "luts", "LUT_ft_x4_4bit_int8_s2_d.npy")).reshape(-1, 16)
    assert exp["s2_d"].dtype == np.int8 and np.array_equal(exp["s2_d"], ref)
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 1, 4, 4))          # CPU tensor: no CPU path
    # one Adam step runs and changes the tables
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    x = torch.rand(4, 1, 16, 16, device="cuda")
    loss = torch.nn.functional.mse_loss(net(x), torch.rand(4, 1, 64, 64, device="cuda"))
    loss.backward()
    before = net.weight_s2_s.detach().clone()
    opt.step()
    assert not torch.equal(before, net.weight_s2_s.detach())


def test_finetune_driver_reduces_loss_and_writes_luts(tmp_path):
    """The driver twin (sr/3_finetune_lut.py) on the Set5 pairs: loss goes down, LUT_ft files appear in the
    reference's int8 format and load back into the inference engine."""
    from mulut_amd import finetune_lut, MuLUTEngine, load_lut_dict
    exp = tmp_path / "exp"
    exp.mkdir()
    for s in (1, 2):
        for m in "sdy":
            t = np.load(os.path.join(GOLDEN, "luts", "LUT_ft_x4_4bit_int8_s%d_%s.npy" % (s, m)))
            # start from a perturbed copy so there is something to learn
            rng = np.random.default_rng(s * 7 + ord(m))
            noisy = np.clip(t.astype(np.int32) + rng.integers(-12, 13, t.shape), -127, 127).astype(np.int8)
            np.save(exp / ("LUT_x4_4bit_int8_s%d_%s.npy" % (s, m)), noisy)
    val_root = str(tmp_path / "bench")                       # {valDir}/Set5/{HR, LR_bicubic/X4}
    os.makedirs(val_root)
    os.symlink(os.path.join(GOLDEN, "Set5"), os.path.join(val_root, "Set5"))
    losses = finetune_lut.main(["--stages", "2", "--modes", "sdy", "-e", str(exp), "--trainDir", os.path.join(GOLDEN, "Set5"),
                                "--batchSize", "16", "--cropSize", "24", "--totalIter", "60", "--displayStep", "20",
                                "--lr0", "1e-3", "--seed", "0", "--valDir", val_root, "--valStep", "60"])
    assert np.mean(losses[-15:]) < np.mean(losses[:15])
    # validation loop (sr/3_finetune_lut.py:23-65): at iterations 1 and 60 every Set5 image was scored and saved
    assert sorted(os.listdir(os.path.join(str(exp), "val", "Set5"))) == sorted(f[:-4] + "_lutft.png" for f in os.listdir(os.path.join(GOLDEN, "Set5", "HR")))
    luts = load_lut_dict(str(exp), 2, "sdy", 4, 4, "LUT_ft")
    assert luts["s2_y"].dtype == np.int8 and luts["s2_y"].shape == (83521, 16)
    eng = MuLUTEngine(0).configure(2, "sdy", 4, 4).set_lut_dict(luts)
    out = eng.pipeline(torch.zeros((8, 8, 3), dtype=torch.uint8, device="cuda"))
    assert out.shape == (32, 32, 3)
    eng.close()


@pytest.mark.parametrize("stages,modes,scale,shape,kind", [
    (1, "y", 4, (2, 3, 5, 7), "u8"), (2, "sdy", 4, (1, 1, 1, 1), "u8"), (3, "sd", 2, (2, 1, 9, 6), "float"),
    (2, "dy", 3, (1, 2, 6, 8), "u8"), (2, "sdy", 4, (1, 1, 10, 10), "extreme"), (2, "s", 1, (1, 1, 7, 5), "float"),
    (2, "sdy", 4, (256, 1, 48, 48), "smooth"), (2, "sdy", 4, (16, 1, 48, 48), "u8"),
    # wide crops: a wave's input-gradient tile does not fit the LDS and the adds go to memory (ft_stage_bwd); planes of 3 rows
    (2, "sdy", 4, (1, 2, 3, 300), "u8"), (2, "sd", 2, (1, 1, 4, 260), "float")])
def test_more_shapes_vs_cpu_oracle(tmp_path, stages, modes, scale, shape, kind):
    """GPU module vs the pinned CPU oracle (oracle/ft_torch.py) on shapes / configurations the fixtures do not hold."""
    from mulut_amd.finetune import MuLUT
    from oracle import ft_torch
    rng = np.random.default_rng(stages * 100 + scale * 10 + len(modes))
    tabs = {}
    for s in range(stages):
        vnum = scale * scale if s + 1 == stages else 1
        for m in modes:
            t = synthetic_lut(3 * s + ord(m), vnum)
            tabs["s%d_%s" % (s + 1, m)] = t
            np.save(tmp_path / ("LUT_x%d_4bit_int8_s%d_%s.npy" % (scale, s + 1, m)), t)
    if kind == "u8":
        x = rng.integers(0, 256, shape).astype(np.float32) / np.float32(255)
    elif kind == "extreme":
        x = rng.choice(np.array([0, 15, 16, 240, 255], np.float32), shape) / np.float32(255)
    elif kind == "smooth":      # BASELINE config 4's batch (bs 256 x 1 x 48 x 48) of photograph-like crops: most passes stay in the tube
        from mulut_amd.synth import natural_frames
        big = natural_frames(1, 1080, 1920, 1, 11)[0, :, :, 0]
        ys, xs = rng.integers(0, 1080 - shape[2], shape[0]), rng.integers(0, 1920 - shape[3], shape[0])
        x = np.stack([big[a:a + shape[2], b:b + shape[3]] for a, b in zip(ys, xs)])[:, None].astype(np.float32) / np.float32(255)
    else:
        x = rng.random(shape, dtype=np.float32)
    tgt = rng.random((shape[0], shape[1], shape[2] * scale, shape[3] * scale), dtype=np.float32)
    # CPU oracle
    wcpu = {k: torch.from_numpy(v.astype(np.float32) / 127.0).requires_grad_(True) for k, v in tabs.items()}
    xc = torch.from_numpy(x).requires_grad_(True)
    yc = ft_torch.forward(wcpu, xc, stages, modes, scale)
    torch.nn.functional.mse_loss(yc, torch.from_numpy(tgt)).backward()
    # GPU module
    net = MuLUT(str(tmp_path), stages, modes, upscale=scale, interval=4).cuda()
    xg = torch.from_numpy(x).cuda().requires_grad_(True)
    yg = net(xg)
    torch.nn.functional.mse_loss(yg, torch.from_numpy(tgt).cuda()).backward()
    assert np.abs(yg.detach().cpu().numpy() - yc.detach().numpy()).max() <= 1e-5
    # Gradients are sums of up to ~10^5 float32 terms per table row whose order is not fixed (float atomics, group-private partial
    # sums): the bar is norm-wise, max |g - ref| <= 2e-5 max |ref| -- measured 6e-6 on the bs-256 batches (tools/ft_err_probe.py),
    # 50x tighter than the rtol 1e-3 / atol 1e-5 this test used through round 2 -- and element-wise rtol 5e-5 where the
    # reference is not itself a cancellation (|ref| > 1 % of its maximum; measured 1.4e-5)
    def close(g, r, what):
        scale = max(float(np.abs(r).max()), 1e-30)
        assert float(np.abs(g - r).max()) <= 2e-5 * scale, (what, float(np.abs(g - r).max()), scale)
        sig = np.abs(r) > 0.01 * scale
        assert np.allclose(g[sig], r[sig], rtol=5e-5, atol=0.0), what
    close(xg.grad.cpu().numpy(), xc.grad.numpy(), "gx")
    for k, w in wcpu.items():
        close(getattr(net, "weight_" + k).grad.cpu().numpy(), w.grad.numpy(), k)


@pytest.mark.parametrize("u,is_last,modes,shape", [(4, 1, "sdy", (3, 1, 13, 10)), (1, 0, "sdy", (2, 2, 9, 11)), (2, 1, "sd", (1, 1, 8, 8)), (3, 1, "y", (1, 1, 6, 5))])
def test_backward_with_the_saved_clamp_mask_equals_the_recomputing_one(u, is_last, modes, shape):
    """The two forms of the stage ABI: mulut_ft_stage_forward/backward (the backward recomputes the stage forward for the clamp mask)
    and the _mask pair (the forward hands the mask over).  Tables with large entries so that the clamp bites on both sides."""
    import ctypes
    from mulut_amd import _native
    lib = _native.load()
    rng = np.random.default_rng(100 * u + len(modes))
    B, C, H, W = shape
    M = len(modes)
    wq = [torch.from_numpy(np.clip(rng.normal(0, 90, (17 ** 4, u * u)).round(), -127, 127).astype(np.float32)).cuda() for _ in modes]
    x = torch.from_numpy(rng.integers(0, 256, shape).astype(np.float32)).cuda()
    gout = torch.from_numpy(rng.standard_normal((B, C, H * u, W * u)).astype(np.float32)).cuda()
    ptrs = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])      # noqa: E731
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    out0, out1 = torch.empty_like(gout), torch.empty_like(gout)
    inside = torch.zeros(shape, dtype=torch.int16, device="cuda")
    assert lib.mulut_ft_stage_forward(0, ptrs(wq), modes.encode(), is_last, u, x.data_ptr(), B, C, H, W, out0.data_ptr(), st) == 0
    assert lib.mulut_ft_stage_forward_mask(0, ptrs(wq), modes.encode(), is_last, u, x.data_ptr(), B, C, H, W, out1.data_ptr(), inside.data_ptr(), st) == 0
    assert torch.equal(out0, out1)
    bits = inside.cpu().numpy().astype(np.uint16)
    clamped = 1.0 - np.mean([(bits >> e) & 1 for e in range(u * u)])
    if is_last:
        assert 0.02 < clamped < 0.98, clamped    # the case exercises both sides of the mask
    else:
        assert clamped == 0.0                    # (a non-final stage cannot leave [0, 255]: |pred / 4M| <= 127, bias 127)
    res = []
    for masked in (False, True):
        gw = [torch.zeros_like(w) for w in wq]
        gx = torch.zeros_like(x)
        if masked:
            rc = lib.mulut_ft_stage_backward_mask(0, ptrs(wq), modes.encode(), is_last, u, x.data_ptr(), gout.data_ptr(), inside.data_ptr(),
                                                  B, C, H, W, ptrs(gw), gx.data_ptr(), st)
        else:
            rc = lib.mulut_ft_stage_backward(0, ptrs(wq), modes.encode(), is_last, u, x.data_ptr(), gout.data_ptr(), B, C, H, W, ptrs(gw), gx.data_ptr(), st)
        assert rc == 0
        res.append([g.cpu().numpy() for g in gw] + [gx.cpu().numpy()])
    for g0, g1 in zip(*res):
        scale = max(float(np.abs(g0).max()), 1e-30)
        assert scale > 1e-6 and float(np.abs(g0 - g1).max()) <= 2e-5 * scale
    assert lib.mulut_ft_stage_forward_mask(0, ptrs(wq), modes.encode(), is_last, u, x.data_ptr(), B, C, H, W, out1.data_ptr(), None, st) == -1
