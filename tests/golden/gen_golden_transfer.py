#!/usr/bin/env python3
"""Golden data for the LUT producer (reference sr/2_transfer_to_lut.py), made by RUNNING THE REFERENCE's network code
(sr/model.py SRNets + common/network.py) on the CPU in this container -- same rules as gen_golden.py.

    python tests/golden/gen_golden_transfer.py      # rewrites tests/golden/transfer_fixtures.npz (+ Model_200000.pth copy)

What is recorded (data only):
  * the trained weights of the shipped checkpoint models/sr_x2sdy/Model_200000.pth as plain arrays (`w/<state_dict key>`),
    and the checkpoint file itself (a pickle of tensors; needed to test loading whole-module checkpoints),
  * for every (stage, mode): sha256 of the int8 table `round(clamp(net(grid), -1, 1) * 127)` over the full 17^4 grid in the
    script's enumeration order (:12-41, a slowest), its shape, and 2048 sampled rows (`rows/...`, indices in `idx`),
  * the same for a seeded random-init 1-stage x2 model with nf=8 (`tiny/...`) incl. its weights.
The reference builds the grid with .cuda() (:19-33); here the identical tensor is built on the CPU.
"""
import hashlib
import os
import shutil
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def grid(interval):
    base = torch.arange(0, 257, 2 ** interval)
    base[-1] -= 1
    L = base.size(0)
    g = torch.cartesian_prod(base, base, base, base)            # a slowest ... d fastest, as :19-37
    return g.reshape(-1, 1, 2, 2).float() / 255.0, L


def main():
    if not os.path.isdir(REF):
        raise SystemExit("gen_golden_transfer.py: /root/reference not present")
    cv2 = types.ModuleType("cv2")
    sys.modules.setdefault("cv2", cv2)
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "sr"))
    os.chdir(os.path.join(REF, "sr"))
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_transfer", os.path.join(REF, "sr", "2_transfer_to_lut.py"))
    ref_transfer = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_transfer)                       # __main__ guard keeps the script body from running
    import model as ref_model

    fx = {}
    x, L = grid(4)
    rng = np.random.default_rng(3)
    idx = np.sort(rng.choice(L ** 4, 2048, replace=False))
    fx["idx"] = idx

    def tables(net, stages, modes, prefix):
        net.eval()
        with torch.no_grad():
            for s in range(stages):
                for mode in modes:
                    t = x if mode == "s" else ref_transfer.get_mode_input_tensor(x, mode)
                    out = torch.cat([net(t[b:b + 8192], stage=s + 1, mode=mode) for b in range(0, t.shape[0], 8192)])
                    q = torch.round(torch.clamp(out, -1, 1) * 127).numpy().astype(np.int8)      # :108-109
                    key = "%s/s%d_%s" % (prefix, s + 1, mode)
                    fx[key + "/sha256"] = np.frombuffer(hashlib.sha256(q.tobytes()).digest(), dtype=np.uint8)
                    fx[key + "/shape"] = np.array(q.shape)
                    fx[key + "/rows"] = q.reshape(q.shape[0], -1)[idx]
                    print(key, q.shape, q.min(), q.max())

    ck = os.path.join(REF, "models", "sr_x2sdy", "Model_200000.pth")
    lm = torch.load(ck, map_location="cpu", weights_only=False)
    for k, v in lm.state_dict().items():
        fx["w/" + k] = v.numpy()
    net = ref_model.SRNets(nf=64, scale=4, modes=list("sdy"), stages=2)
    net.load_state_dict(lm.state_dict(), strict=True)
    tables(net, 2, "sdy", "shipped")
    shutil.copyfile(ck, os.path.join(HERE, "Model_200000.pth"))

    torch.manual_seed(11)
    tiny = ref_model.SRNets(nf=8, scale=2, modes=list("sdy"), stages=1)
    for k, v in tiny.state_dict().items():
        fx["tinyw/" + k] = v.numpy()
    tables(tiny, 1, "sdy", "tiny")
    np.savez_compressed(os.path.join(HERE, "transfer_fixtures.npz"), **fx)


if __name__ == "__main__":
    main()
