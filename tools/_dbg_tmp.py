import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from PIL import Image
from mulut_amd import MuLUTEngine, load_lut_dict
from oracle import c_oracle
G = "tests/golden"
luts = load_lut_dict(os.path.join(G, "luts"), 2, "sdy", 4, 4, "LUT_ft")
e = MuLUTEngine(0).configure(2, "sdy", 4, 4).set_lut_dict(luts)
d = os.path.join(G, "Set5", "LR_bicubic", "X4")
for rep in range(2):
  for fn in sorted(os.listdir(d)):
    im = np.array(Image.open(os.path.join(d, fn)))
    want = c_oracle.pipeline(luts, 2, "sdy", 4, im)
    for key, val in (("default", None), ("detail_kernel", 1), ("final_stage_kernel", 5)):
        if val is not None: e.set_tuning(key, val)
        got = e.pipeline(torch.from_numpy(im).cuda()).cpu().numpy()
        if val is not None: e.set_tuning(key, 0)
        bad = (got != want)
        print(rep, fn, im.shape, key, int(bad.sum()), (tuple(int(v) for v in np.argwhere(bad)[0]) if bad.any() else ""), e.last_detail_counters()["fix_pixels"] if key == "default" else "")
