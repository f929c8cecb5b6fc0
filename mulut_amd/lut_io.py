"""The reference's on-disk LUT format: NPY v1 int8, C order, (83521, v_num) or (83521, 1, u, u).

File naming follows the READER (sr/4_test_lut.py:331-332): ``{lutName}_x{scale}_{8-interval}bit_int8_s{stage}_{mode}.npy``.
The writers (sr/2_transfer_to_lut.py:114-115, sr/3_finetune_lut.py:165-167) use ``{interval}bit``; the two agree only at
the default ``--interval 4``, which is the only interval supported here.
"""
import os

import numpy as np

L_ROWS = 17 ** 4


def lut_file_name(lut_name, scale, interval, stage, mode):
    return "{}_x{}_{}bit_int8_s{}_{}.npy".format(lut_name, scale, 8 - interval, stage, mode)


def load_lut_dict(exp_dir, stages, modes, scale=4, interval=4, lut_name="LUT_ft"):
    """{ 's{stage}_{mode}': int8 array (83521, v_num) } -- the keys of the reference's ``lutDict`` (:330).
    A missing file raises FileNotFoundError (as np.load does at :333); a wrong shape raises ValueError
    (as the reference's reshape does)."""
    out = {}
    for s in range(stages):
        v_num = scale * scale if (s + 1) == stages else 1
        for mode in modes:
            path = os.path.join(exp_dir, lut_file_name(lut_name, scale, interval, s + 1, mode))
            arr = np.load(path)                       # FileNotFoundError here, in the reference's order
            if arr.dtype != np.int8:
                # the reference casts whatever it loads to float32; tables are int8 by construction
                if not np.array_equal(arr, np.round(arr)) or np.abs(arr).max() > 127:
                    raise ValueError("LUT {} is not int8-valued".format(path))
                arr = arr.astype(np.int8)
            out["s{}_{}".format(s + 1, mode)] = np.ascontiguousarray(arr.reshape(-1, v_num))
    # the reference never checks the row count (a (83521,16) file loaded as a non-final stage just
    # becomes a (1336336,1) table and indexes garbage); here that is an error
    for key, arr in out.items():
        if arr.shape[0] != L_ROWS:
            raise ValueError("LUT {} has shape {}, expected ({}, v_num)".format(key, arr.shape, L_ROWS))
    return out


def synthetic_lut(seed, vnum):
    """Seeded int8 table for configurations no shipped LUT exists for (deep cascades, x2, ...)."""
    rng = np.random.default_rng(seed)
    return rng.integers(-127, 128, size=(L_ROWS, vnum), dtype=np.int8)


def inspect_lut_dir(exp_dir, scale=4, interval=4):
    """Describe every ``*_x{scale}_*bit_int8_s{stage}_{mode}.npy`` table in a directory: both the reader-side
    (``LUT_ft_x4_4bit_...``) and writer-side (``LUT_x4_4bit_...``, sr/2_transfer_to_lut.py:114-116) names, the
    ``(83521, v_num)`` and ``(83521, 1, u, u)`` shape variants, value range, and the share of rows inside the diagonal
    band the final-stage kernel keeps in LDS.  Returns a list of dicts (and is what ``python -m mulut_amd.lut_io`` prints)."""
    import re
    out = []
    for fn in sorted(os.listdir(exp_dir)):
        m = re.match(r"(.+)_x(\d+)_(\d+)bit_int8_s(\d+)_([a-z])\.npy$", fn)
        if not m:
            continue
        arr = np.load(os.path.join(exp_dir, fn))
        rec = {"file": fn, "lut_name": m.group(1), "scale": int(m.group(2)), "bits": int(m.group(3)),
               "stage": int(m.group(4)), "mode": m.group(5), "dtype": str(arr.dtype), "shape": tuple(arr.shape)}
        flat = arr.reshape(arr.shape[0], -1) if arr.ndim >= 2 else arr.reshape(-1, 1)
        rec["rows_ok"] = flat.shape[0] == L_ROWS
        rec["v_num"] = int(flat.shape[1])
        rec["upscale"] = int(round(flat.shape[1] ** 0.5)) if int(round(flat.shape[1] ** 0.5)) ** 2 == flat.shape[1] else None
        rec["min"], rec["max"] = int(flat.min()), int(flat.max())
        rec["int8_valued"] = bool(arr.dtype == np.int8 or (np.array_equal(arr, np.round(arr)) and np.abs(arr).max() <= 128))
        rec["uses_minus128"] = bool(flat.min() == -128)
        out.append(rec)
    return out


if __name__ == "__main__":
    import json
    import sys
    for r in inspect_lut_dir(sys.argv[1] if len(sys.argv) > 1 else "."):
        print(json.dumps(r))
