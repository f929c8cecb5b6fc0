"""Single-image convenience API of the fork's working test script (sr/5_test_lut.py:241-662), on the GPU.

Same function names, arguments, return values and error behaviour; `lutDict` is what `load_luts` returns (here a
configured MuLUTEngine, which owns the device copies of the tables, instead of a dict of float32 arrays).

    from mulut_amd.single import main_gui
    out_path, psnr, ssim = main_gui("lr.png", "sr.png", stages=2, modes="sdy", scale=4, exp_dir="models/sr_x2sdy")
"""
import os
from types import SimpleNamespace

import numpy as np
import torch
from PIL import Image

from .engine import MuLUTEngine
from .metrics import modcrop, psnr as _psnr, rgb2ycbcr, ssim as _ssim


def create_simple_options(stages=2, modes="sdy", scale=2, interval=4, exp_dir="../models/sr_x2sdy",
                          lut_name="MuLUT", test_dir=None, result_root="../temp_output"):
    """sr/5_test_lut.py:455-486 (same defaults, including scale=2 and lut_name="MuLUT")."""
    opt = SimpleNamespace()
    opt.stages = stages
    opt.modes = list(modes)
    opt.scale = scale
    opt.interval = interval
    opt.expDir = exp_dir
    opt.lutName = lut_name
    opt.testDir = test_dir
    opt.resultRoot = result_root
    return opt


def load_luts(opt, device=0):
    """sr/5_test_lut.py:417-452: loads {expDir}/{lutName}_x{scale}_{8-interval}bit_int8_s{stage}_{mode}.npy for every
    stage and mode (FileNotFoundError if one is missing) -- into a configured engine."""
    eng = MuLUTEngine(device)
    eng.configure(opt.stages, "".join(opt.modes), opt.scale, opt.interval)
    eng.load_luts(opt.expDir, opt.lutName)
    return eng


def _read_rgb(path):
    img = np.array(Image.open(path))
    if img.ndim == 2:                      # gray -> three equal channels (sr/5_test_lut.py:262-265)
        img = np.stack([img] * 3, axis=2)
    return np.ascontiguousarray(img[:, :, :3].astype(np.uint8))


def _run(image_path, output_path, lutDict):
    if not os.path.exists(image_path):
        raise FileNotFoundError("Input image not found: {}".format(image_path))
    x = torch.from_numpy(_read_rgb(image_path)).to(lutDict.device)
    dev_out = lutDict.pipeline(x)
    img_out = dev_out.cpu().numpy()
    out_dir = os.path.dirname(output_path)
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
    Image.fromarray(img_out).save(output_path)
    return dev_out, img_out


def process_single_image(image_path, output_path, opt, lutDict):
    """sr/5_test_lut.py:241-323 -> (output_path, None, None)."""
    _run(image_path, output_path, lutDict)
    return output_path, None, None


def process_single_image_with_gt(image_path, gt_path, output_path, opt, lutDict, device_metrics=False):
    """sr/5_test_lut.py:326-414 -> (output_path, psnr, ssim); Y-channel scores with shave = scale."""
    dev_out, img_out = _run(image_path, output_path, lutDict)
    img_gt = modcrop(np.array(Image.open(gt_path)), opt.scale)       # :353-358
    if img_gt.ndim == 2:
        img_gt = np.stack([img_gt] * 3, axis=2)
    img_gt = np.ascontiguousarray(img_gt[:, :, :3].astype(np.uint8))
    if device_metrics:
        p, s = lutDict.eval_y(torch.from_numpy(np.ascontiguousarray(img_gt)).to(lutDict.device), dev_out, opt.scale)
    else:
        y_gt, y_out = rgb2ycbcr(img_gt)[:, :, 0], rgb2ycbcr(img_out)[:, :, 0]
        p, s = _psnr(y_gt, y_out, opt.scale), _ssim(y_gt, y_out)
    return output_path, p, s


def main_gui(input_image_path, output_path=None, stages=2, modes="sdy", scale=2, exp_dir="../models/sr_x2sdy", gt_path=None):
    """sr/5_test_lut.py:581-621."""
    opt = create_simple_options(stages=stages, modes=modes, scale=scale, exp_dir=exp_dir)
    lutDict = load_luts(opt)
    if output_path is None:
        input_name = os.path.splitext(os.path.basename(input_image_path))[0]
        output_dir = os.path.join(opt.resultRoot, "gui_results")
        os.makedirs(output_dir, exist_ok=True)
        output_path = os.path.join(output_dir, "{}_sr_x{}.png".format(input_name, scale))
    if gt_path and os.path.exists(gt_path):
        return process_single_image_with_gt(input_image_path, gt_path, output_path, opt, lutDict)
    return process_single_image(input_image_path, output_path, opt, lutDict)


def test_single_image_direct(input_image_path, output_path, stages=2, modes="sdy", scale=2, exp_dir="../models/sr_x2sdy",
                             interval=4, lut_name="LUT_ft"):
    """sr/5_test_lut.py:624-662 -> output path."""
    opt = create_simple_options(stages=stages, modes=modes, scale=scale, exp_dir=exp_dir,
                                result_root=os.path.dirname(output_path))
    opt.interval = interval
    opt.lutName = lut_name
    lutDict = load_luts(opt)
    result_path, _, _ = process_single_image(input_image_path, output_path, opt, lutDict)
    return result_path
