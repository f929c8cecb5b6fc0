"""GPU twin of the reference's LUT test script (sr/4_test_lut.py:240-340; path handling per the fork's
working copy sr/5_test_lut.py:489-577).

    python -m mulut_amd.test_lut --stages 2 --modes sdy -e ../models/sr_x2sdy

Same inputs ({testDir}/{dataset}/HR/*.png and LR_bicubic/X{scale}/<same name>), same outputs
({resultRoot}/{basename(expDir)}/{dataset}/X{scale}/{stem}_{lutName}_{8-interval}bit.png), same summary line.
The per-image stage/mode/rotation loop runs as one `mulut_pipeline` call on the GPU instead of
`multiprocessing.Pool(24)` over NumPy.
"""
import os

import numpy as np
import torch
from PIL import Image

from .engine import MuLUTEngine
from .metrics import modcrop, psnr, rgb2ycbcr, ssim
from .options import TestOptions


class eltr:
    """Same name and role as the reference's evaluator class (sr/4_test_lut.py:240)."""

    def __init__(self, dataset, opt, engine, device_metrics=None):
        # device_metrics: score on the GPU (engine.eval_y) instead of NumPy/SciPy on the host; same numbers
        self.device_metrics = bool(getattr(opt, "deviceMetrics", False)) if device_metrics is None else bool(device_metrics)
        folder = os.path.join(opt.testDir, dataset, 'HR')
        files = os.listdir(folder)
        files.sort()
        exp_name = opt.expDir.rstrip("/").split("/")[-1]
        result_path = os.path.join(opt.resultRoot, exp_name, dataset, "X{}".format(opt.scale))
        os.makedirs(result_path, exist_ok=True)
        self.result_path = result_path
        self.dataset = dataset
        self.files = files
        self.opt = opt
        self.engine = engine

    def run(self, num_worker=None):
        """The reference maps `_worker` over the files with Pool(24) (:257-259).  Here the GPU does the cascade, so the
        host work is what is left to overlap: `num_worker` threads decode the PNG pairs ahead of the GPU and another
        `num_worker` encode / score finished images behind it (PIL and zlib release the GIL), while the main thread feeds
        the device.  num_worker <= 1 runs the files strictly one after the other."""
        import time
        nw = int(num_worker if num_worker is not None else getattr(self.opt, "ioWorkers", 4))
        t0 = time.perf_counter()
        if nw <= 1 or len(self.files) < 2:
            psnr_ssim_s = [self._worker(i) for i in range(len(self.files))]
        else:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(nw) as dec, ThreadPoolExecutor(nw) as enc:
                loads = [dec.submit(self._load, i) for i in range(len(self.files))]
                done = []
                for i, fut in enumerate(loads):
                    img_lr, img_gt = fut.result()
                    dev_out = self.super_resolve_device(img_lr)
                    if self.device_metrics:
                        gt = torch.from_numpy(np.ascontiguousarray(img_gt)).to(self.engine.device)
                        score = list(self.engine.eval_y(gt, dev_out, self.opt.scale))
                        done.append(enc.submit(self._finish, i, dev_out.cpu().numpy(), None, score))
                    else:
                        done.append(enc.submit(self._finish, i, dev_out.cpu().numpy(), img_gt, None))
                psnr_ssim_s = [f.result() for f in done]
        self.seconds = time.perf_counter() - t0
        if getattr(self.opt, "timing", False):
            print('Dataset {} | {} images in {:.2f} s end to end ({:.2f} images/s, {} I/O threads each way)'.format(
                self.dataset, len(self.files), self.seconds, len(self.files) / max(self.seconds, 1e-9), nw))
        arr = np.asarray(psnr_ssim_s)
        print('Dataset {} | AVG LUT PSNR: {:.2f} SSIM: {:.4f}'.format(self.dataset, np.mean(arr[:, 0]),
                                                                      np.mean(arr[:, 1])))
        return arr

    def super_resolve_device(self, img_lr):
        """uint8 HWC (or HW gray -> replicated to 3 channels, :268-270) -> uint8 HWC tensor on the GPU."""
        if img_lr.ndim == 2:
            img_lr = np.stack([img_lr] * 3, axis=2)
        x = torch.from_numpy(np.ascontiguousarray(img_lr)).to(self.engine.device)
        return self.engine.pipeline(x)

    def super_resolve(self, img_lr):
        return self.super_resolve_device(img_lr).cpu().numpy()

    def _load(self, i):
        """decode the LR / HR pair of file i (:265-277)"""
        opt = self.opt
        img_lr = np.array(Image.open(
            os.path.join(opt.testDir, self.dataset, 'LR_bicubic/X{}'.format(opt.scale), self.files[i])))
        img_gt = np.array(Image.open(os.path.join(opt.testDir, self.dataset, 'HR', self.files[i])))
        img_gt = modcrop(img_gt, opt.scale)
        if img_gt.ndim == 2:
            img_gt = np.stack([img_gt] * 3, axis=2)
        return img_lr, img_gt

    def _finish(self, i, img_out, img_gt, score):
        """save the result (:309-312) and, unless the device scored it already, compute Y-PSNR / SSIM on the host (:313-316)"""
        opt = self.opt
        Image.fromarray(img_out).save(os.path.join(
            self.result_path, '{}_{}_{}bit.png'.format(self.files[i].split('/')[-1][:-4], opt.lutName,
                                                       8 - opt.interval)))
        if score is not None:
            return score
        y_gt, y_out = rgb2ycbcr(img_gt)[:, :, 0], rgb2ycbcr(img_out)[:, :, 0]
        return [psnr(y_gt, y_out, opt.scale), ssim(y_gt, y_out)]

    def _worker(self, i):
        img_lr, img_gt = self._load(i)
        dev_out = self.super_resolve_device(img_lr)
        if self.device_metrics:
            gt = torch.from_numpy(np.ascontiguousarray(img_gt)).to(self.engine.device)
            return self._finish(i, dev_out.cpu().numpy(), None, list(self.engine.eval_y(gt, dev_out, self.opt.scale)))
        return self._finish(i, dev_out.cpu().numpy(), img_gt, None)


def build_engine(opt):
    eng = MuLUTEngine(opt.device)
    eng.configure(opt.stages, opt.modes, opt.scale, opt.interval)
    eng.load_luts(opt.expDir, opt.lutName)     # sr/4_test_lut.py:322-333
    return eng


def main(argv=None):
    opt = TestOptions().parse(argv)
    engine = build_engine(opt)
    results = {}
    for dataset in opt.datasets.split(","):   # reference: all_datasets = ['Set5'] (:336)
        results[dataset] = eltr(dataset, opt, engine).run()
    return results


if __name__ == "__main__":
    main()
