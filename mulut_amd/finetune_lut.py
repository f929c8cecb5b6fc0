"""GPU twin of the reference's LUT fine-tuning driver (sr/3_finetune_lut.py:68-172).

    python -m mulut_amd.finetune_lut --stages 2 --modes sdy -e <expDir> --trainDir <DIV2K-like dir> [--totalIter 2000]

Reads the transferred tables ``LUT_x{scale}_{interval}bit_int8_s{stage}_{mode}.npy`` from expDir (sr/model.py:51-53),
optimises them with Adam + the reference's cosine LambdaLR on random single-channel crops with the reference's
rigid augmentation (sr/data.py:96-124), and writes ``LUT_ft_x{scale}_{interval}bit_int8_s{stage}_{mode}.npy``
(sr/3_finetune_lut.py:162-169).  The model is ``mulut_amd.finetune.MuLUT`` (HIP forward/backward kernels).
Training pairs: ``{trainDir}/HR/<stem>.png`` with ``{trainDir}/LR/X{scale}/<stem>x{scale}.png`` (DIV2K layout) or
``{trainDir}/LR_bicubic/X{scale}/<stem>.png`` (benchmark layout).  ``valid_steps`` is the reference's validation loop
(:23-65): every ``--valStep`` iterations (and at iteration 1) each benchmark image goes through the module, the result is
saved as ``{valoutDir}/{dataset}/{last '_'-token of the stem}_lutft.png`` (the reference's naming) and Y-PSNR / SSIM are averaged per dataset -- computed on the device
(``mulut_eval_y``), logged with the reference's line.
"""
import argparse
import math
import os
import random
import time

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

from .finetune import MuLUT


class CropProvider:
    """Random (LR crop, HR crop) batches, one colour channel each, flips + rot90 as sr/data.py:96-124."""

    def __init__(self, path, scale, patch, batch, seed=None):
        self.scale, self.sz, self.batch = scale, patch, batch
        self.rng = random.Random(seed)
        hr_dir = os.path.join(path, "HR")
        self.pairs = []
        for fn in sorted(os.listdir(hr_dir)):
            stem = fn[:-4]
            for lr in (os.path.join(path, "LR", "X%d" % scale, "%sx%d.png" % (stem, scale)),
                       os.path.join(path, "LR_bicubic", "X%d" % scale, fn)):
                if os.path.exists(lr):
                    hr_im = np.array(Image.open(os.path.join(hr_dir, fn)))
                    lr_im = np.array(Image.open(lr))
                    if hr_im.ndim == 2:
                        hr_im, lr_im = hr_im[:, :, None], lr_im[:, :, None]
                    if min(lr_im.shape[:2]) >= patch:
                        self.pairs.append((lr_im, hr_im))
                    break
        if not self.pairs:
            raise FileNotFoundError("no HR/LR training pairs with LR >= %d px under %s" % (patch, path))

    def next(self):
        ims, lbs = [], []
        for _ in range(self.batch):
            im, lb = self.rng.choice(self.pairs)
            i = self.rng.randint(0, im.shape[0] - self.sz)
            j = self.rng.randint(0, im.shape[1] - self.sz)
            c = self.rng.randrange(im.shape[2])
            s = self.scale
            lb = lb[i * s:(i + self.sz) * s, j * s:(j + self.sz) * s, c]
            im = im[i:i + self.sz, j:j + self.sz, c]
            if self.rng.uniform(0, 1) < 0.5:
                lb, im = np.fliplr(lb), np.fliplr(im)
            if self.rng.uniform(0, 1) < 0.5:
                lb, im = np.flipud(lb), np.flipud(im)
            k = self.rng.choice([0, 1, 2, 3])
            lbs.append(np.rot90(lb, k).astype(np.float32)[None] / 255.0)
            ims.append(np.rot90(im, k).astype(np.float32)[None] / 255.0)
        return torch.from_numpy(np.stack(ims)).cuda(), torch.from_numpy(np.stack(lbs)).cuda()


def valid_steps(net, opt, it, log=print):
    """Twin of sr/3_finetune_lut.py:23-65: datasets under {valDir}/{dataset}/HR with LR_bicubic/X{scale}; a dataset that is
    not on disk is skipped (the reference's Provider would have failed at start-up instead)."""
    import ctypes
    from . import _native
    lib = _native.load()
    datasets = ['Set5', 'Set14'] if opt.debug else ['Set5', 'Set14', 'B100', 'Urban100', 'Manga109']
    was_training = net.training
    net.eval()
    results = {}
    with torch.no_grad():
        for ds in datasets:
            hr_dir = os.path.join(opt.valDir, ds, "HR")
            if not os.path.isdir(hr_dir):
                continue
            out_dir = os.path.join(opt.valoutDir, ds)
            os.makedirs(out_dir, exist_ok=True)
            psnrs, ssims = [], []
            for fn in sorted(os.listdir(hr_dir)):
                lb = np.array(Image.open(os.path.join(hr_dir, fn)))
                im = np.array(Image.open(os.path.join(opt.valDir, ds, "LR_bicubic", "X%d" % opt.scale, fn)))
                if im.ndim == 2:                                   # grey images are replicated to three channels (sr/data.py)
                    im, lb = np.stack([im] * 3, 2), np.stack([lb] * 3, 2)
                x = torch.from_numpy(np.ascontiguousarray(im.transpose(2, 0, 1)[None]).astype(np.float32) / 255.0).cuda()
                pred = net(x) * 255.0
                pred_u8 = torch.round(torch.clamp(pred[0].permute(1, 2, 0), 0, 255)).to(torch.uint8).contiguous()
                H, W = pred_u8.shape[:2]
                gt = torch.from_numpy(np.ascontiguousarray(lb[:H, :W, :3])).cuda()
                n = int(lib.mulut_eval_ws_doubles(H, W))
                ws = torch.empty(n, dtype=torch.float64, device="cuda")
                ps, ss = ctypes.c_double(), ctypes.c_double()
                rc = lib.mulut_eval_y(pred_u8.device.index, gt.data_ptr(), pred_u8.data_ptr(), H, W, int(opt.scale), ws.data_ptr(), n,
                                      ctypes.byref(ps), ctypes.byref(ss),
                                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
                if rc:
                    raise RuntimeError(lib.mulut_strerror(rc).decode())
                psnrs.append(ps.value)
                ssims.append(ss.value)
                # the reference names the file after the LAST '_'-separated token of "{dataset}_{stem}" (sr/3_finetune_lut.py:41,61):
                # Urban100's img_001.png becomes 001_lutft.png -- kept, a drop-in writes the same names
                Image.fromarray(pred_u8.cpu().numpy()).save(os.path.join(out_dir, '{}_lutft.png'.format((ds + '_' + fn[:-4]).split('_')[-1])))
            if psnrs:
                results[ds] = (float(np.mean(psnrs)), float(np.mean(ssims)))
                log('Iter {} | Dataset {} | AVG PSNR: {:02f}, AVG: SSIM: {:04f}'.format(it, ds, results[ds][0], results[ds][1]))
    net.train(was_training)
    return results


def build_parser():
    p = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    # the flags sr/3_finetune_lut.py reads from TrainOptions (common/option.py:15-29,160-187)
    p.add_argument('--scale', '-r', type=int, default=4)
    p.add_argument('--stages', type=int, default=2)
    p.add_argument('--modes', type=str, default='sdy')
    p.add_argument('--interval', type=int, default=4)
    p.add_argument('--expDir', '-e', type=str, required=True)
    p.add_argument('--batchSize', type=int, default=32)
    p.add_argument('--cropSize', type=int, default=48)
    p.add_argument('--trainDir', type=str, default="../data/DIV2K")
    p.add_argument('--valDir', type=str, default='../data/SRBenchmark')
    p.add_argument('--valStep', type=int, default=2000, help='validate every N iterations (and at iteration 1); 0 = never')
    p.add_argument('--valoutDir', type=str, default=None, help='default: {expDir}/val')
    p.add_argument('--debug', default=False, action='store_true')
    p.add_argument('--totalIter', type=int, default=200000)
    p.add_argument('--displayStep', type=int, default=100)
    p.add_argument('--lr0', type=float, default=1e-3)
    p.add_argument('--lr1', type=float, default=1e-4)
    p.add_argument('--weightDecay', type=float, default=0)
    p.add_argument('--seed', type=int, default=None)
    return p


def finetune(opt, log=print):
    net = MuLUT(opt.expDir, opt.stages, list(opt.modes), upscale=opt.scale, interval=opt.interval).cuda()
    params = [p for p in net.parameters() if p.requires_grad]
    optim = torch.optim.Adam(params, lr=opt.lr0, betas=(0.9, 0.999), eps=1e-8, weight_decay=opt.weightDecay, amsgrad=False,
                             fused=True)      # the same update as one launch over the six tables (the default is ~10 per step)
    if opt.lr1 < 0:                                                        # sr/3_finetune_lut.py:89-95
        lf = lambda x: (((1 + math.cos(x * math.pi / opt.totalIter)) / 2) ** 1.0) * 0.8 + 0.2   # noqa: E731
    else:
        lr_b = opt.lr1 / opt.lr0
        lr_a = 1 - lr_b
        lf = lambda x: (((1 + math.cos(x * math.pi / opt.totalIter)) / 2) ** 1.0) * lr_a + lr_b   # noqa: E731
    sched = torch.optim.lr_scheduler.LambdaLR(optim, lr_lambda=lf)
    data = CropProvider(opt.trainDir, opt.scale, opt.cropSize, opt.batchSize, opt.seed)
    accum, t_run, losses = 0.0, 0.0, []
    if getattr(opt, "valoutDir", None) is None:
        opt.valoutDir = os.path.join(opt.expDir, "val")
    for i in range(1, opt.totalIter + 1):
        im, lb = data.next()
        st = time.time()
        optim.zero_grad()
        loss = F.mse_loss(net(im), lb)
        loss.backward()
        optim.step()
        sched.step()
        accum += loss.item()
        t_run += time.time() - st
        losses.append(loss.item())
        if i % opt.displayStep == 0:
            log("{} | Iter:{:6d}, Sample:{:6d}, GPixel:{:.2e}, rT:{:.4f}".format(opt.expDir, i, i * opt.batchSize,
                                                                               accum / opt.displayStep, t_run / opt.displayStep))
            accum, t_run = 0.0, 0.0
        if getattr(opt, "valStep", 0) and (i % opt.valStep == 0 or i == 1) and os.path.isdir(getattr(opt, "valDir", "")):   # :152-158
            valid_steps(net, opt, i, log)
    for key, table in net.export_int8().items():                          # :162-169
        np.save(os.path.join(opt.expDir, "LUT_ft_x{}_{}bit_int8_{}.npy".format(opt.scale, opt.interval, key)), table)
    log("Finetuned LUT saved to {}".format(opt.expDir))
    return losses


def main(argv=None):
    opt = build_parser().parse_args(argv)
    if opt.debug:                                              # common/option.py:147-151
        opt.displayStep, opt.valStep, opt.totalIter = 10, 50, min(opt.totalIter, 200)
    return finetune(opt)


if __name__ == "__main__":
    main()
