"""The reference's test-time command line (common/option.py:15-29,189-199), kept flag for flag.

Deviations, both documented in DESIGN.md:
  * ``parse()`` does not copy every ``*.py`` under cwd into ``<expDir>/code`` (common/option.py:104-110,155-156);
    pass ``--saveCode`` to get that side effect back.  ``--debug`` is accepted and means what it meant.
  * new optional flags (``--device``, ``--datasets``, ``--batch``, ``--ioWorkers``, ``--timing``) default to the reference's behaviour.
"""
import argparse
import os


class TestOptions:
    def __init__(self, debug=False):
        self.debug = debug
        self.isTrain = False

    def initialize(self, parser):
        # experiment specifics (BaseOptions.initialize, common/option.py:13-31)
        parser.add_argument('--model', type=str, default='SRNets')
        parser.add_argument('--task', '-t', type=str, default='sr')
        parser.add_argument('--scale', '-r', type=int, default=4, help="up scale factor")
        parser.add_argument('--sigma', '-s', type=int, default=25, help="noise level")
        parser.add_argument('--qf', '-q', type=int, default=20, help="deblocking quality factor")
        parser.add_argument('--nf', type=int, default=64, help="number of filters of convolutional layers")
        parser.add_argument('--stages', type=int, default=2, help="stages of MuLUT")
        parser.add_argument('--modes', type=str, default='sdy', help="sampling modes to use in every stage")
        parser.add_argument('--interval', type=int, default=4, help='N bit uniform sampling')
        parser.add_argument('--modelRoot', type=str, default='../models')
        parser.add_argument('--expDir', '-e', type=str, default='', help="experiment folder")
        parser.add_argument('--load_from_opt_file', action='store_true', default=False)
        parser.add_argument('--debug', default=False, action='store_true')
        # TestOptions.initialize, common/option.py:190-196
        parser.add_argument('--loadIter', '-i', type=int, default=200000)
        parser.add_argument('--testDir', type=str, default='../data/SRBenchmark')
        parser.add_argument('--resultRoot', type=str, default='../results')
        parser.add_argument('--lutName', type=str, default='LUT_ft')
        # additions
        parser.add_argument('--device', type=int, default=0, help="GPU index")
        parser.add_argument('--datasets', type=str, default='Set5', help="comma separated (reference: ['Set5'])")
        parser.add_argument('--deviceMetrics', action='store_true', default=False,
                            help="score PSNR/SSIM on the GPU (same numbers; saves the host-side SciPy convolutions)")
        parser.add_argument('--ioWorkers', type=int, default=4,
                            help='PNG decode threads ahead of the GPU and encode / score threads behind it (1 = strictly serial)')
        parser.add_argument('--timing', action='store_true', default=False, help='also print end-to-end images/s per dataset')
        parser.add_argument('--saveCode', action='store_true', default=False,
                            help="copy *.py under cwd into <expDir>/code like the reference's parse()")
        return parser

    def parse(self, argv=None):
        parser = self.initialize(argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter))
        opt = parser.parse_args([] if self.debug else argv)
        opt.isTrain = False
        opt.flag = opt.scale if "sr" in opt.task else (opt.sigma if "dn" in opt.task else opt.qf if "db" in opt.task else "0")
        if opt.expDir == '':
            # common/option.py:119-131: fresh ../models/debug/expr_N
            opt.modelDir = os.path.join(opt.modelRoot, "debug")
            os.makedirs(opt.modelDir, exist_ok=True)
            count = 1
            while os.path.isdir(os.path.join(opt.modelDir, 'expr_{}'.format(count))):
                count += 1
            opt.expDir = os.path.join(opt.modelDir, 'expr_{}'.format(count))
            os.mkdir(opt.expDir)
        elif not os.path.isdir(opt.expDir):
            os.makedirs(opt.expDir)
        opt.modelPath = os.path.join(opt.expDir, "Model.pth")
        if opt.saveCode and not opt.debug:
            import shutil
            from pathlib import Path
            for f in Path("./").rglob("*.py"):
                trg = os.path.join(opt.expDir, "code", f)
                os.makedirs(os.path.dirname(trg), exist_ok=True)
                shutil.copy(f, trg, follow_symlinks=False)
        self.opt = opt
        return opt
