#!/usr/bin/env python3
"""Drop-in for the reference's `python 3_finetune_lut.py --stages 2 --modes sdy -e <expDir>` (run from sr/):
the LUT-aware fine-tuning driver on the HIP forward / backward kernels (mulut_amd.finetune_lut)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mulut_amd.finetune_lut import main  # noqa: E402

if __name__ == "__main__":
    main()
