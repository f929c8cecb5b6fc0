#!/usr/bin/env python3
"""Run from this directory exactly like the reference's script of the same name:

    cd sr && python 2_transfer_to_lut.py --stages 2 --modes sdy -e ../models/sr_x2sdy

Everything happens in mulut_amd.transfer_to_lut (network evaluated on the GPU)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mulut_amd.transfer_to_lut import main  # noqa: E402

if __name__ == "__main__":
    main()
