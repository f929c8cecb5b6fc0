#!/usr/bin/env python3
"""Run from this directory exactly like the reference's script of the same name:

    cd sr && python 4_test_lut.py --stages 2 --modes sdy -e ../models/sr_x2sdy

Everything happens in mulut_amd.test_lut (GPU path through libmulut_hip.so)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mulut_amd.test_lut import main  # noqa: E402

if __name__ == "__main__":
    main()
