#!/usr/bin/env python3
"""The fork's working variant of the test script (reference sr/5_test_lut.py): the same CLI as 4_test_lut.py when
called with arguments, plus the single-image helpers (main_gui, process_single_image, ...) for import.

    cd sr && python 5_test_lut.py --stages 2 --modes sdy -e ../models/sr_x2sdy
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mulut_amd.single import (create_simple_options, load_luts, main_gui, process_single_image,  # noqa: E402,F401
                              process_single_image_with_gt, test_single_image_direct)
from mulut_amd.test_lut import main  # noqa: E402

if __name__ == "__main__":
    main()
