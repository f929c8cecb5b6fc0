"""mulut_amd -- MI355X-native MuLUT LUT inference (drop-in for the reference's sr/4_test_lut.py path).

Host side in Python (mirroring the reference's callables and CLI), device side in hand-written HIP
behind the C ABI of include/mulut.h.  There is no CPU fallback: every compute entry point goes
through libmulut_hip.so and raises if it (or a GPU) is missing.
"""
from .lut_io import lut_file_name, load_lut_dict, synthetic_lut  # noqa: F401
from .engine import MuLUTEngine, MuLUTError  # noqa: F401
from .interp import FourSimplexInterpFaster  # noqa: F401
from . import finetune_lut  # noqa: F401

__version__ = "0.1.0"
