"""Top-level ``block`` module (test_image/architecture.py:4 ``import block as B``; see architecture.py here)."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from esrganplus_amd.block import *  # noqa: E402,F401,F403
from esrganplus_amd.block import (ResidualDenseBlock_5C, RRDB, GaussianNoise, conv_block, act, norm, pad,  # noqa: E402,F401
                                  sequential, ShortcutBlock, upconv_blcok)
