"""Top-level ``architecture`` module for scripts that import the reference's modules by bare name
(test_image/test.py:7 ``import architecture as arch``; test_image/architecture.py:4 ``import block as B``).
Put THIS directory on ``sys.path`` instead of /root/reference/test_image and the script runs unchanged on the
HIP path:  ``arch.RRDB_Net(3, 3, 64, 23, gc=32, upscale=4, norm_type=None, act_type='leakyrelu', mode='CNA',
res_scale=1, upsample_mode='upconv')`` (test.py:15-17), ``load_state_dict(..., strict=True)``, ``.eval()``,
``.to(device)``, ``model(img_LR)``.  The classes are the drop-in ones of ``esrganplus_amd.architecture``."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from esrganplus_amd.architecture import *  # noqa: E402,F401,F403
from esrganplus_amd.architecture import (RRDBNet, RRDB_Net, Discriminator_VGG_128,  # noqa: E402,F401
                                         VGGFeatureExtractor)
