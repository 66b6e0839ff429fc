#!/usr/bin/env python
"""Short-K / wide-N 3x3 convs (the dense block's dgrad shapes) under both >=4-cout-block kernels:
ESR_WIDE_MIN_CIN_GROUPS=0 forces the 8-wave register-weight kernel, =99 the 4-wave LDS-weight one."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conv_probe as CP   # noqa: E402

for cin, cout in ((32, 128), (32, 160), (64, 192), (64, 128), (128, 128), (128, 256), (256, 256)):
    t = CP.probe(cin, cout, 0, 0)
    print('cin %3d cout %3d: %7.1f us  %6.0f TF/s' % (cin, cout, t, 2 * CP.B * CP.H * CP.W * cin * cout * 9 / t / 1e6))
