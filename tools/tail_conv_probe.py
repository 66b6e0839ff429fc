#!/usr/bin/env python
"""Where the time of the 512x512 tail convs goes (GPU): HR_conv0 (64 -> 64) and HR_conv1 (64 -> 3) at 16 x 512 x 512 with the
measurement-only debug flags of esr_conv (1 = no epilogue, 2 = no fragment reads / MFMAs, 4 = no activation DMA)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conv_probe as P   # noqa: E402

P.B, P.H, P.W, P.REP = 16, 512, 512, 6
print('cin cout | full   noEpi  noMMA  noDMA  noMMA+noDMA  noEpi+noMMA  all-off  nt-store  flavour2  contiguous-store(wrong layout)   [us per launch]')
for cin, cout in ((64, 64), (64, 3), (16, 64)):
    r = [P.probe(cin, cout, 0, f) for f in (0, 1, 2, 4, 6, 3, 7, 8, 16, 24)]
    print('%3d %3d | %s' % (cin, cout, '  '.join('%6.1f' % v for v in r)))
