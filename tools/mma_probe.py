#!/usr/bin/env python
"""K-loop probe (GPU): per-K-step cost of the 3x3 conv with DMA / LDS reads / MFMAs toggled by the
measurement-only debug flags (4 = no DMA, 32 = no LDS fragment reads, 2 = no reads and no MFMAs).
Flag 32 needs a library built with -DESR_PROBES=1 (ESR_LIB_PATH selects it)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conv_probe import probe   # noqa: E402  (prints its own table on import)

print('cin | full  noDMA  noDMA+noRD  noRD  noRD+noMMA(=2)  noDMA+noRD+noMMA(=6)   [us/launch]; slope per 16-cin K step')
rows = {}
for cin in (64, 128, 192):
    rows[cin] = [probe(cin, 32, 0, f) for f in (0, 4, 36, 32, 2, 6)]
    print('%3d | %s' % (cin, '  '.join('%6.2f' % v for v in rows[cin])))
print('us per K step (128->192): ' + '  '.join('%6.3f' % ((b - a) / 4) for a, b in zip(rows[128], rows[192])))
print('us per K step (64->128):  ' + '  '.join('%6.3f' % ((b - a) / 4) for a, b in zip(rows[64], rows[128])))
