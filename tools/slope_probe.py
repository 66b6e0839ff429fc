#!/usr/bin/env python
"""Per-K-step slope of the N=32 and N=64 3x3 convs (GPU), full kernel only."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conv_probe import probe   # noqa: E402

for cout in (32, 64):
    r = {cin: probe(cin, cout, 0, 0) for cin in (64, 128, 192)}
    print('cout %d: %s us | per K step %.3f %.3f us | noEpi %s' % (
        cout, '  '.join('%6.2f' % r[c] for c in (64, 128, 192)), (r[128] - r[64]) / 4, (r[192] - r[128]) / 4,
        '  '.join('%6.2f' % probe(c, cout, 0, 1) for c in (64, 128, 192))))
