#!/usr/bin/env python
"""Latency anatomy of the 3x3 conv at training-tile sizes (GPU): batch 16 of 32x32 (32 workgroups)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conv_probe as CP   # noqa: E402

CP.B, CP.H, CP.W = 16, 32, 32
for cout in (32, 64):
    full = {cin: CP.probe(cin, cout, 0, 0) for cin in (64, 128, 192)}
    noepi = {cin: CP.probe(cin, cout, 0, 1) for cin in (64, 128, 192)}
    off = {cin: CP.probe(cin, cout, 0, 7) for cin in (64, 128, 192)}
    print('cout %d | full %s | per K step %.3f us | noEpi %s | launch-only %s' % (
        cout, ' '.join('%5.2f' % full[c] for c in full), (full[192] - full[64]) / 8,
        ' '.join('%5.2f' % noepi[c] for c in noepi), ' '.join('%5.2f' % off[c] for c in off)))
