#!/usr/bin/env python
"""Micro-probe of the fused conv kernel (GPU): back-to-back launches of ONE conv shape, timed as a
batch — isolates prologue/epilogue/K-loop costs with the measurement-only debug flags."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esrganplus_amd import engine as E, _lib as L

dev = torch.device('cuda:0')
B, H, W = 16, 128, 128
REP = 40


def probe(cin, cout, res, flags, prec='fp16', ups=0, ks=3):
    w = torch.randn(cout, cin, ks, ks, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    wp = E.WeightPack([('c', w, b)], prec, dev)
    st = E.current_stream()
    wp.ensure(st, force=True)
    src = E.G32(B, max(192, cin), H // (2 if ups else 1), W // (2 if ups else 1), prec, dev)
    src.t.normal_()
    dst = E.G32(B, max(192, cout), H, W, prec, dev)
    c = E._conv(wp.esr_dtype, B, H, W, src.view(0), cin, dst.view(0, cout), wp.entries['c'], L.ACT_LRELU, upsample=ups)
    if res:
        c.res1, c.alpha = dst.view(64, cout), 0.2
    c.debug_flags = flags
    ops = L.OpList()
    for _ in range(REP):
        ops.add_conv(c)
    ops.run(st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        ops.run(st)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / REP * 1e3)
    return best


if __name__ == '__main__':
    print('cin cout res | full   noEpi  noMMA  noDMA  noMMA+noDMA  onlyLaunch(all off)  nt-store  wt-store [us per launch, back-to-back]')
    for cin, cout, res in ((16, 32, 0), (64, 32, 0), (128, 32, 0), (160, 32, 1), (192, 64, 1), (64, 64, 0)):
        r = [probe(cin, cout, res, f) for f in (0, 1, 2, 4, 6, 7, 8, 16)]
        mac = B * H * W * cin * cout * 9
        print('%3d %3d %d | %s   | %.0f TF/s full' % (cin, cout, res, '  '.join('%5.1f' % v for v in r), 2 * mac / r[0] / 1e6))
