#!/usr/bin/env python
"""Per-op timing table of one RRDBNet forward plan (debug/perf tool; GPU only)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esrganplus_amd import architecture as arch, synth, engine as E, _lib as L
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
LR = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device('cuda:0')
net = arch.RRDBNet(3, 3, 64, 23).to(dev).eval().set_precision(sys.argv[3] if len(sys.argv) > 3 else 'fp16')
net.load_state_dict(synth.rrdbnet_state_dict(23, 0))
x = synth.image_batch(1, B, 3, LR, LR).to(dev)
with torch.no_grad():
    for _ in range(3):
        net(x)
    torch.cuda.synchronize()
    plan = next(iter(net._plans.values()))
    desc = bench.describe_plan(net, plan)
    acc = None
    reps = 5
    for _ in range(reps):
        ms = plan.ops.run_timed(E.current_stream())
        acc = ms if acc is None else [a + b for a, b in zip(acc, ms)]
ms = [a / reps for a in acc]
print('total %.3f ms' % sum(ms))
shown = list(range(0, 18)) + list(range(len(ms) - 6, len(ms)))
for i in shown:
    o = plan.ops.ops[i]
    name, fl = desc[i]
    if o.kind == L.OP_CONV:
        c = o.u.conv
        print('%3d %-20s %dx%dx%d cin_groups=%2d cbk=%d  %8.1f us  %7.1f TF/s' % (i, name, c.B, c.H, c.W, c.cin_groups, c.cout_blocks, ms[i] * 1e3, fl / (ms[i] * 1e-3) / 1e12))
    else:
        print('%3d %-20s %8.1f us' % (i, name, ms[i] * 1e3))
