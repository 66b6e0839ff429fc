#!/usr/bin/env python
"""Per-op timing of the two launch lists of the fwd+bwd probe (train-mode RRDBNet, 16 x 128^2, fp16): HIP events per op."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from esrganplus_amd import architecture as arch, synth, engine as E, functional as Fn, _lib as L
dev = torch.device('cuda:0')
net = arch.RRDBNet(3, 3, 64, 23).to(dev).train().set_precision('fp16')
net.load_state_dict(synth.rrdbnet_state_dict(23, 0, gain=0.5))
lr = synth.image_batch(300, 16, 3, 128, 128, name='bench.fb.lr').to(dev)
hr = synth.image_batch(301, 16, 3, 512, 512, name='bench.fb.hr').to(dev)
for _ in range(2):
    for q in net.parameters():
        q.grad = None
    (F.l1_loss(net(lr), hr) * 1024).backward()
torch.cuda.synchronize()
tp = next(t for k, pool in net._plans.items() if isinstance(k, tuple) and k and k[0] == 'train' for t in pool)
st = E.current_stream()
out = torch.empty(tp.fwd.out_shape, dtype=torch.float32, device=dev)
gy = torch.full(tp.fwd.out_shape, 1024.0 / out.numel(), dtype=torch.float32, device=dev)
KIND = {L.OP_CONV: 'conv', L.OP_WGRAD: 'wgrad', L.OP_LAYOUT: 'layout', L.OP_RDB_CHAIN: 'chain', L.OP_RDB_CHAIN_BWD: 'chain_bwd',
        L.OP_RDB_WGRAD: 'rdb_wgrad', L.OP_PACK_BATCH: 'pack', L.OP_UNPERMUTE: 'unpermute', L.OP_PACK: 'pack1'}
accf = accb = None
for rep in range(4):
    tp.fwd.run(lr, out, st, 1234, None)
    mf = tp.fwd.ops.run_timed(st)
    Fn._train_backward(tp, gy, st, True, False, 1234, False)
    mb = tp.bwd.run_timed(st)
    if rep:
        accf = mf if accf is None else [a + b for a, b in zip(accf, mf)]
        accb = mb if accb is None else [a + b for a, b in zip(accb, mb)]
for name, ops, acc in (('forward', tp.fwd.ops.ops, accf), ('backward', tp.bwd.ops, accb)):
    print(name, 'total %.3f ms' % (sum(acc) / 3))
    for o, t in zip(ops, acc):
        d = ''
        if o.kind == L.OP_CONV:
            c = o.u.conv
            d = 'ks %d ups %d %dx%d cin_g %d cout_b %d' % (c.ks, c.upsample, c.H, c.W, c.cin_groups, c.cout_blocks)
        elif o.kind == L.OP_WGRAD:
            w = o.u.wgrad
            d = 'ks %d ups %d %dx%d cout %d cin %d' % (w.ks, w.upsample, w.H, w.W, w.cout, w.cin)
        print('   %-10s %7.3f ms  %s' % (KIND.get(o.kind, str(o.kind)), t / 3, d))
