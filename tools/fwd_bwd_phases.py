"""fwd+bwd at the bench shape split into forward / backward (HIP events), noise on and off.
Usage: python tools/fwd_bwd_phases.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from esrganplus_amd import architecture as arch, synth
dev = torch.device('cuda:0')
B, LR = 16, 128
lr = synth.image_batch(300, B, 3, LR, LR, name='bench.fb.lr').to(dev)
hr = synth.image_batch(301, B, 3, 4 * LR, 4 * LR, name='bench.fb.hr').to(dev)
for train in (True, False):
    net = arch.RRDBNet(3, 3, 64, 23).to(dev).train(train).set_precision('fp16')
    net.load_state_dict(synth.rrdbnet_state_dict(23, 0, gain=0.5))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for it in range(6):
        for q in net.parameters():
            q.grad = None
        ev[0].record()
        loss = F.l1_loss(net(lr), hr)
        ev[1].record()
        (loss * 1024.0).backward()
        ev[2].record()
        torch.cuda.synchronize()
        if it >= 2:
            tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
    print('noise %s: forward %.2f ms, backward %.2f ms, total %.2f ms' % ('on' if train else 'off', tf / 4, tb / 4, (tf + tb) / 4))
    del net
    torch.cuda.empty_cache()
