"""Latency of ONE 128x128 LR tile through RRDBNet x4 (BASELINE configs[0] / test_image/test.py), fp16, per tile height.
Usage (GPU box): python tools/single_tile_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esrganplus_amd import architecture as arch, synth
dev = torch.device('cuda:0')
sd = synth.rrdbnet_state_dict(23, 0)
x = synth.image_batch(1, 1, 3, 128, 128, name='one.x').to(dev)
for rows in ('4', '2', '1', ''):
    if rows:
        os.environ['ESR_RDB_ROWS'] = rows
    else:
        os.environ.pop('ESR_RDB_ROWS', None)
    net = arch.RRDBNet(3, 3, 64, 23).to(dev).eval().set_precision('fp16')
    net.load_state_dict(sd)
    with torch.no_grad():
        for _ in range(5):
            net(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            net(x)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 30
    print('rows per wave %-4s  %.3f ms per tile  %.1f HR-Mpix/s' % (rows or 'auto', dt * 1e3, 512 * 512 / 1e6 / dt))
