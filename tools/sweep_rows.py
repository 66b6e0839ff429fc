"""Inference forward (nb = 23, fp16, 128x128 LR) per batch size and chain tile height (ESR_RDB_ROWS = 4 / 2 / 1): the data
behind rdb_fused.hip: rows_per_wave().  Usage (GPU box): python tools/sweep_rows.py"""
import os, sys, time
sys.path.insert(0, '.')
import torch
from esrganplus_amd import architecture as arch, synth
dev = torch.device('cuda:0')
sd = synth.rrdbnet_state_dict(23, 0)
for B in (2, 3, 4, 6, 8):
    x = synth.image_batch(1, B, 3, 128, 128, name='one.x').to(dev)
    res = []
    for rows in ('4', '2', '1'):
        os.environ['ESR_RDB_ROWS'] = rows
        net = arch.RRDBNet(3, 3, 64, 23).to(dev).eval().set_precision('fp16')
        net.load_state_dict(sd)
        with torch.no_grad():
            for _ in range(4):
                net(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                net(x)
            torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 20 * 1e3)
    print('B %d (16-row tiles %3d): rows 4/2/1 = %.3f / %.3f / %.3f ms' % (B, B * 32, *res))
