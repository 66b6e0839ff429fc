"""Whole-image inference of a DIV2K-sized LR image (339x510: 352 tiles of 16x32 > 256 CUs): the fused trunk in row
bands (esr_rdb_chain.band_rows) against the per-conv launches.  Usage (GPU box): python tools/big_image_probe.py [H W]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esrganplus_amd import architecture as arch, synth, engine as E
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (339, 510)
dev = torch.device('cuda:0')
os.environ['ESR_RDB_BANDS'] = '1'
sd = synth.rrdbnet_state_dict(23, 0)
x = synth.image_batch(1, 1, 3, H, W, name='big.x').to(dev)
print('band geometry (rows, margin, bands):', E.rdb_band_geometry(H, W), ' one chain launch possible:', E.rdb_chain_ok(1, H, W, False, False))
ys = {}
for prec in ('fp16', 'fp32'):
    for fused in ('1', '0'):
        os.environ['ESR_RDB_FUSED'] = fused
        net = arch.RRDBNet(3, 3, 64, 23).to(dev).eval().set_precision(prec)
        net.load_state_dict(sd)
        with torch.no_grad():
            for _ in range(3):
                y = net(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                y = net(x)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        ys[prec, fused] = y
        fl = 2.0 * 18068160 * H * W
        print('%s %-9s %7.2f ms  %6.1f HR-Mpix/s  %6.1f TFLOP/s' % (prec, 'banded' if fused == '1' else 'per-conv', dt * 1e3, 16 * H * W / 1e6 / dt, fl / dt / 1e12))
    print(prec, 'banded == per-conv:', torch.equal(ys[prec, '1'], ys[prec, '0']), ' max|diff| %.2e' % (ys[prec, '1'] - ys[prec, '0']).abs().max().item())
