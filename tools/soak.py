#!/usr/bin/env python
"""Soak test (GPU): many train steps / mixed-shape generator steps / forwards; checks finiteness, that
the loss moves, and that device memory does not grow (plan pools, side stream, graph handles)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from esrganplus_amd import architecture as arch, synth, train
from esrganplus_amd.optim import FusedAdam

dev = torch.device('cuda:0')
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
netG = arch.RRDBNet(3, 3, 64, 23).to(dev).train().set_precision('fp16')
netD = arch.Discriminator_VGG_128(3, 64).to(dev).train().set_precision('fp16')
netF = arch.VGGFeatureExtractor(34, False, True, dev).to(dev).eval().set_precision('fp16')
netG.load_state_dict(synth.rrdbnet_state_dict(23, 0, gain=0.5))
netD.load_state_dict(synth.discriminator_state_dict(0))
netF.load_state_dict(synth.vgg19_state_dict(0, 34), strict=False)
st = train.ESRGANPlusStep(netG, netD, netF, loss_scale=1024.0)
lr = synth.image_batch(200, 16, 3, 32, 32, name='bench.lr').to(dev)
hr = synth.image_batch(300, 16, 3, 128, 128, name='bench.hr').to(dev)
for _ in range(5):
    st.step(lr, hr, sync_log=False)
torch.cuda.synchronize()
m0 = torch.cuda.memory_allocated()
t0 = time.perf_counter()
first = None
for i in range(steps):
    log = st.step(lr, hr, sync_log=(i % 50 == 0))
    if i % 50 == 0:
        assert all(v == v and abs(v) < 1e6 for v in log.values()), log
        first = first or dict(log)
        print('step %4d  l_g_pix %.5f  l_g_fea %.5f  l_d_real %.4f  mem %.1f MB' % (
            i, log['l_g_pix'], log['l_g_fea'], log['l_d_real'], torch.cuda.memory_allocated() / 2**20), flush=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
m1 = torch.cuda.memory_allocated()
print('train: %d steps, %.2f ms/step, memory %+.1f MB' % (steps, dt / steps * 1e3, (m1 - m0) / 2**20))
assert m1 - m0 < 64 * 2**20, 'device memory grew'
last = {k: float(v) for k, v in st.step(lr, hr, sync_log=True).items()}
assert last['l_g_pix'] < first['l_g_pix'], (first, last)      # the generator is learning the fixed batch

# mixed-shape generator steps + eval forwards interleaved (plan pools for several shapes)
opt = FusedAdam(netG.parameters(), lr=1e-4)
shapes = [(2, 48, 40), (1, 64, 64), (3, 24, 56)]
xs = [(torch.rand(b, 3, h, w, device=dev), torch.rand(b, 3, 4 * h, 4 * w, device=dev)) for b, h, w in shapes]
m0 = None
for i in range(max(steps // 3, 30)):
    x, y = xs[i % 3]
    opt.zero_grad(set_to_none=True)
    loss = F.l1_loss(netG(x), y)
    (loss * 1024).backward()
    opt.step(grad_scale=1 / 1024)
    if i % 3 == 2:
        netG.eval()
        with torch.no_grad():
            o = netG(xs[0][0])
        netG.train()
        assert torch.isfinite(o).all()
    if i == 8:
        torch.cuda.synchronize()
        m0 = torch.cuda.memory_allocated()
torch.cuda.synchronize()
assert torch.isfinite(loss)
print('mixed shapes ok, memory %+.1f MB' % ((torch.cuda.memory_allocated() - m0) / 2**20))
assert torch.cuda.memory_allocated() - m0 < 64 * 2**20
print('SOAK OK')
