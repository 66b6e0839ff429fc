#!/usr/bin/env python
"""Host-side profile of the ESRGAN+ train step with the autograd engine on the calling thread (so that cProfile sees
the backward nodes too): where the enqueue time of a step goes.  Usage (GPU box): python tools/host_profile.py [n]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esrganplus_amd import architecture as arch, synth, train

dev = torch.device('cuda:0')
netG = arch.RRDBNet(3, 3, 64, 23).to(dev).train().set_precision('fp16')
netD = arch.Discriminator_VGG_128(3, 64).to(dev).train().set_precision('fp16')
netF = arch.VGGFeatureExtractor(34, False, True, dev).to(dev).eval().set_precision('fp16')
netG.load_state_dict(synth.rrdbnet_state_dict(23, 0, gain=0.5))
netD.load_state_dict(synth.discriminator_state_dict(0))
netF.load_state_dict(synth.vgg19_state_dict(0, 34), strict=False)
st = train.ESRGANPlusStep(netG, netD, netF, loss_scale=1024.0)
lr = synth.image_batch(200, 16, 3, 32, 32, name='bench.lr').to(dev)
hr = synth.image_batch(300, 16, 3, 128, 128, name='bench.hr').to(dev)
for _ in range(3):
    st.step(lr, hr, sync_log=False)
torch.cuda.synchronize()
for mt in (True, False):
    torch.autograd.set_multithreading_enabled(mt)
    for _ in range(2):
        st.step(lr, hr, sync_log=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        st.step(lr, hr, sync_log=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('autograd multithreading %s: enqueue %.2f ms/step, until GPU idle %.2f ms/step'
          % (mt, (t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
import cProfile
import pstats
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    st.step(lr, hr, sync_log=False)
pr.disable()
torch.cuda.synchronize()
print('--- %d steps, by tottime' % n)
pstats.Stats(pr).sort_stats('tottime').print_stats(40)
print('--- by cumulative')
pstats.Stats(pr).sort_stats('cumulative').print_stats(70)
