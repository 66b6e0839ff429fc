#!/usr/bin/env python
"""Which Python lines issue the torch glue launches (copy_ / fill_ / zero_ / cat / mul / add / clone) of one ESRGAN+ train
step?  torch.profiler with stacks; prints, per call site inside this repo, the launches per step."""
import collections
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
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
torch.autograd.set_multithreading_enabled(False)
N = 3
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    for _ in range(N):
        st.step(lr, hr, sync_log=False)
torch.cuda.synchronize()
agg = collections.Counter()
for ev in prof.events():
    if ev.name in ('aten::copy_', 'aten::fill_', 'aten::zero_', 'aten::cat', 'aten::mul', 'aten::add', 'aten::add_', 'aten::clone',
                   'aten::div', 'aten::zeros', 'aten::full', 'aten::empty_like', 'aten::sum', 'aten::_foreach_copy_'):
        site = next((s for s in (ev.stack or []) if 'esrganplus_amd' in s or 'bench.py' in s), None) or ((ev.stack or ['?'])[0])
        agg[(ev.name, site)] += 1
for (name, site), n in sorted(agg.items(), key=lambda kv: -kv[1]):
    print('%5.1f /step  %-16s %s' % (n / N, name, site[-110:]))
