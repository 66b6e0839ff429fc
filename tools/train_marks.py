"""Where the MAIN stream of the pipelined ESRGAN+ train step (bench.py --mode train) spends its time, untraced: timed
events recorded on the main stream between the phases of ESRGANPlusStep._step_manual (st._marks), averaged over steps.
The main stream carries the step's critical path (G forward -> netD(fake) forward -> its input-gradient pass -> G
backward -> Adam(G) + pack); what runs on the side streams shows up only as waits.
Usage (GPU box): python tools/train_marks.py [steps] [sync]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esrganplus_amd import architecture as arch, synth, train

dev = torch.device('cuda:0')
NB = 23
netG = arch.RRDBNet(3, 3, 64, NB).to(dev).train().set_precision('fp16')
netD = arch.Discriminator_VGG_128(3, 64).to(dev).train().set_precision('fp16')
netF = arch.VGGFeatureExtractor(34, False, True, dev).to(dev).eval().set_precision('fp16')
netG.load_state_dict(synth.rrdbnet_state_dict(NB, 0, gain=0.5))
netD.load_state_dict(synth.discriminator_state_dict(0))
netF.load_state_dict(synth.vgg19_state_dict(0, 34), strict=False)
st = train.ESRGANPlusStep(netG, netD, netF, loss_scale=1024.0)
lr = synth.image_batch(200, 16, 3, 32, 32, name='bench.lr').to(dev)
hr = synth.image_batch(300, 16, 3, 128, 128, name='bench.hr').to(dev)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
SYNC = len(sys.argv) > 2 and sys.argv[2] == 'sync'      # the logging form: the seven losses read back every step
for _ in range(6):
    st.step(lr, hr, hr, sync_log=False)
st.finish(); torch.cuda.synchronize()
st._marks = []
t0 = time.perf_counter()
for _ in range(steps):
    st.step(lr, hr, hr, sync_log=SYNC)
st.finish(); torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / steps * 1e3
m = st._marks
per = len(m) // steps
names = [n for n, _ in m[:per]]
acc = [0.0] * per
for s in range(1, steps):           # (skip the first step: its start mark follows the warm-up's tail)
    for i in range(per):
        a = m[s * per + i][1]
        b = m[s * per + i + 1][1] if i + 1 < per else (m[(s + 1) * per][1] if s + 1 < steps else None)
        if b is not None:
            acc[i] += a.elapsed_time(b)
print(('logging-form' if SYNC else 'pipelined') + ' step %.3f ms (wall / steps); main-stream phases, ms (mean of %d steps):' % (wall, steps - 1))
tot = 0.0
for i in range(per):
    nxt = names[i + 1] if i + 1 < per else 'next step start'
    v = acc[i] / (steps - 1 if i + 1 < per else steps - 2)
    tot += v
    print('  %-60s %7.3f' % (nxt, v))
print('  %-60s %7.3f' % ('sum', tot))
