"""Where one ESRGAN+ train step (bench.py --mode train) spends its time: GPU time per phase from events on the
current stream and the host time at which each phase was ENQUEUED (no syncs in between), so that host-bound
stretches (GPU idle, waiting for launches) show up as phases whose enqueue finishes after the GPU could have.
Usage (GPU box): python tools/train_phases.py [steps]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from esrganplus_amd import architecture as arch, synth, train, dp as DP, losses as LS

dev = torch.device('cuda:0')
NB = 23
netG = arch.RRDBNet(3, 3, 64, NB).to(dev).train().set_precision('fp16')
netD = arch.Discriminator_VGG_128(3, 64).to(dev).train().set_precision('fp16')
netF = arch.VGGFeatureExtractor(34, False, True, dev).to(dev).eval().set_precision('fp16')
netG.load_state_dict(synth.rrdbnet_state_dict(NB, 0, gain=0.5))
netD.load_state_dict(synth.discriminator_state_dict(0))
netF.load_state_dict(synth.vgg19_state_dict(0, 34), strict=False)
st = train.ESRGANPlusStep(netG, netD, netF, loss_scale=1024.0)
NBATCH = int(os.environ.get('TP_BATCH', '16'))
lr = synth.image_batch(200, NBATCH, 3, 32, 32, name='bench.lr').to(dev)
hr = synth.image_batch(300, NBATCH, 3, 128, 128, name='bench.hr').to(dev)
SCALE = torch.full((), 1024.0, device=dev)

def step(marks):
    def mark(name):
        e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e, time.perf_counter()))
    mark('start')
    for p in netD.parameters(): p.requires_grad = False
    st.optimizer_G.zero_grad(set_to_none=True)
    fake = netG(lr); mark('G fwd')
    l_pix = LS.l1_loss(fake, hr, 1e-2)
    ff, rf = netF.forward_pair(fake, hr); l_fea = LS.l1_loss(ff, rf, 1.0); mark('VGG fwd x2')
    pg, pr = netD.forward_pair(fake, hr)
    l_gan, _ = LS.ragan_loss(pr, pg, False, True, 5e-3); mark('D fwd x2 + losses')
    main = torch.cuda.current_stream(); side = st._side(dev)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        for p in netD.parameters(): p.requires_grad = True
        st.optimizer_D.zero_grad(set_to_none=True)
        with netD.weights_unchanged():
            pr, pf = netD.forward_pair(hr, fake.detach())
        ld, _ = LS.ragan_loss(pr, pf, True, False, 1.0)
        torch.autograd.backward([ld], [SCALE])
    mark('D step enqueued on side stream')
    torch.autograd.backward([l_pix, l_fea, l_gan], [SCALE, SCALE, SCALE]); main.wait_stream(side); mark('backward (D, VGG, G) || D step')
    st.optimizer_G.step(grad_scale=1 / 1024.0); st.optimizer_D.step(grad_scale=1 / 1024.0); mark('Adam x2')

for _ in range(3): step([])
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
acc = {}
t0 = time.perf_counter()
allm = []
for _ in range(n):
    m = []; step(m); allm.append(m)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n * 1e3
for m in allm:
    for (a, ea, ta), (b, eb, tb) in zip(m, m[1:]):
        g = ea.elapsed_time(eb); acc.setdefault(b, [0, 0]); acc[b][0] += g / n; acc[b][1] += (tb - ta) * 1e3 / n
print('wall %.2f ms/step' % wall)
print('%-28s %8s %8s' % ('phase', 'gpu ms', 'host ms'))
for k, (g, h) in acc.items(): print('%-28s %8.2f %8.2f' % (k, g, h))
print('%-28s %8.2f %8.2f' % ('sum', sum(v[0] for v in acc.values()), sum(v[1] for v in acc.values())))
