"""Pipelined (step(sync_log=False)) against synchronised training loops at the bench size (nb = 23, batch 16): the
weights and BatchNorm buffers after 40 steps must agree bit for bit.  Usage (GPU box): python tools/pipe_check.py"""
import sys, torch
sys.path.insert(0, '.')
from esrganplus_amd import architecture as arch, synth, train
dev = torch.device('cuda:0')
def run(pipelined, steps=40):
    torch.manual_seed(99)
    netG = arch.RRDBNet(3, 3, 64, 23).to(dev).train().set_precision('fp16')
    netD = arch.Discriminator_VGG_128(3, 64).to(dev).train().set_precision('fp16')
    netF = arch.VGGFeatureExtractor(34, False, True, dev).to(dev).eval().set_precision('fp16')
    netG.load_state_dict(synth.rrdbnet_state_dict(23, 0, gain=0.5)); netD.load_state_dict(synth.discriminator_state_dict(0))
    netF.load_state_dict(synth.vgg19_state_dict(0, 34), strict=False)
    st = train.ESRGANPlusStep(netG, netD, netF, loss_scale=1024.0)
    lr = synth.image_batch(200, 16, 3, 32, 32, name='bench.lr').to(dev); hr = synth.image_batch(300, 16, 3, 128, 128, name='bench.hr').to(dev)
    for i in range(steps):
        st.step(lr, hr, sync_log=not pipelined)
    st.finish(); torch.cuda.synchronize()
    out = {'G.' + k: v.detach().clone() for k, v in netG.state_dict().items()}
    out.update({'D.' + k: v.detach().clone() for k, v in netD.state_dict().items()})
    return out
a = run(False); b = run(True); c = run(True)
print('sync vs pipelined differing tensors:', sum(not torch.equal(a[k], b[k]) for k in a), 'of', len(a))
print('pipelined vs pipelined differing tensors:', sum(not torch.equal(c[k], b[k]) for k in a))
