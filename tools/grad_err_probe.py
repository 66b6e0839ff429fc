"""Per-tensor relative error of the fp32 backward vs a float64 CPU restatement (oracle), next to the error of the
float32 CPU restatement itself.  Usage: python tools/grad_err_probe.py [nb] [size]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esrganplus_amd import architecture as arch, synth
from oracle import ref_torch as RT
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device('cuda:0')
sd = synth.rrdbnet_state_dict(nb=nb, seed=0, gain=0.5)
x = synth.image_batch(41, 1, 3, n, n, name='probe.x')
gy = synth.normal_like(41, 'probe.gy', (1, 3, 4 * n, 4 * n)) / (3 * 16 * n * n)
def cpu(dt):
    sdr = {k: v.to(dt).clone().requires_grad_(True) for k, v in sd.items()}
    y = RT.rrdbnet_forward(x.to(dt), sdr, nb, None, 'codes')
    (y * gy.to(dt)).sum().backward()
    return y.detach(), {k: v.grad for k, v in sdr.items()}
y64, g64 = cpu(torch.float64)
y32, g32 = cpu(torch.float32)
net = arch.RRDBNet(3, 3, 64, nb).to(dev).eval().set_precision('fp32')
net.load_state_dict(sd)
y = net(x.to(dev))
(y * gy.to(dev)).sum().backward()
print('y: gpu %.2e cpu32 %.2e' % ((y.detach().cpu().double() - y64).abs().max() / y64.abs().max(), (y32.double() - y64).abs().max() / y64.abs().max()))
for k, p in net.named_parameters():
    r = g64[k]
    e1 = ((p.grad.cpu().double() - r).abs().max() / r.abs().max()).item()
    e2 = ((g32[k].double() - r).abs().max() / r.abs().max()).item()
    if 'RDB' not in k or 'RDB1.conv1.' in k or 'conv5' in k:
        print('%-40s gpu %.2e  cpu32 %.2e' % (k, e1, e2))
