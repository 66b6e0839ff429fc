"""K-split build of the 4-row training chains (ESR_RDB_KS) against the one-row build: outputs / gradients of stand-alone
blocks and a small RRDBNet, and chain times at the training-crop shape.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esrganplus_amd import synth, block as B, architecture as arch

dev = torch.device('cuda:0')
os.environ['ESR_RDB_ROWS'] = '1'


def rel(a, b):
    return (a.double() - b.double()).norm().item() / (b.double().norm().item() + 1e-30)


def grads(make, x, gy, ks):
    os.environ['ESR_RDB_KS'] = ks
    m = make()
    xr = x.clone().requires_grad_(True)
    torch.manual_seed(77)
    y = m(xr)
    (y * gy).sum().backward()
    torch.cuda.synchronize()
    return y.detach(), xr.grad, {k: p.grad.clone() for k, p in m.named_parameters()}


for kind, shape in [('rdb', (2, 64, 16, 32)), ('rrdb', (2, 64, 24, 40)), ('rrdb_ti', (1, 64, 33, 31)), ('rdb', (3, 64, 7, 70))]:
    def make():
        torch.manual_seed(3)
        m = B.ResidualDenseBlock_5C(64) if kind == 'rdb' else B.RRDB(64, extra_noise=(kind == 'rrdb_ti'))
        return m.to(dev).train().set_precision('fp16')
    x = synth.normal_like(15, 'th.x', shape).to(dev)
    gy = synth.normal_like(16, 'th.gy', shape).to(dev)
    y0, gx0, g0 = grads(make, x, gy, '0')
    y1, gx1, g1 = grads(make, x, gy, '1')
    y2, gx2, g2 = grads(make, x, gy, '1')
    print(kind, shape, 'y %.2e gx %.2e worst param %.2e | run-to-run equal: %s' % (
        rel(y1, y0), rel(gx1, gx0), max(rel(g1[k], g0[k]) for k in g0),
        torch.equal(y1, y2) and torch.equal(gx1, gx2) and all(torch.equal(g1[k], g2[k]) for k in g1)), flush=True)

# timing at the training-crop shape: nb = 23 generator forward + backward
sd = synth.rrdbnet_state_dict(nb=23, seed=8, gain=0.7)
x = synth.image_batch(8, 16, 3, 32, 32, name='ks.x').to(dev)
gy = synth.normal_like(9, 'ks.gy', (16, 3, 128, 128)).to(dev)
for ks in ('0', '1', '0', '1'):
    os.environ['ESR_RDB_KS'] = ks
    net = arch.RRDBNet(3, 3, 64, 23).to(dev).train().set_precision('fp16')
    net.load_state_dict(sd)
    for it in range(8):
        if it == 3:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        net.zero_grad(set_to_none=True)
        y = net(x)
        (y * gy).sum().backward()
    torch.cuda.synchronize()
    print('KS=%s: generator fwd+bwd %.3f ms per step' % (ks, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
