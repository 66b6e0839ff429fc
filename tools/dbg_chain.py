import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from esrganplus_amd import synth, block as B
dev = torch.device('cuda:0')
def make(train=True):
    torch.manual_seed(3)
    m = B.RRDB(64)
    with torch.no_grad():
        for p in m.parameters(): p.mul_(1.5)
    return m.to(dev).train(train)
shape = (2, 64, 24, 40)
x = synth.normal_like(5, 'tc.x', shape).to(dev); gy = synth.normal_like(6, 'tc.gy', shape).to(dev)
def run(chain, prec='fp16', train=True):
    os.environ['ESR_RDB_TRAIN_CHAIN'] = '1' if chain else '0'
    m = make(train).set_precision(prec)
    xr = x.clone().requires_grad_(True)
    torch.manual_seed(77)
    y = m(xr); (y * gy).sum().backward(); torch.cuda.synchronize()
    return y.detach(), xr.grad, {k: p.grad.clone() for k, p in m.named_parameters()}
rel = lambda a, b: (a.double() - b.double()).norm().item() / (b.double().norm().item() + 1e-30)
for train in (True, False):
    a = run(False, train=train); b = run(True, train=train); c = run(False, 'fp32', train=train)
    print('train', train, 'y', rel(b[0], a[0]), 'gx', rel(b[1], a[1]))
    for k in a[2]:
        print('  %-22s chain/perconv %.2e   chain/fp32 %.2e   perconv/fp32 %.2e   |g| %.3e' % (k, rel(b[2][k], a[2][k]), rel(b[2][k], c[2][k]), rel(a[2][k], c[2][k]), c[2][k].norm().item()))
