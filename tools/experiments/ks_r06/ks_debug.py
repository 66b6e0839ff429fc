import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esrganplus_amd import synth, block as B
dev = torch.device('cuda:0')
os.environ['ESR_RDB_ROWS'] = '1'
shape = (1, 64, 8, 32)
def make():
    torch.manual_seed(3)
    return B.ResidualDenseBlock_5C(64).to(dev).eval().set_precision('fp16')
x = synth.normal_like(15, 'th.x', shape).to(dev)
ys = {}
for ks in ('0', '1'):
    os.environ['ESR_RDB_KS'] = ks
    m = make()
    xr = x.clone().requires_grad_(True)
    y = m(xr)
    ys[ks] = y.detach().float().cpu()
    # the saved slices of the training plan: x1..x4
    tp = [t for k, pool in m._plans.items() if isinstance(pool, list) for t in pool][0]
    print(ks, [a for a in dir(tp) if not a.startswith('_')][:60])
d = (ys['1'] - ys['0']).abs()
print('max', d.max().item(), 'ref max', ys['0'].abs().max().item())
print('by row   :', ' '.join('%.1e' % v for v in d.amax(dim=(0, 1, 3)).tolist()))
print('by col   :', ' '.join('%.0e' % v for v in d.amax(dim=(0, 1, 2)).tolist()))
print('by ch/8  :', ' '.join('%.1e' % v for v in d.amax(dim=(0, 2, 3)).view(8, 8).amax(1).tolist()))
