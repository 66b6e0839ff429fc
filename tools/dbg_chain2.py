import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from esrganplus_amd import synth, block as B
dev = torch.device('cuda:0')
torch.manual_seed(3)
m = B.ResidualDenseBlock_5C(64).to(dev).eval().set_precision('fp16')
x = synth.normal_like(5, 'tc.x', (2, 64, 16, 32)).to(dev)
with torch.no_grad():
    y0 = m(x)                      # DIR 0 chain
xr = x.clone().requires_grad_(True)
y1 = m(xr)                         # DIR 1 chain (no noise)
os.environ['ESR_RDB_TRAIN_CHAIN'] = '0'
m2 = B.ResidualDenseBlock_5C(64).to(dev).eval().set_precision('fp16')
m2.load_state_dict(m.state_dict())
y2 = m2(x.clone().requires_grad_(True))
rel = lambda a, b: ((a - b).norm() / b.norm()).item()
print('DIR1 vs DIR0', rel(y1.detach(), y0), ' perconv-train vs DIR0', rel(y2.detach(), y0))
d = (y1.detach() - y0).abs()
print('max abs', d.max().item(), 'where', (d > 1e-2).nonzero()[:10].tolist())
print('per-row err', d.amax(dim=(0, 1, 3)).tolist())
print('per-col err', d.amax(dim=(0, 1, 2)).tolist())
print('per-ch err', d.amax(dim=(0, 2, 3)).tolist())
