"""Per-phase timeline of the BACKWARD chain (esr_rdb_backward, trace): as tools/chain_trace.py.
Usage: python tools/chain_trace_bwd.py [noise 0/1] [H] [W] [rows per wave 4 | 2 | 1]"""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esrganplus_amd import architecture as arch, synth, _lib as L

B, nb = 16, 2
noise = int(sys.argv[1]) if len(sys.argv) > 1 else 0
H = int(sys.argv[2]) if len(sys.argv) > 2 else 128
W = int(sys.argv[3]) if len(sys.argv) > 3 else 128
ROWS = int(sys.argv[4]) if len(sys.argv) > 4 else 4
os.environ['ESR_RDB_ROWS'] = str(ROWS)
dev = torch.device('cuda:0')
net = arch.RRDBNet(3, 3, 64, nb).to(dev).train(bool(noise)).set_precision('fp16')
x = torch.rand(B, 3, H, W, device=dev)
y = net(x)
y.sum().backward()
pool = [v for k, v in net._plans.items() if isinstance(v, list)][0]
tp = pool[0]
ntiles = B * ((H + 4 * ROWS - 1) // (4 * ROWS)) * ((W + 31) // 32)
tr = torch.zeros(ntiles * 64, dtype=torch.int64, device=dev)
arr = tp.bwd.array()
arr[tp.bwd_chain_ops[0]].u.rdb_chain.trace = tr.data_ptr()
for _ in range(3):
    y = net(x); y.sum().backward()
torch.cuda.synchronize()
t = tr.cpu().numpy().reshape(ntiles, 64).astype(np.int64)
names = ['poll1', 'halo1', 'crit1', 'ep1', 'bulk1', 'put', 'poll2', 'halo2', 'crit2', 'ep2', 'bulk2',
         'poll3', 'halo3', 'crit3', '1x1t+ep3', 'bulk3', 'poll4', 'halo4', 'crit4', 'ep4', 'bulk4',
         'poll5', 'halo5', 'crit5', 'tail', 'drain5', 'flag']
d = np.diff(t[:, :len(names) + 1], axis=1) / 100.0
print('backward chain, noise %d: timeline of each tile\'s second block (us), mean over tiles [min..max]' % noise)
tot = 0.0
for i, n in enumerate(names):
    col = d[:, i]
    tot += col.mean()
    print('  %-10s %7.2f  [%6.2f .. %6.2f]' % (n, col.mean(), col.min(), col.max()))
print('  total      %7.2f' % tot)
if os.environ.get('ESR_TRACE_UNITS'):          # build with -DESR_ABL=32 [-DESR_DBG_SEG=<first unit>]: s_memtime stamps of one segment
    u = t[:, 32:64]
    n = int((u[0] != 0).sum())
    du = np.diff(u[:, :n], axis=1).astype(np.float64)
    lab = ['wait+bar', 'first loads'] + sum([['u%d mfma+wait' % k, 'u%d barrier' % k, 'u%d hook' % k] for k in range(20)], [])
    print('unit stamps of the traced segment (shader ticks), mean over tiles [min..max]; %d stamps' % n)
    for i in range(n - 1):
        print('  %-14s %7.0f  [%6.0f .. %6.0f]' % (lab[i], du[:, i].mean(), du[:, i].min(), du[:, i].max()))
    print('  total          %7.0f' % du.sum(axis=1).mean())
