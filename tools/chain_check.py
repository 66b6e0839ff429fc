"""Fused chain vs per-conv path on one shape (fp16): python tools/chain_check.py B H W [nb]"""
import os
import sys
import torch
sys.path.insert(0, '.')
from esrganplus_amd import architecture as arch, synth

B, H, W = (int(a) for a in sys.argv[1:4])
nb = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device('cuda:0')
sd = synth.rrdbnet_state_dict(nb=nb, seed=3)
x = synth.image_batch(5, B, 3, H, W, name='cc.x').to(dev)
ys = {}
for fused in ('1', '0'):
    os.environ['ESR_RDB_FUSED'] = fused
    net = arch.RRDBNet(3, 3, 64, nb).to(dev).eval().set_precision('fp16')
    net.load_state_dict(sd)
    with torch.no_grad():
        ys[fused] = net(x).cpu()
d = (ys['1'] - ys['0']).abs()
per_img = d.flatten(1).max(1).values
print('B=%d %dx%d nb=%d tiles=%d: max|fused - per-conv| = %.3e; per image: %s' % (
    B, H, W, nb, B * ((H + 15) // 16) * ((W + 31) // 32), d.max().item(), ' '.join('%.1e' % v for v in per_img.tolist())))
# where the differences sit relative to the 16x32 LR tiles (x4 in HR)
e = d.max(1).values            # B x 4H x 4W
rows = e.amax(dim=(0, 2)).view(-1, 64).amax(0)       # by HR row inside a tile
cols = e.amax(dim=(0, 1)).view(-1, 128).amax(0)      # by HR column inside a tile
print('by LR row in tile :', ' '.join('%.0e' % rows[4 * i:4 * i + 4].max().item() for i in range(16)))
print('by LR col in tile :', ' '.join('%.0e' % cols[4 * i:4 * i + 4].max().item() for i in range(32)))
