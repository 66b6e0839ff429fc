"""Run-to-run determinism of the fused chain (fp16): python tools/chain_repeat.py B H W nb reps"""
import sys
import torch
sys.path.insert(0, '.')
from esrganplus_amd import architecture as arch, synth

B, H, W, nb, reps = (int(a) for a in sys.argv[1:6])
dev = torch.device('cuda:0')
net = arch.RRDBNet(3, 3, 64, nb).to(dev).eval().set_precision('fp16')
net.load_state_dict(synth.rrdbnet_state_dict(nb=nb, seed=3))
x = synth.image_batch(5, B, 3, H, W, name='cc.x').to(dev)
with torch.no_grad():
    ref = net(x).clone()
    bad = 0
    for i in range(reps):
        y = net(x)
        if not torch.equal(y, ref):
            bad += 1
            d = (y - ref).abs()
            imgs = [j for j in range(B) if d[j].max() > 0]
            print('run %d differs: max %.3e, images %s, nan %d' % (i, d.max().item(), imgs[:8], int(torch.isnan(y).sum())))
print('B=%d %dx%d nb=%d: %d of %d repeats differ from the first run' % (B, H, W, nb, bad, reps))
