"""Per-step time of the headline forward (batch 16 x 128^2 LR, fp16 eval) over a long run: does the clock settle?
Usage (GPU box): python tools/step_trend.py [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esrganplus_amd import architecture as arch, synth
dev = torch.device('cuda:0')
net = arch.RRDBNet(3, 3, 64, 23).to(dev).eval().set_precision('fp16')
net.load_state_dict(synth.rrdbnet_state_dict(23, 0, gain=0.5))
x = synth.image_batch(1, 16, 3, 128, 128, name='trend.x').to(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
with torch.no_grad():
    net(x); torch.cuda.synchronize()
    ev[0].record()
    for i in range(n):
        net(x); ev[i + 1].record()
torch.cuda.synchronize()
t = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
for a in range(0, n, 10):
    print('steps %3d-%3d: mean %.3f ms  min %.3f  max %.3f' % (a, a + 9, sum(t[a:a + 10]) / 10, min(t[a:a + 10]), max(t[a:a + 10])))
