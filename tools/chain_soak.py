"""Run-to-run determinism soak of the fused chains (fp16): the inference chain and the training forward + backward
(+ rdb_wgrad) at the shapes that pick the 16-, 8- and 4-row builds, repeated and compared bit for bit with the first
run — a missed wait or hand-off shows up as a rare difference.  Usage (GPU box): python tools/chain_soak.py [seconds per case]"""
import sys, time
import torch
sys.path.insert(0, '.')
from esrganplus_amd import architecture as arch, synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
dev = torch.device('cuda:0')
nb = 4
sd = synth.rrdbnet_state_dict(nb=nb, seed=3)
bad_total = 0
for (B, H, W) in [(16, 32, 32), (1, 128, 128), (3, 128, 128), (16, 128, 128), (2, 33, 70), (5, 96, 40)]:
    net = arch.RRDBNet(3, 3, 64, nb).to(dev).eval().set_precision('fp16')
    net.load_state_dict(sd)
    x = synth.image_batch(5, B, 3, H, W, name='soak.x').to(dev)
    with torch.no_grad():
        ref = net(x).clone()
        n = bad = 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < budget:
            for _ in range(20):
                y = net(x)
                n += 1
                if not torch.equal(y, ref):
                    bad += 1
    print('inference %2d x %3dx%-3d: %5d runs, %d differ' % (B, H, W, n, bad), flush=True)
    bad_total += bad
    tnet = arch.RRDBNet(3, 3, 64, nb).to(dev).train().set_precision('fp16')
    tnet.load_state_dict(sd)
    gy = synth.normal_like(6, 'soak.gy', (B, 3, 4 * H, 4 * W)).to(dev)

    def fb():
        torch.manual_seed(7)
        for p in tnet.parameters():
            p.grad = None
        y = tnet(x)
        (y * gy).sum().backward()
        return y.detach().clone(), torch.cat([p.grad.reshape(-1) for p in tnet.parameters()])
    y0, g0 = fb()
    n = bad = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget:
        for _ in range(5):
            y, g = fb()
            n += 1
            if not (torch.equal(y, y0) and torch.equal(g, g0)):
                bad += 1
    print('training  %2d x %3dx%-3d: %5d runs, %d differ' % (B, H, W, n, bad), flush=True)
    bad_total += bad
print('TOTAL differing runs:', bad_total)
sys.exit(1 if bad_total else 0)
