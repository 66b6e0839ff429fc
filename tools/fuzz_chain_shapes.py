"""Random small shapes through the fused chains (inference, training forward + backward; tile height chosen by the
library and forced to 16 / 8 / 4 rows) against the per-conv launches.  Usage (GPU box): python tools/fuzz_chain_shapes.py [n] [seed]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esrganplus_amd import architecture as arch, synth
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
sd = synth.rrdbnet_state_dict(nb=2, seed=61, gain=0.7)
worst = 0.0
for it in range(n):
    B, H, W = rnd.randint(1, 4), rnd.randint(1, 45), rnd.randint(1, 75)
    x = synth.image_batch(it, B, 3, H, W, name='fuzz.x').to(dev)
    gy = synth.normal_like(it, 'fuzz.gy', (B, 3, 4 * H, 4 * W)).to(dev)
    res = {}
    for tag, env in (('conv', {'ESR_RDB_FUSED': '0', 'ESR_RDB_TRAIN_CHAIN': '0'}), ('auto', {}), ('r4', {'ESR_RDB_ROWS': '4'}),
                     ('r2', {'ESR_RDB_ROWS': '2'}), ('r1', {'ESR_RDB_ROWS': '1'})):
        for k in ('ESR_RDB_FUSED', 'ESR_RDB_TRAIN_CHAIN', 'ESR_RDB_ROWS'):
            os.environ.pop(k, None)
        os.environ.update(env)
        net = arch.RRDBNet(3, 3, 64, 2).to(dev).set_precision('fp16')
        net.load_state_dict(sd)
        with torch.no_grad():
            ye = net.eval()(x).clone()
        net.train()
        torch.manual_seed(5)
        yt = net(x)
        (yt * gy).sum().backward()
        torch.cuda.synchronize()
        g = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
        res[tag] = (ye, yt.detach().clone(), g.clone())
        assert torch.isfinite(ye).all() and torch.isfinite(g).all(), (tag, B, H, W)
    for tag in ('r2', 'r1'):
        for a, b in zip(res[tag], res['r4']):
            assert torch.equal(a, b), ('tile heights differ', tag, B, H, W)
    for a, b in zip(res['auto'], res['r4']):
        assert torch.equal(a, b), ('auto differs', B, H, W)
    e_eval = (res['r4'][0] - res['conv'][0]).abs().max().item()
    e_tr = (res['r4'][1] - res['conv'][1]).abs().max().item()
    rg = ((res['r4'][2] - res['conv'][2]).norm() / res['conv'][2].norm()).item()
    worst = max(worst, e_eval, e_tr)
    assert e_eval <= 3e-3 and e_tr <= 3e-3 and rg <= 3e-2, (B, H, W, e_eval, e_tr, rg)
    print('%2d  B %d  %2dx%2d  eval %.1e  train %.1e  grads rel %.1e' % (it, B, H, W, e_eval, e_tr, rg), flush=True)
print('fuzz ok, worst abs diff %.2e' % worst)
