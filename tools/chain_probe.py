import sys, torch, time
sys.path.insert(0, '.')
from esrganplus_amd import architecture as arch, block as Bk, synth, _lib as L
from oracle import ref_torch as RT
dev = torch.device('cuda:0')
torch.manual_seed(0)
def check(nb, shape, prec='fp32'):
    sd = synth.rrdbnet_state_dict(nb=nb, seed=1)
    x = synth.image_batch(1, shape[0], 3, shape[1], shape[2], name='probe.x')
    with torch.no_grad():
        ref = RT.rrdbnet_forward(x, sd, nb)
        net = arch.RRDBNet(3, 3, 64, nb).to(dev).eval()
        net.load_state_dict(sd, strict=True)
        net.set_precision(prec)
        y = net(x.to(dev)).cpu()
        torch.cuda.synchronize()
    plans = list(net._plans.values())
    ws = [p.chain_ws for p in plans if getattr(p, 'chain_ws', None) is not None]
    ab = [int(w[1].item()) for w in ws]
    print('nb=%d %s %s: max|hip-oracle| = %.3e  chain plans %d abort %s' % (nb, shape, prec, (y - ref).abs().max().item(), len(ws), ab), flush=True)
check(1, (1, 12, 20))
check(1, (1, 16, 32))
check(1, (2, 24, 40))
check(2, (2, 40, 72))
check(2, (1, 57, 86))
check(2, (2, 40, 72), 'fp16')
