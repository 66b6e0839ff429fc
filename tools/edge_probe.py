import sys, torch
sys.path.insert(0, '/root/repo')
from esrganplus_amd import architecture as arch, synth
from oracle import ref_torch as RT
dev = torch.device('cuda:0')
nb = 1
sd = synth.rrdbnet_state_dict(nb=nb, seed=2)
net = arch.RRDBNet(3, 3, 64, nb).to(dev).eval()
net.load_state_dict(sd)
for shape in [(1, 3, 1, 1), (1, 3, 2, 3), (3, 3, 7, 5), (1, 3, 33, 31), (2, 3, 1, 40), (1, 3, 64, 1)]:
    x = synth.image_batch(5, *shape, name='edge.x')
    with torch.no_grad():
        ref = RT.rrdbnet_forward(x, sd, nb)
        y = net(x.to(dev)).cpu()
        y16 = net.set_precision('fp16')(x.to(dev)).cpu()
        net.set_precision('fp32')
    print(shape, 'fp32 err %.2e fp16 err %.2e' % ((y - ref).abs().max().item(), (y16 - ref).abs().max().item()))
for shape in [(0, 3, 8, 8), (1, 3, 0, 8)]:
    try:
        with torch.no_grad():
            y = net(torch.zeros(shape, device=dev))
        print(shape, '->', tuple(y.shape))
    except Exception as e:
        print(shape, 'raised', type(e).__name__, str(e)[:100])
# training on odd size
net.train()
x = synth.image_batch(6, 2, 3, 5, 9, name='edge.t').to(dev)
y = net(x); y.mean().backward(); torch.cuda.synchronize()
print('train 5x9 ok', y.shape, float(net.model[0].weight.grad.abs().sum()))
# big tile
net.eval().set_precision('fp16')
with torch.no_grad():
    y = net(torch.rand(1, 3, 512, 384, device=dev)); torch.cuda.synchronize()
print('big ok', tuple(y.shape), bool(torch.isfinite(y).all()))
