import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esrganplus_amd import architecture as arch, synth
dev = torch.device('cuda:0')
prec = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
mode = sys.argv[2] if len(sys.argv) > 2 else 'dp'
net = arch.RRDBNet(3, 3, 64, 2).to(dev).train().set_precision(prec)
net.load_state_dict(synth.rrdbnet_state_dict(nb=2, seed=3))
x = synth.image_batch(3, 4, 3, 24, 32, name='dpt.x').to(dev)
gy = synth.normal_like(3, 'dpt.gy', (4, 3, 96, 128)).to(dev)
net.eval()
ref = net(x)
(ref * gy).sum().backward()
g_ref = {k: p.grad.clone() for k, p in net.named_parameters()}
net.zero_grad(set_to_none=True)
if mode == 'dp':
    y = torch.nn.DataParallel(net, device_ids=[0, 0])(x)
elif mode == 'dpseq':         # DataParallel's scatter / replicate / gather, replicas applied one after the other
    dpn = torch.nn.DataParallel(net, device_ids=[0, 0])
    dpn.parallel_apply = lambda replicas, inputs, kwargs: [r(*i, **k) for r, i, k in zip(replicas, inputs, kwargs)]
    y = dpn(x)
elif mode == 'thr':           # replicate + two threads (no scatter / gather autograd nodes)
    import threading
    reps = torch.nn.parallel.replicate(net, [0, 0])
    outs = [None, None]
    def run(i):
        with torch.cuda.device(0):
            outs[i] = reps[i](x[2 * i:2 * i + 2])
    ts = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    [t.start() for t in ts]; [t.join() for t in ts]
    y = torch.cat(outs)
elif mode == 'seq':           # two replicas, driven one after the other from this thread
    reps = torch.nn.parallel.replicate(net, [0, 0])
    y = torch.cat([reps[0](x[:2]), reps[1](x[2:])])
elif mode == 'seqdel':        # as 'seq', but nothing keeps the replicas alive (as under DataParallel)
    reps = torch.nn.parallel.replicate(net, [0, 0])
    y = torch.cat([reps[0](x[:2]), reps[1](x[2:])])
    del reps
    import gc; gc.collect()
elif mode == 'seqsc':         # as 'seq', inputs through Scatter / outputs through Gather
    from torch.nn.parallel.scatter_gather import scatter, gather
    reps = torch.nn.parallel.replicate(net, [0, 0])
    xs = scatter(x, [0, 0])
    y = gather([reps[0](xs[0]), reps[1](xs[1])], 0)
elif mode == 'one':           # ONE replica over the whole batch
    reps = torch.nn.parallel.replicate(net, [0])
    y = reps[0](x)
(y * gy).sum().backward()
torch.cuda.synchronize()
print(prec, mode, 'out err %.3e' % (y - ref).abs().max().item())
errs = sorted(((net.get_parameter(k).grad - g).abs().max().item() / (g.abs().max().item() + 1e-12), k) for k, g in g_ref.items())
print('worst:', errs[-6:])
print('n bad:', sum(1 for e, _ in errs if e > 1e-3), 'of', len(errs))
