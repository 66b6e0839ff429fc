import sys, os, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ['x', '3']
import runpy, torch
from esrganplus_amd import _lib as L, functional as Fn, optim
acc = collections.defaultdict(lambda: [0, 0.0])
def wrap(obj, name, label=None):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); d = time.perf_counter() - t
        n = len(a[0].ops) if (label is None and hasattr(a[0], 'ops')) else 0
        key = label or ('%s[%d ops]' % (name, n))
        acc[key][0] += 1; acc[key][1] += d
        return r
    setattr(obj, name, g)
wrap(L.OpList, 'run'); wrap(L.OpList, 'run_range')
wrap(Fn, '_grad_views', 'grad_views'); wrap(Fn, '_train_backward', 'G _train_backward')
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'train_phases.py'))
acc.clear()
step = ns['step']
N = 10
t0 = time.perf_counter()
for _ in range(N): step([])
th = time.perf_counter() - t0
torch.cuda.synchronize()
print('host per step %.2f ms' % (th / N * 1e3))
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print('%-34s calls/step %5.1f  %7.3f ms/step' % (k, n / N, t / N * 1e3))
