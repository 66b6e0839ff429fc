import torch
dev = torch.device('cuda:0')
x = torch.empty(16 * 512 * 512 * 64, dtype=torch.float16, device=dev)
y = torch.empty_like(x)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
us = t(lambda: x.fill_(1.0)); print('fill 537 MB: %.1f us = %.2f TB/s write' % (us, x.numel() * 2 / us / 1e6))
us = t(lambda: y.copy_(x)); print('copy 537 MB: %.1f us = %.2f TB/s read + same write' % (us, x.numel() * 2 / us / 1e6))
us = t(lambda: x.sum()); print('sum 537 MB: %.1f us = %.2f TB/s read' % (us, x.numel() * 2 / us / 1e6))
