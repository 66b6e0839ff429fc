#!/usr/bin/env python
"""Which torch streams run CONCURRENTLY on this box?  HIP maps streams onto a few hardware queues (GPU_MAX_HW_QUEUES,
default 4); two streams on one queue serialise.  For every pair (a, b) of candidate streams: hold one workgroup on a
(esr_debug_hold_cus, 30 ms), time a tiny kernel on b.  Prints the concurrency matrix."""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esrganplus_amd import _lib as L

dev = torch.device('cuda:0')
torch.zeros(1, device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(n)]
words = torch.zeros(16, dtype=torch.int32).pin_memory()
p_release, p_started = C.c_void_p(words.data_ptr()), C.c_void_p(words.data_ptr() + 4)
x = torch.zeros(64, device=dev)
torch.cuda.synchronize()
print('cuda_stream handles:', [hex(s.cuda_stream) for s in streams])
for i, a in enumerate(streams):
    row = []
    for j, b in enumerate(streams):
        if i == j:
            row.append(' . ')
            continue
        words.zero_()
        L.check(L.lib().esr_debug_hold_cus(1, p_release, 30, p_started, C.c_void_p(a.cuda_stream)), 'hold')
        with torch.cuda.stream(b):
            t0 = time.perf_counter()
            x.add_(1.0)
            b.synchronize()
            dt = time.perf_counter() - t0
        words[0] = 1
        torch.cuda.synchronize()
        row.append(' C ' if dt < 0.01 else ' s ')
    print('%2d' % i, ''.join(row))
