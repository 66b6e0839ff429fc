#!/usr/bin/env python
"""Upper-bound probe: one RDB as 5 launches vs as ONE (unsynchronised, numerically invalid) launch."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esrganplus_amd import architecture as arch, synth, engine as E, _lib as L

dev = torch.device('cuda:0')
net = arch.RRDBNet(3, 3, 64, 1).to(dev).eval().set_precision('fp16')
net.load_state_dict(synth.rrdbnet_state_dict(1, 0))
BATCH = int(sys.argv[1]) if len(sys.argv) > 1 else 16
x = synth.image_batch(1, BATCH, 3, 128, 128).to(dev)
with torch.no_grad():
    net(x)
plan = next(iter(net._plans.values()))
ops = plan.ops.ops
convs = [o.u.conv for o in ops[2:7]]          # RDB1: conv1..conv5
arr = (L.esr_conv * 5)(*convs)
dbuf = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
st = E.current_stream()
lib = L.lib()
lib.esr_rdb_nosync_probe.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
tiles = BATCH * (128 // 16) * (128 // 32)
five = L.OpList()
for _ in range(20):
    for c in convs:
        five.add_conv(c)
import time


def timed(fn, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


one = L.OpList()
for c in convs:
    one.add_conv(c)
print('batch', BATCH)
for _ in range(2):
    print('5 launches              %.1f us per RDB' % timed(lambda: one.run(st), 200))
    nrep = int(os.environ.get('ESR_PROBE_NREP', '1'))
    print('1 launch (no halo sync, %d RDBs per launch, delay %s ticks) %.1f us per RDB' % (nrep, os.environ.get('ESR_PROBE_DELAY', '0'), timed(lambda: L.check(lib.esr_rdb_nosync_probe(dbuf.data_ptr(), tiles, st)), 100) / nrep))
