"""Stand-alone timing of esr_rdb_wgrad_run (csrc/rdb_wgrad.hip) at the fwd+bwd bench shape: n_blocks dense blocks'
weight gradients in ONE launch, batch 16 of 128x128 LR, fp16.  Usage: python tools/rdb_wgrad_probe.py [n_blocks] [B] [H] [W]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from esrganplus_amd import _lib as L, engine as E

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 69
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
H = int(sys.argv[3]) if len(sys.argv) > 3 else 128
W = int(sys.argv[4]) if len(sys.argv) > 4 else 128
dev = torch.device('cuda:0')
torch.manual_seed(0)
SHARED = os.environ.get('ESR_PROBE_SHARED') == '1'     # every block reads the SAME two tensors (cache-resident operands)
ins = [E.G32(B, 192, H, W, 'fp16', dev) for _ in range(1 if SHARED else nb)]
qs = [E.G32(B, 224, H, W, 'fp16', dev) for _ in range(1 if SHARED else nb)]
for g in ins + qs:
    g.t[:, :, 1:H + 1, 1:W + 1].normal_()
flat = torch.zeros(nb * (241664 + 192), dtype=torch.float32, device=dev)
blocks = (L.esr_rdb_wgrad_block * nb)()
couts, cins = [32, 32, 32, 32, 64, 32], [64, 96, 128, 160, 192, 64]
for i in range(nb):
    b = blocks[i]
    b.in_, b.q = ins[0 if SHARED else i].view(0, 192), qs[0 if SHARED else i].view(0, 224)
    off = i * (241664 + 192)
    for k in range(6):
        b.dw[k] = flat.data_ptr() + 4 * off
        off += couts[k] * cins[k] * (9 if k < 5 else 1)
    for k in range(5):
        b.db[k] = flat.data_ptr() + 4 * off
        off += couts[k]
blk_t = torch.frombuffer(bytearray(bytes(blocks)), dtype=torch.uint8).to(dev)
need = int(L.lib().esr_rdb_wgrad_workspace_elems(B, H, W, nb))
arena = torch.empty(need, dtype=torch.float32, device=dev)
rw = L.esr_rdb_wgrad()
rw.dtype, rw.B, rw.H, rw.W, rw.n_blocks, rw.tap_major = L.ESR_F16, B, H, W, nb, 1
rw.scale5, rw.scale, rw.blocks = 0.2, 1.0, blk_t.data_ptr()
rw.partial, rw.partial_elems = arena.data_ptr(), need
st = E.current_stream()
for _ in range(2):
    L.check(L.lib().esr_rdb_wgrad_run(C.byref(rw), C.c_void_p(st)), 'run')
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
reps = 5
ev[0].record()
for _ in range(reps):
    L.check(L.lib().esr_rdb_wgrad_run(C.byref(rw), C.c_void_p(st)), 'run')
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / reps
fl = 2.0 * 241664 * B * H * W * nb
steps = nb * 4 * B * ((W + 31) // 32) * ((H + 3) // 4)
print('  %s operands; %d workgroup-steps, %.2f us per step on 256 CUs' % ('shared' if SHARED else 'distinct', steps, ms * 1e3 / (steps / 256.0)))
print('rdb_wgrad: %d blocks, %dx%dx%d: %.3f ms per launch pair = %.1f us per block, %.1f TFLOP/s (%.3f of fp16 peak); arena %.2f GB'
      % (nb, B, H, W, ms, ms * 1e3 / nb, fl / ms / 1e9, fl / ms / 1e9 / 2500.0, need * 4 / 1e9))
