"""esr_rdb_wgrad_run (csrc/rdb_wgrad.hip): the six weight / bias gradients of a ResidualDenseBlock_5C
(block.py:239-268; autograd's conv backward-weight, SRRaGAN_model.py:140) in one pass, against
 (a) an fp64 torch evaluation of the same sums on the fp16-rounded operands (the oracle of this kernel: a weight
     gradient is a plain correlation, torch.nn.grad-free), and
 (b) the per-conv esr_conv_wgrad launches it replaces;
edge cases: ragged sizes, single rows / columns, several column strips and image groups; run-to-run bit identity."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from esrganplus_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _run(dev, B, H, W, seed, tap_major=False):
    from esrganplus_amd import _lib as L, engine as E
    x = synth.normal_like(seed, 'rw.in', (B, 192, H, W)).to(dev)
    q = (0.05 * synth.normal_like(seed, 'rw.q', (B, 224, H, W))).to(dev)
    bin_, bq = E.G32(B, 192, H, W, 'fp16', dev), E.G32(B, 224, H, W, 'fp16', dev)
    st = E.current_stream()
    for t, g, c in ((x, bin_, 192), (q, bq, 224)):
        lo = L.esr_layout()
        lo.dtype, lo.to_g32, lo.B, lo.C, lo.H, lo.W = L.ESR_F16, 1, B, c, H, W
        lo.nchw, lo.g32 = t.contiguous().data_ptr(), g.view(0, c)
        L.check(L.lib().esr_convert_layout(C.byref(lo), C.c_void_p(st)), 'layout')
    couts, cins = [32, 32, 32, 32, 64, 32], [64, 96, 128, 160, 192, 64]
    dws = [torch.zeros((co, ci, 3, 3) if k < 5 else (co, ci, 1, 1), dtype=torch.float32, device=dev)
           for k, (co, ci) in enumerate(zip(couts, cins))]
    if tap_major:
        dws = [torch.zeros((9, co, ci), dtype=torch.float32, device=dev) if k < 5 else d
               for k, (d, co, ci) in enumerate(zip(dws, couts, cins))]
    dbs = [torch.zeros(co, dtype=torch.float32, device=dev) for co in couts[:5]]
    wb = L.esr_rdb_wgrad_block()
    wb.in_, wb.q = bin_.view(0, 192), bq.view(0, 224)
    for k in range(6):
        wb.dw[k] = dws[k].data_ptr()
    for k in range(5):
        wb.db[k] = dbs[k].data_ptr()
    blk_t = torch.frombuffer(bytearray(bytes(wb)), dtype=torch.uint8).to(dev)
    need = int(L.lib().esr_rdb_wgrad_workspace_elems(B, H, W, 1))
    arena = torch.empty(need, dtype=torch.float32, device=dev)
    rw = L.esr_rdb_wgrad()
    rw.dtype, rw.B, rw.H, rw.W, rw.n_blocks, rw.tap_major = L.ESR_F16, B, H, W, 1, 1 if tap_major else 0
    rw.scale5, rw.scale, rw.blocks = 0.2, 1.0, blk_t.data_ptr()
    rw.partial, rw.partial_elems = arena.data_ptr(), need
    L.check(L.lib().esr_rdb_wgrad_run(C.byref(rw), C.c_void_p(st)), 'esr_rdb_wgrad_run')
    torch.cuda.synchronize()
    return x, q, dws, dbs, (bin_, bq)


def _reference(x, q):
    """fp64 correlation sums on the fp16-rounded operands."""
    xh, qh = x.half().double(), q.half().double()
    gsl = {4: (0, 64, 0.2), 3: (64, 32, 1.0), 2: (96, 32, 1.0), 1: (128, 32, 1.0), 0: (160, 32, 1.0)}   # conv k -> Q slice
    dws, dbs = [], []
    for k in range(5):
        c0, n, sc = gsl[k]
        g = qh[:, c0:c0 + n] * sc
        xin = xh[:, :64 + 32 * k]
        # dW[co, ci, kh, kw] = sum_b,y,x g[b, co, y, x] * xpad[b, ci, y + kh, x + kw]
        xp = F.pad(xin, (1, 1, 1, 1))
        H, W = g.shape[-2:]
        dw = torch.stack([torch.stack([torch.einsum('bohw,bihw->oi', g, xp[:, :, kh:kh + H, kw:kw + W]) for kw in range(3)], -1)
                          for kh in range(3)], -2)
        dws.append(dw)
        dbs.append(g.sum((0, 2, 3)))
    dws.append(torch.einsum('bohw,bihw->oi', qh[:, 192:224], xh[:, :64])[:, :, None, None])
    return dws, dbs


@pytest.mark.parametrize('shape', [(2, 16, 32), (1, 5, 7), (3, 33, 40), (1, 1, 1), (4, 12, 70), (9, 8, 32)])
def test_rdb_wgrad_matches_fp64_correlation(dev, shape):
    B, H, W = shape
    x, q, dws, dbs, _ = _run(dev, B, H, W, seed=3 + H)
    rdw, rdb = _reference(x, q)
    for k in range(6):
        a, r = dws[k].double(), rdw[k]
        err = (a - r).abs().max().item()
        scale = r.abs().max().item() + 1e-12
        # fp32 accumulation of exact fp16 products over B*H*W pixels
        assert err <= 2e-5 * scale + 1e-6, (k, err, scale)
    for k in range(5):
        err = (dbs[k].double() - rdb[k]).abs().max().item()
        assert err <= 2e-5 * (rdb[k].abs().max().item() + 1e-12) + 1e-6, (k, err)


def test_rdb_wgrad_tap_major_and_accumulate(dev):
    """tap_major: 3x3 gradients as [tap][cout][cin] (what esr_grad_unpermute rewrites); a second call adds on top."""
    B, H, W = 2, 20, 36
    x, q, dws, dbs, _ = _run(dev, B, H, W, seed=11, tap_major=True)
    rdw, _ = _reference(x, q)
    for k in range(5):
        want = rdw[k].permute(2, 3, 0, 1).reshape(9, rdw[k].shape[0], rdw[k].shape[1])
        assert (dws[k].double() - want).abs().max().item() <= 2e-5 * want.abs().max().item() + 1e-6, k
    assert (dws[5].double() - rdw[5]).abs().max().item() <= 2e-5 * rdw[5].abs().max().item() + 1e-6


def test_rdb_wgrad_is_bit_identical_run_to_run_and_matches_per_conv_launches(dev):
    from esrganplus_amd import _lib as L, engine as E
    B, H, W = 4, 32, 64
    outs = [_run(dev, B, H, W, seed=21) for _ in range(2)]
    for a, b in zip(outs[0][2] + outs[0][3], outs[1][2] + outs[1][3]):
        assert torch.equal(a, b)
    # the per-conv fp16 kernel on the same G32 tensors (atomics: compare to rounding)
    x, q, dws, dbs, (bin_, bq) = outs[0]
    st = E.current_stream()
    spec = [(0, 160, 32, 64, 3, 1.0), (1, 128, 32, 96, 3, 1.0), (2, 96, 32, 128, 3, 1.0), (3, 64, 32, 160, 3, 1.0),
            (4, 0, 64, 192, 3, 0.2), (5, 192, 32, 64, 1, 1.0)]
    for k, c0, co, ci, ks, sc in spec:
        wg = L.esr_wgrad()
        wg.dtype, wg.ks, wg.stride, wg.upsample = L.ESR_F16, ks, 1, 0
        wg.B, wg.H, wg.W, wg.cout, wg.cin = B, H, W, co, ci
        wg.g, wg.in_ = bq.view(c0, co), bin_.view(0, ci)
        dw = torch.zeros((co, ci, ks, ks), dtype=torch.float32, device=dev)
        db = torch.zeros(co, dtype=torch.float32, device=dev)
        wg.dw, wg.dbias, wg.scale = dw.data_ptr(), (db.data_ptr() if k < 5 else None), sc
        L.check(L.lib().esr_conv_wgrad(C.byref(wg), C.c_void_p(st)), 'esr_conv_wgrad')
        torch.cuda.synchronize()
        assert (dw - dws[k]).abs().max().item() <= 1e-5 * dw.abs().max().item() + 1e-6, k
        if k < 5:
            assert (db - dbs[k]).abs().max().item() <= 1e-5 * db.abs().max().item() + 1e-6, k
