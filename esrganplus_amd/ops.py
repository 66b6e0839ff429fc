"""Single-op convenience wrappers over the C ABI (NCHW fp32 in/out).  Used by tests and by callers
that want one fused conv without building a plan; each call allocates its G32 buffers."""
import torch

from . import engine as E
from . import _lib as L


def conv2d(x, weight, bias=None, stride=1, act=None, upsample=False, precision='fp32',
           residual=None, alpha=1.0, subpix=False, ksplit=0, stats=None):
    """act(conv2d(x, weight, bias, stride, padding=(k-1)//2)) [* alpha + residual] — the
    conv_block of block.py:125-151 (optionally on a nearest-x2 upsampled input, block.py:315-322;
    subpix: that up-conv in its 4-phase 2x2 form, esr_conv.upsample == 3)."""
    E.require_cuda(x, 'input')
    x = x.detach().contiguous().float()
    B, Cin, H, W = x.shape
    Cout, Cin2, ks, _ = weight.shape
    assert Cin == Cin2
    dev = x.device
    pad = (ks - 1) // 2
    Hi, Wi = (2 * H, 2 * W) if upsample else (H, W)
    Ho, Wo = (Hi + 2 * pad - ks) // stride + 1, (Wi + 2 * pad - ks) // stride + 1
    w = weight.detach().contiguous().float()
    b = bias.detach().contiguous().float() if bias is not None else None
    wp = E.WeightPack([('c', w, b)], precision, dev, ('c',) if (subpix and upsample) else ())
    st = E.current_stream()
    wp.ensure(st, force=True)
    dt_e = wp.esr_dtype
    ops = L.OpList()
    xin = E.G32(B, Cin, H, W, precision, dev)
    # output buffer channel count rounded to whole 32-blocks so every group store is in range
    out = E.G32(B, ((Cout + 31) // 32) * 32, Ho, Wo, precision, dev)
    lo = L.esr_layout()
    lo.dtype, lo.to_g32, lo.B, lo.C, lo.H, lo.W = dt_e, 1, B, Cin, H, W
    lo.nchw, lo.g32 = x.data_ptr(), xin.view(0, Cin)
    ops.add(L.OP_LAYOUT, 'layout', lo)
    a = {None: L.ACT_NONE, 'leakyrelu': L.ACT_LRELU, 'relu': L.ACT_RELU}[act]
    c = E._conv(dt_e, B, Ho, Wo, xin.view(0), Cin, out.view(0), wp.entries['c'], a, stride=stride,
                upsample=1 if upsample else 0)
    keep = []
    if ksplit > 1:         # esr_conv.ksplit: packed tiles + split K (4x4/s2 on small square maps); stats: [groups] fp64 sums
        ws = torch.empty(ksplit * B * Ho * Wo * ((Cout + 31) // 32) * 32, dtype=torch.float32, device=dev)
        c.ksplit, c.split_ws = ksplit, ws.data_ptr()
        keep.append(ws)
        if stats is not None:
            c.stat_sums, c.stat_groups, c.stat_C = stats.data_ptr(), stats.shape[0], Cout
    if residual is not None:
        r = residual.detach().contiguous().float()
        rb = E.G32(B, ((Cout + 31) // 32) * 32, Ho, Wo, precision, dev)
        lr = L.esr_layout()
        lr.dtype, lr.to_g32, lr.B, lr.C, lr.H, lr.W = dt_e, 1, B, Cout, Ho, Wo
        lr.nchw, lr.g32 = r.data_ptr(), rb.view(0, Cout)
        ops.add(L.OP_LAYOUT, 'layout', lr)
        c.res1, c.alpha = rb.view(0), alpha
        keep += [r, rb]
    ops.add_conv(c)
    y = torch.empty(B, Cout, Ho, Wo, dtype=torch.float32, device=dev)
    lo2 = L.esr_layout()
    lo2.dtype, lo2.to_g32, lo2.B, lo2.C, lo2.H, lo2.W = dt_e, 0, B, Cout, Ho, Wo
    lo2.nchw, lo2.g32 = y.data_ptr(), out.view(0, Cout)
    ops.add(L.OP_LAYOUT, 'layout', lo2)
    ops.run(st)
    torch.cuda.current_stream().synchronize()   # buffers above die with this frame
    return y


def philox_normal(shape, seed, layer, device):
    """The N(0,1) tensor the fused noise epilogue uses for (seed, layer) — NCHW fp32."""
    B, C_, H, W = shape
    z = torch.empty(shape, dtype=torch.float32, device=device)
    nf = L.esr_noise_fill()
    nf.dst, nf.B, nf.C, nf.H, nf.W, nf.seed, nf.layer = z.data_ptr(), B, C_, H, W, seed, layer
    L.check(L.lib().esr_fill_noise(nf, E.current_stream()), 'esr_fill_noise')
    return z
