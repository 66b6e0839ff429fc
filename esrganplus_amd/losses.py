"""The ESRGAN+ train step's losses as single launches (csrc/loss_kernels.hip): ``nn.L1Loss`` (cri_pix / cri_fea,
SRRaGAN_model.py:124-131) and the relativistic-average GAN term built from ``GANLoss('vanilla')``
(codes/models/modules/loss.py:6-38; SRRaGAN_model.py:133-137, 150-156).  Each forward launch also produces the
gradient w.r.t. its differentiable operands; backward is one multiply by the upstream scalar.

Same numbers as the torch formulas (tests/test_gpu_losses.py).  No fallback: operands the kernels do not cover (CPU
tensors, non-fp32, shape mismatch, a target that needs a gradient) raise ``HipExtensionError``; non-contiguous or
16-byte-misaligned views are handled (a contiguous copy / the kernel's scalar path).  Data-parallel runs keep the
relativistic means GLOBAL: the same kernel in three modes with two scalar all-reduces in between
(``ragan_loss(..., global_mean=True)``).
"""
import ctypes as C

import torch

from . import _lib as L
from . import engine as E
from . import dp as DP

_scratch = {}


def _dev_scratch(dev):
    t = _scratch.get(dev)
    if t is None:
        t = _scratch[dev] = torch.zeros(2, dtype=torch.float64, device=dev)
    return t


def _require(ok, what):
    if not ok:
        raise L.HipExtensionError('esrganplus_amd.losses: ' + what + ' (the fused loss kernels take fp32 tensors on '
                                  'the MI355X; there is no torch fallback)')


def _check_operands(name, *ts):
    for t in ts:
        _require(torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32,
                 '%s: operand is %s / %s' % (name, getattr(t, 'device', type(t)), getattr(t, 'dtype', '')))


class _L1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, weight):
        a_, b_ = a.detach().contiguous(), b.detach().contiguous()
        loss = torch.empty((), dtype=torch.float32, device=a.device)
        grad = torch.empty_like(a_) if ctx.needs_input_grad[0] else None
        p = L.esr_l1_loss()
        p.a, p.b, p.loss, p.n, p.weight = a_.data_ptr(), b_.data_ptr(), loss.data_ptr(), a_.numel(), weight
        p.grad_a = grad.data_ptr() if grad is not None else None
        p.scratch = _dev_scratch(a.device).data_ptr()
        L.check(L.lib().esr_l1_loss_forward(C.byref(p), C.c_void_p(E.current_stream())), 'esr_l1_loss_forward')
        ctx.grad = grad
        return loss

    @staticmethod
    def backward(ctx, g):
        return (ctx.grad * g if ctx.grad is not None else None), None, None


def _as_f32(t):
    # fp16 / bf16 operands (an autocast caller): upcast — autograd carries the gradient back through the cast
    return t.float() if torch.is_tensor(t) and t.is_cuda and t.dtype in (torch.float16, torch.bfloat16) else t


def _scale_args(p, grad_scale, scale_dev):
    p.grad_scale = float(grad_scale)
    if scale_dev is not None:
        p.grad_scale_dev = scale_dev.data_ptr()


def l1_raw(a, b, weight, grad_out=None, grad_scale=1.0, scale_dev=None):
    """One launch, no autograd: ``(weight * mean|a - b|, grad)`` with ``grad = grad_scale [* scale_dev[0]] * weight *
    sign(a - b) / n`` written into ``grad_out`` (a contiguous fp32 tensor of a's size — e.g. the buffer a backward launch
    list reads its upstream gradient from) or not formed at all (``grad_out=None``).  The hand-written train step
    (train.ESRGANPlusStep) calls the losses this way; ``l1_loss`` is the autograd face of the same kernel."""
    a_, b_ = a.detach().contiguous(), b.detach().contiguous()
    _check_operands('l1_raw', a_, b_)
    _require(a_.shape == b_.shape, 'l1_raw: shapes %s vs %s' % (tuple(a_.shape), tuple(b_.shape)))
    loss = torch.empty((), dtype=torch.float32, device=a_.device)
    p = L.esr_l1_loss()
    p.a, p.b, p.loss, p.n, p.weight = a_.data_ptr(), b_.data_ptr(), loss.data_ptr(), a_.numel(), float(weight)
    if grad_out is not None:
        _require(grad_out.is_contiguous() and grad_out.dtype == torch.float32 and grad_out.numel() == a_.numel(), 'l1_raw: grad_out')
        p.grad_a = grad_out.data_ptr()
    p.scratch = _dev_scratch(a_.device).data_ptr()
    _scale_args(p, grad_scale, scale_dev)
    L.check(L.lib().esr_l1_loss_forward(C.byref(p), C.c_void_p(E.current_stream())), 'esr_l1_loss_forward')
    return loss


def ragan_raw(x, y, x_is_real, y_is_real, weight, grad_x=None, grad_y=None, grad_scale=1.0, scale_dev=None,
              global_mean=False):
    """The relativistic-average GAN term without autograd: ``(loss, aux)`` as ``ragan_loss``; the gradients w.r.t. the
    logits, times ``grad_scale [* scale_dev[0]]``, are written into ``grad_x`` / ``grad_y`` (contiguous fp32, n floats;
    None: not formed).  ``global_mean``: the batch means run over all ranks (two scalar all-reduces between the launches,
    as ``_RaGANGlobalFn``)."""
    import torch.distributed as dist
    x_, y_ = x.detach().contiguous().view(-1), y.detach().contiguous().view(-1)
    _check_operands('ragan_raw', x_, y_)
    _require(x_.numel() == y_.numel(), 'ragan_raw: %d vs %d logits' % (x_.numel(), y_.numel()))
    dev = x_.device
    st = C.c_void_p(E.current_stream())
    loss = torch.empty((), dtype=torch.float32, device=dev)
    out = torch.empty(4, dtype=torch.float32, device=dev)        # mean(x), mean(y), BCE_x, BCE_y
    p = L.esr_ragan_loss()
    p.x, p.y, p.n = x_.data_ptr(), y_.data_ptr(), x_.numel()
    p.tx, p.ty, p.weight = (1.0 if x_is_real else 0.0), (1.0 if y_is_real else 0.0), float(weight)
    p.loss, p.mean_x, p.mean_y = loss.data_ptr(), out.data_ptr(), out.data_ptr() + 4
    p.bce_x, p.bce_y = out.data_ptr() + 8, out.data_ptr() + 12
    _scale_args(p, grad_scale, scale_dev)
    gx = grad_x.data_ptr() if grad_x is not None else None
    gy = grad_y.data_ptr() if grad_y is not None else None
    if not (global_mean and DP.active()):
        p.grad_x, p.grad_y = gx, gy
        L.check(L.lib().esr_ragan_loss_forward(C.byref(p), st), 'esr_ragan_loss_forward')
        return loss, out
    ext = torch.zeros(5, dtype=torch.float32, device=dev)       # sum x, sum y, n | D1, D2 (all ranks)
    ext[2] = float(x_.numel())
    p.mode, p.sums = 1, ext.data_ptr()
    L.check(L.lib().esr_ragan_loss_forward(C.byref(p), st), 'esr_ragan_loss_forward')
    dist.all_reduce(ext[0:3])
    dsum = torch.zeros(2, dtype=torch.float32, device=dev)
    p.mode, p.sums, p.ext = 2, dsum.data_ptr(), ext.data_ptr()
    L.check(L.lib().esr_ragan_loss_forward(C.byref(p), st), 'esr_ragan_loss_forward')
    if gx is not None or gy is not None:
        dist.all_reduce(dsum)
        ext[3:5] = dsum
        p.mode, p.grad_x, p.grad_y = 3, gx, gy
        L.check(L.lib().esr_ragan_loss_forward(C.byref(p), st), 'esr_ragan_loss_forward')
    return loss, out


def l1_loss(a, b, weight=1.0):
    """``weight * F.l1_loss(a, b)``; the gradient flows to ``a`` only (``b`` is the target: var_H / real_fea — a target
    that requires a gradient is refused, not silently detached).  fp16 / bf16 operands are upcast."""
    a, b = _as_f32(a), _as_f32(b)
    _check_operands('l1_loss', a, b)
    _require(a.shape == b.shape, 'l1_loss: shapes %s vs %s' % (tuple(a.shape), tuple(b.shape)))
    _require(not b.requires_grad, 'l1_loss: the target must not require a gradient')
    return _L1Fn.apply(a, b, float(weight))


class _RaGANFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, tx, ty, weight):
        x_, y_ = x.detach().contiguous().view(-1), y.detach().contiguous().view(-1)
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        out = torch.empty(4, dtype=torch.float32, device=x.device)        # mean(x), mean(y), BCE_x, BCE_y
        gx = torch.empty_like(x_) if ctx.needs_input_grad[0] else None
        gy = torch.empty_like(y_) if ctx.needs_input_grad[1] else None
        p = L.esr_ragan_loss()
        p.x, p.y, p.n = x_.data_ptr(), y_.data_ptr(), x_.numel()
        p.grad_x = gx.data_ptr() if gx is not None else None
        p.grad_y = gy.data_ptr() if gy is not None else None
        p.loss, p.mean_x, p.mean_y = loss.data_ptr(), out.data_ptr(), out.data_ptr() + 4
        p.bce_x, p.bce_y = out.data_ptr() + 8, out.data_ptr() + 12
        p.tx, p.ty, p.weight = tx, ty, weight
        L.check(L.lib().esr_ragan_loss_forward(C.byref(p), C.c_void_p(E.current_stream())), 'esr_ragan_loss_forward')
        ctx.g = (gx, gy, x.shape, y.shape)
        ctx.mark_non_differentiable(out)
        return loss, out

    @staticmethod
    def backward(ctx, g, _unused):
        gx, gy, sx, sy = ctx.g
        return ((gx * g).view(sx) if gx is not None else None, (gy * g).view(sy) if gy is not None else None,
                None, None, None)


class _RaGANGlobalFn(torch.autograd.Function):
    """The same loss with the batch means taken over ALL ranks (SRRaGAN_model.py:136-137,151-152: under the
    reference's DataParallel the loss is formed on the gathered outputs).  Three launches of the one
    kernel; between them the rank-local scalar sums are all-reduced: [sum x, sum y, n] before the loss,
    [sum (sigmoid(z1) - tx), sum (sigmoid(z2) - ty)] before the gradients (every rank's loss sees the means, and the
    parameter gradients are averaged over ranks afterwards — GradExchange — which together reproduce the gradient of
    the single global-batch loss)."""

    @staticmethod
    def forward(ctx, x, y, tx, ty, weight):
        import torch.distributed as dist
        x_, y_ = x.detach().contiguous().view(-1), y.detach().contiguous().view(-1)
        dev = x.device
        st = C.c_void_p(E.current_stream())
        ext = torch.zeros(5, dtype=torch.float32, device=dev)       # sum x, sum y, n | D1, D2 (all ranks)
        ext[2] = float(x_.numel())
        p = L.esr_ragan_loss()
        p.x, p.y, p.n = x_.data_ptr(), y_.data_ptr(), x_.numel()
        p.tx, p.ty, p.weight = tx, ty, weight
        p.mode, p.sums = 1, ext.data_ptr()
        L.check(L.lib().esr_ragan_loss_forward(C.byref(p), st), 'esr_ragan_loss_forward')
        if DP.active():
            dist.all_reduce(ext[0:3])
        loss = torch.empty((), dtype=torch.float32, device=dev)
        out = torch.empty(4, dtype=torch.float32, device=dev)
        dsum = torch.zeros(2, dtype=torch.float32, device=dev)
        p.mode, p.sums, p.ext = 2, dsum.data_ptr(), ext.data_ptr()
        p.loss, p.mean_x, p.mean_y = loss.data_ptr(), out.data_ptr(), out.data_ptr() + 4
        p.bce_x, p.bce_y = out.data_ptr() + 8, out.data_ptr() + 12
        L.check(L.lib().esr_ragan_loss_forward(C.byref(p), st), 'esr_ragan_loss_forward')
        ctx.need = (ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        ctx.keep = (x_, y_, ext, dsum, tx, ty, weight, x.shape, y.shape)
        ctx.mark_non_differentiable(out)
        return loss, out

    @staticmethod
    def backward(ctx, g, _unused):
        import torch.distributed as dist
        x_, y_, ext, dsum, tx, ty, weight, sx, sy = ctx.keep
        if DP.active():
            dist.all_reduce(dsum)
        ext[3:5] = dsum
        gx = torch.empty_like(x_) if ctx.need[0] else None
        gy = torch.empty_like(y_) if ctx.need[1] else None
        p = L.esr_ragan_loss()
        p.x, p.y, p.n = x_.data_ptr(), y_.data_ptr(), x_.numel()
        p.tx, p.ty, p.weight = tx, ty, weight
        p.mode, p.ext = 3, ext.data_ptr()
        p.grad_x = gx.data_ptr() if gx is not None else None
        p.grad_y = gy.data_ptr() if gy is not None else None
        L.check(L.lib().esr_ragan_loss_forward(C.byref(p), C.c_void_p(E.current_stream())), 'esr_ragan_loss_forward')
        return ((gx * g).view(sx) if gx is not None else None, (gy * g).view(sy) if gy is not None else None,
                None, None, None)


def ragan_loss(x, y, x_is_real, y_is_real, weight=1.0, global_mean=False):
    """``weight * (BCE(x - mean(y), x_is_real) + BCE(y - mean(x), y_is_real)) / 2`` and the two means
    (returned as ``(loss, aux)`` with ``aux = [mean_x, mean_y, BCE_x, BCE_y]``, detached).  ``global_mean``: the means
    run over the batch of ALL ranks (data-parallel training; == the local batch on one rank)."""
    x, y = _as_f32(x), _as_f32(y)
    _check_operands('ragan_loss', x, y)
    _require(x.numel() == y.numel(), 'ragan_loss: %d vs %d logits' % (x.numel(), y.numel()))
    fn = _RaGANGlobalFn if global_mean else _RaGANFn
    return fn.apply(x, y, 1.0 if x_is_real else 0.0, 1.0 if y_is_real else 0.0, float(weight))
