"""The ESRGAN+ train step's losses as single launches (csrc/loss_kernels.hip): ``nn.L1Loss`` (cri_pix / cri_fea,
SRRaGAN_model.py:124-131) and the relativistic-average GAN term built from ``GANLoss('vanilla')``
(codes/models/modules/loss.py:6-38; SRRaGAN_model.py:133-137, 150-156).  Each forward launch also produces the
gradient w.r.t. its differentiable operands; backward is one multiply by the upstream scalar.

Same numbers as the torch formulas (tests/test_gpu_losses.py); anything the kernels do not cover (CPU tensors,
non-fp32, misaligned views) takes the torch formulas.
"""
import ctypes as C

import torch
import torch.nn.functional as F

from . import _lib as L
from . import engine as E

_scratch = {}


def _dev_scratch(dev):
    t = _scratch.get(dev)
    if t is None:
        t = _scratch[dev] = torch.zeros(2, dtype=torch.float64, device=dev)
    return t


def _fusable(*ts):
    return all(t.is_cuda and t.dtype == torch.float32 for t in ts)


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


def l1_loss(a, b, weight=1.0):
    """``weight * F.l1_loss(a, b)``; the gradient flows to ``a`` only (``b`` is the target: var_H / real_fea)."""
    if (not _fusable(a, b) or a.shape != b.shape or b.requires_grad or a.data_ptr() % 16 or b.data_ptr() % 16
            or not a.is_contiguous() or not b.is_contiguous()):
        return weight * F.l1_loss(a, b)
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


def ragan_loss(x, y, x_is_real, y_is_real, weight=1.0, mean=None):
    """``weight * (BCE(x - mean(y), x_is_real) + BCE(y - mean(x), y_is_real)) / 2`` and the two means
    (returned as ``(loss, aux)`` with ``aux = [mean_x, mean_y, BCE_x, BCE_y]``, detached).  ``mean``: a differentiable batch mean other than ``torch.mean``
    (dp.global_mean over all ranks) — then the torch formulas run, the fused kernel only knows the local batch."""
    if mean is not None or not _fusable(x, y) or x.numel() != y.numel():
        m = mean if mean is not None else torch.mean
        t = lambda v, real: torch.ones_like(v) if real else torch.zeros_like(v)
        lx = F.binary_cross_entropy_with_logits(x - m(y), t(x, x_is_real))
        ly = F.binary_cross_entropy_with_logits(y - m(x), t(y, y_is_real))
        aux = torch.stack([x.detach().mean(), y.detach().mean(), lx.detach(), ly.detach()])
        return weight * (lx + ly) / 2, aux
    return _RaGANFn.apply(x, y, 1.0 if x_is_real else 0.0, 1.0 if y_is_real else 0.0, float(weight))
