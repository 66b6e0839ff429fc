"""Glue between the drop-in nn.Modules and the recorded HIP launch plans."""
import torch

from . import engine as E
from . import _lib as L


def _draw_seed():
    # torch's CPU generator -> reproducible under torch.manual_seed (utils/util.py:43-47)
    return int(torch.randint(0, 2 ** 62, (1,)).item())


def _needs_grad(module, x):
    if not torch.is_grad_enabled():
        return False
    return x.requires_grad or any(p.requires_grad for p in module.parameters())


def _prep_input(x, what):
    E.require_cuda(x, what)
    if x.dim() != 4:
        raise ValueError('%s must be NCHW, got shape %s' % (what, tuple(x.shape)))
    return x.detach().contiguous().float()


def _zs_list(z, n, shape, device):
    if z is None:
        return None
    zs = list(z) if isinstance(z, (list, tuple)) else [z]
    if len(zs) != n:
        raise ValueError('expected %d noise tensors, got %d' % (n, len(zs)))
    out = []
    for t in zs:
        E.require_cuda(t, 'noise tensor z')
        if tuple(t.shape) != tuple(shape):
            raise ValueError('noise tensor shape %s != %s' % (tuple(t.shape), tuple(shape)))
        out.append(t.detach().contiguous().float())
    return out


def run_block(mod, kind, x, z=None):
    """ResidualDenseBlock_5C / RRDB forward (block.py:260-268, 287-291) on the HIP path."""
    if _needs_grad(mod, x):
        raise NotImplementedError('autograd through a stand-alone %s is not wired yet; use '
                                  'torch.no_grad() or the full RRDBNet' % kind)
    xin = _prep_input(x, 'input')
    B, C_, H, W = xin.shape
    if C_ != 64:
        raise ValueError('expected 64 input channels, got %d' % C_)
    wp = mod._weights(xin.device)
    has_noise = getattr(mod, 'noise', None) is not None if kind == 'rdb' else True
    noise = bool(mod.training and has_noise)
    n_noise = 0 if not noise else (1 if kind == 'rdb' else (4 if mod.variant == 'test_image' else 3))
    zs = _zs_list(z, n_noise, (B, 64, H, W), xin.device) if noise else None
    key = (kind, B, H, W, mod.precision, noise, zs is not None, wp.generation)
    plan = mod._plans.get(key)
    if plan is None:
        plan = E.build_block_plan(kind, wp, B, H, W, mod.precision, xin.device, noise,
                                  mod.variant, zs is not None)
        mod._plans = {key: plan}
    out = torch.empty(plan.out_shape, dtype=torch.float32, device=xin.device)
    plan.run(xin, out, E.current_stream(), _draw_seed() if (noise and zs is None) else 0, zs)
    return out


def run_rrdbnet(net, x, z=None):
    """RRDBNet.forward (architecture.py:76-78) on the HIP path."""
    if _needs_grad(net, x):
        raise NotImplementedError('RRDBNet backward is not wired yet; call under torch.no_grad()')
    xin = _prep_input(x, 'input')
    B, C_, H, W = xin.shape
    if C_ != net.in_nc:
        raise ValueError('expected %d input channels, got %d' % (net.in_nc, C_))
    wp = net._weights(xin.device)
    noise = bool(net.training)
    per = 4 if net.variant == 'test_image' else 3
    zs = _zs_list(z, per * net.nb, (B, 64, H, W), xin.device) if noise else None
    key = (B, H, W, net.precision, noise, zs is not None, wp.generation)
    plan = net._plans.get(key)
    if plan is None:
        if len(net._plans) >= net.max_cached_plans:
            net._plans.clear()
        plan = E.build_rrdbnet_plan(wp, net.nb, net.in_nc, net.out_nc, B, H, W, net.precision,
                                    xin.device, noise, net.variant, zs is not None)
        net._plans[key] = plan
    out = torch.empty(plan.out_shape, dtype=torch.float32, device=xin.device)
    plan.run(xin, out, E.current_stream(), _draw_seed() if (noise and zs is None) else 0, zs)
    return out
