"""Drop-in for the reference's ``block`` module on the ESRGAN+ hot path
(codes/models/modules/block.py and test_image/block.py).

Same factory names, constructor arguments, module tree and state-dict keys as the reference, so
``networks.init_weights`` (networks.py:30-74, matches class names 'Conv'/'BatchNorm2d'), optimizers,
``load_state_dict`` and checkpoints keep working.  The leaf ``nn.Conv2d`` / ``nn.LeakyReLU`` /
``nn.Upsample`` modules are *parameter holders only*: arithmetic never runs through them.  The
composite modules (``ResidualDenseBlock_5C``, ``RRDB``, and ``RRDBNet`` in architecture.py) execute
as fused HIP launch plans (engine.py -> libesrgan_hip.so) and raise on CPU tensors.

``ResNetBlock`` and ``pixelshuffle_block`` (the SRResNet siblings) are chains of ``Conv2dHIP``
one-layer plans further down.  Out of scope exactly as SURVEY.md §2.1 row 1 marks them: ConcatBlock,
minibatch_std_concat_layer, reflect/replicate padding, prelu, instance norm, NAC / CNAC modes.
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from . import engine as E
from . import _lib as L


def act(act_type, inplace=True, neg_slope=0.2, n_prelu=1):
    """block.py:12-25 (leakyrelu / relu on this path)."""
    t = act_type.lower()
    if t == 'relu':
        return nn.ReLU(inplace)
    if t == 'leakyrelu':
        return nn.LeakyReLU(neg_slope, inplace)
    raise NotImplementedError('activation layer [{:s}] is not found'.format(t))


def norm(norm_type, nc):
    """block.py:28-37 (batch norm only: Discriminator_VGG_128)."""
    t = norm_type.lower()
    if t == 'batch':
        return nn.BatchNorm2d(nc, affine=True)
    raise NotImplementedError('normalization layer [{:s}] is not found'.format(t))


def pad(pad_type, padding):
    """block.py:40-52: only conv-internal zero padding exists on the hot path."""
    if padding == 0 or pad_type.lower() == 'zero':
        return None
    raise NotImplementedError('padding layer [{:s}] is not implemented'.format(pad_type.lower()))


def get_valid_padding(kernel_size, dilation):
    """block.py:55-58."""
    kernel_size = kernel_size + (kernel_size - 1) * (dilation - 1)
    return (kernel_size - 1) // 2


def sequential(*args):
    """block.py:95-108: flattening Sequential (fixes the ``model.N`` key numbering)."""
    if len(args) == 1:
        if isinstance(args[0], OrderedDict):
            raise NotImplementedError('sequential does not support OrderedDict input.')
        return args[0]
    mods = []
    for m in args:
        if isinstance(m, nn.Sequential):
            mods.extend(m.children())
        elif isinstance(m, nn.Module):
            mods.append(m)
    return nn.Sequential(*mods)


def conv_block(in_nc, out_nc, kernel_size, stride=1, dilation=1, groups=1, bias=True,
               pad_type='zero', norm_type=None, act_type='relu', mode='CNA', hip=False):
    """block.py:125-151 — Conv(zero pad (k-1)//2) -> Norm -> Act holder chain ('CNA' only).
    hip=True: the conv is a ``Conv2dHIP`` (runs by itself on the HIP kernels, forward and backward) — what the
    module families that are NOT planned as a whole (SRResNet, pixelshuffle_block) are built from; the planned
    networks (RRDBNet, the discriminators) only read ``.weight`` / ``.bias`` of plain nn.Conv2d holders."""
    if mode != 'CNA':
        raise NotImplementedError('conv mode [{:s}] is outside the ESRGAN+ hot path'.format(mode))
    if dilation != 1 or groups != 1:
        raise NotImplementedError('dilation/groups are outside the ESRGAN+ hot path')
    p = pad(pad_type, get_valid_padding(kernel_size, dilation)) if pad_type else None
    if hip:
        from .architecture import Conv2dHIP
        c = Conv2dHIP(in_nc, out_nc, kernel_size=kernel_size, stride=stride,
                      padding=get_valid_padding(kernel_size, dilation), bias=bias)
    else:
        c = nn.Conv2d(in_nc, out_nc, kernel_size=kernel_size, stride=stride,
                      padding=get_valid_padding(kernel_size, dilation), dilation=1, bias=bias, groups=1)
    a = act(act_type) if act_type else None
    n = norm(norm_type, out_nc) if norm_type else None
    return sequential(p, c, n, a)


def conv1x1(in_planes, out_planes, stride=1):
    """block.py:153-154 (bias-free)."""
    return nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=False)


def upconv_blcok(in_nc, out_nc, upscale_factor=2, kernel_size=3, stride=1, bias=True,
                 pad_type='zero', norm_type=None, act_type='relu', mode='nearest', hip=False):
    """block.py:315-322 (sic: reference spelling)."""
    if upscale_factor != 2 or mode != 'nearest':
        raise NotImplementedError('only nearest x2 upconv is on the ESRGAN+ hot path')
    up = nn.Upsample(scale_factor=upscale_factor, mode=mode)
    return sequential(up, conv_block(in_nc, out_nc, kernel_size, stride, bias=bias, pad_type=pad_type,
                                     norm_type=norm_type, act_type=act_type, hip=hip))


def pixelshuffle_block(in_nc, out_nc, upscale_factor=2, kernel_size=3, stride=1, bias=True,
                       pad_type='zero', norm_type=None, act_type='relu', hip=True):
    """block.py:299-312: conv to out_nc * r^2 channels (HIP conv, ``Conv2dHIP``) -> nn.PixelShuffle(r) -> act.
    Keys as the reference: ``0.weight`` / ``0.bias`` (the shuffle and the activation hold no parameters).
    hip=False: parameter holders only (inside a network that is planned as a whole: architecture.SRResNet)."""
    if norm_type:
        raise NotImplementedError('pixelshuffle_block with a norm layer is not used by the reference configs')
    conv = conv_block(in_nc, out_nc * (upscale_factor ** 2), kernel_size, stride, bias=bias, pad_type=pad_type,
                      norm_type=None, act_type=None, hip=hip)
    return sequential(conv, nn.PixelShuffle(upscale_factor), act(act_type) if act_type else None)


class ResNetBlock(nn.Module):
    """block.py:199-232: x + res_scale * conv1(act(conv0(x))) ('CNA', no norm: train_SRResNet.json:39-41), both
    convs on the HIP kernels (``Conv2dHIP``)."""

    def __init__(self, in_nc, mid_nc, out_nc, kernel_size=3, stride=1, dilation=1, groups=1, bias=True,
                 pad_type='zero', norm_type=None, act_type='relu', mode='CNA', res_scale=1, hip=True):
        super().__init__()
        if norm_type or mode != 'CNA':
            raise NotImplementedError('HIP ResNetBlock: mode CNA without a norm layer (train_SRResNet.json:40-41)')
        # hip=False: parameter holders only (inside a network that is planned as a whole: architecture.SRResNet)
        conv0 = conv_block(in_nc, mid_nc, kernel_size, stride, dilation, groups, bias, pad_type, None, act_type,
                           mode, hip=hip)
        conv1 = conv_block(mid_nc, out_nc, kernel_size, stride, dilation, groups, bias, pad_type, None, None,
                           mode, hip=hip)
        self.res = sequential(conv0, conv1)
        self.res_scale = res_scale

    def forward(self, x):
        return x + self.res(x).mul(self.res_scale)


class GaussianNoise(nn.Module):
    """block.py:110-122.  Holds no parameters/buffers (empty state dict, like the reference); the
    multiplicative noise x*(1+sigma*z) itself is a fused conv epilogue.  Unlike the reference it
    does not pin a tensor to device 0 at construction, so replicas on any device work."""

    def __init__(self, sigma=0.1, is_relative_detach=False):
        super().__init__()
        if is_relative_detach:
            raise NotImplementedError('is_relative_detach=True is never used by the reference')
        self.sigma = sigma
        self.is_relative_detach = is_relative_detach

    def forward(self, x):
        raise L.HipExtensionError('GaussianNoise is fused into its producer conv; call the '
                                  'enclosing ResidualDenseBlock_5C / RRDB / RRDBNet instead')


class ShortcutBlock(nn.Module):
    """block.py:78-92 — x + sub(x); executed as the LR_conv epilogue inside RRDBNet's plan."""

    def __init__(self, submodule):
        super().__init__()
        self.sub = submodule

    def forward(self, x):
        # inside RRDBNet this module is never called (the skip is LR_conv's epilogue).  Called directly it runs
        # x + sub(x) only when every conv below is a HIP module (SRResNet); plain nn.Conv2d holders would fall back
        # to torch's convolution silently — refused.
        ok = self.__dict__.get('_hip_only')
        if ok is None:
            from .architecture import Conv2dHIP
            ok = all(isinstance(m, Conv2dHIP) for m in self.sub.modules() if isinstance(m, nn.Conv2d))
            self.__dict__['_hip_only'] = ok
        if not ok:
            raise L.HipExtensionError('ShortcutBlock is fused into RRDBNet.forward; call the network')
        return x + self.sub(x)

    def __repr__(self):
        return 'Identity + \n|' + self.sub.__repr__().replace('\n', '\n|')


class _PlannedModule(nn.Module):
    """Shared machinery: weight pack + per-shape launch plans + autograd hookup."""

    variant = 'codes'

    def _init_planned(self):
        self._wp = {}
        self._plans = {}
        self.precision = 'fp32'       # 'fp32' (exact path, reference numerics) or 'fp16'
        self._force_repack = True

    def weights_unchanged(self):
        """Context manager: the caller vouches that no parameter changed since this module's last forward (the D step
        of the train loop runs right after the G step's D pass, before any optimizer step) — the forced re-pack of
        training-mode modules is skipped; a changed storage or version still re-packs."""
        mod = self

        class _Ctx(object):
            def __enter__(self):
                object.__setattr__(mod, '_weights_clean', True)

            def __exit__(self, *a):
                object.__setattr__(mod, '_weights_clean', False)
        return _Ctx()

    # ---- work a training loop left in flight on ANOTHER stream (train.ESRGANPlusStep's pipelined form keeps the end of
    # the D step, D's Adam and the weight packs on its side stream): the public entry points order it in front of the
    # caller's stream, so a validation forward / a checkpoint between two pipelined steps sees finished weights ----
    def _defer_to(self, event):
        self.__dict__['_pending_ev'] = event

    def _join_pending(self):
        # The event STAYS until the next step replaces it: a second forward / state_dict on ANOTHER stream has to be
        # ordered behind the same tail (round 5 popped it: only the first caller's stream was).  A stream waits once.
        pend = self.__dict__.get('_pending_ev')
        if pend is None:
            return
        if not isinstance(pend, tuple):
            pend = self.__dict__['_pending_ev'] = (pend, set())
        ev, joined = pend
        cur = torch.cuda.current_stream()
        key = (cur.device.index, cur.cuda_stream)
        if key not in joined:
            cur.wait_event(ev)
            joined.add(key)

    # (nn.DataParallel's replicate() broadcasts the weights on the caller's stream before any hook of a module runs, and
    # `parameters()` is on the train step's own fast path, so it cannot join here: a loop that hands the networks of a
    # PIPELINED step to DataParallel calls ESRGANPlusStep.finish() first.  Replicas keep the event: their forwards wait.)

    def state_dict(self, *a, **k):
        self._join_pending()
        return super().state_dict(*a, **k)

    def set_precision(self, precision):
        """'fp32': v_mfma_f32_32x32x2_f32 (bitwise an fp32 fma chain) — the <=1e-3 parity path.
        'fp16': fp16 storage + v_mfma_f32_32x32x16_f16 with fp32 accumulation — the fast path."""
        if precision not in ('fp16', 'fp32'):
            raise ValueError(precision)
        self.precision = precision
        return self

    def invalidate(self):
        """Force a weight re-pack on the next forward (needed after ``p.data`` surgery that does
        not bump the parameter version, e.g. networks.py:32-34)."""
        self._force_repack = True

    def train(self, mode=True):
        # optim.FusedAdam updates the parameters through raw pointers (no `_version` bump): the packed
        # copies made by the last training forward are one optimizer step stale, so a train <-> eval
        # flip always re-packs before the next forward.
        if mode != self.training:
            self._force_repack = True
        return super().train(mode)

    # nn.Module hooks that change parameter values behind our back
    def apply(self, fn):
        self._force_repack = True
        return super().apply(fn)

    def _apply(self, fn, *a, **k):
        self._force_repack = True
        self.__dict__['_conv_cache'] = None
        return super()._apply(fn, *a, **k)

    def _replicate_for_data_parallel(self):
        # nn.DataParallel (networks.py:105-107) shallow-copies __dict__ and then swaps in per-device
        # parameter copies: anything derived from the ORIGINAL's parameters must not leak into the replica
        replica = super()._replicate_for_data_parallel()
        replica.__dict__['_conv_cache'] = None
        replica.__dict__['_gstore'] = None
        replica.__dict__.pop('_grad_proxy', None)     # its weights are non-leaf copies: per-tensor gradient outputs
        replica.__dict__['_wp'] = {}
        replica.__dict__['_plans'] = {}
        replica.__dict__['_force_repack'] = True
        return replica

    def _convs(self):
        """Cached (key, weight, bias) list + the flat parameter list autograd sees (rebuilding them
        walks ~350 modules: 1.8 ms of host time per training step)."""
        c = self.__dict__.get('_conv_cache')
        if c is None:
            lst = self._conv_list()
            flat = []
            for _, w, b in lst:
                flat.append(w)
                if b is not None:
                    flat.append(b)
            c = (lst, flat)
            self.__dict__['_conv_cache'] = c
        return c

    def load_state_dict(self, *a, **k):
        self._force_repack = True
        return super().load_state_dict(*a, **k)

    # ---- module-owned gradient store (the fused backward of a whole network emits ALL parameter gradients as one
    # flat fp32 buffer: handing ~770 tensors through autograd one by one — Function inputs, AccumulateGrad nodes,
    # fresh views — cost ~5 ms of host time per training step, more than enqueueing every kernel of the step) ----
    flat_param_grads = False      # RRDBNet: True

    def _grad_store(self, device):
        """Persistent flat gradient buffer + one cached view per parameter, in `_convs()` order."""
        gs = self.__dict__.get('_gstore')
        flat_params = self._convs()[1]
        if gs is None or gs['dev'] != device or gs['params'] is not flat_params:
            sizes = [p.numel() for p in flat_params]
            flat = torch.zeros(sum(sizes), dtype=torch.float32, device=device)
            views = [t.view(p.shape) for t, p in zip(flat.split(sizes), flat_params)]
            gs = dict(dev=device, params=flat_params, sizes=sizes, flat=flat, views=views, stale=False)
            self.__dict__['_gstore'] = gs
        return gs

    def _deliver_flat_grads(self, flat_new, adopt=False):
        """What AccumulateGrad does for every parameter, on the flat buffer: `.grad` of every parameter becomes (stays) its
        view of the module's store, which receives `flat_new` — copied when the gradients were None (zero_grad
        (set_to_none=True)) or marked stale, added otherwise (autograd's accumulation).  Gradients somebody else put in
        place are honoured tensor by tensor.
        adopt=True (the hand-written training loops, which own the gradients and consume them before the plan's next
        backward: train.ESRGANPlusStep, bench.py's generator loop): when the delivery would OVERWRITE — gradients None
        or marked stale — `.grad` becomes views of `flat_new` itself (the plan's buffer, cached per buffer) and nothing
        is copied (67 MB per generator step); an accumulating delivery takes the copying route.
        Adopted views ALIAS the plan's buffer, which the plan's next backward rewrites: whoever is about to rewrite it
        calls `_release_adopted(buf)` first (functional.py does), so an accumulating delivery finds the old values in
        the module's own store; every branch below leaves `.grad` on the store's views unless it adopts."""
        gs = self._grad_store(flat_new.device)
        params, views = gs['params'], gs['views']
        ad = gs.get('adopted')
        on_store = all(p.grad is v for p, v in zip(params, views))
        on_adopted = (not on_store) and ad is not None and all(p.grad is v for p, v in zip(params, ad[1]))
        none = (not on_store) and (not on_adopted) and all(p.grad is None for p in params)

        def to_store():
            for p_, v in zip(params, views):
                p_.grad = v

        if on_adopted and not gs['stale']:
            # accumulate onto gradients that live in a plan's buffer
            if ad[0] is flat_new or ad[0].data_ptr() == flat_new.data_ptr():
                raise RuntimeError(
                    'gradient accumulation onto a plan buffer that its own backward has just rewritten: the earlier '
                    'gradients are gone (call net._release_adopted(buf) before the backward overwrites the buffer — '
                    'functional.rrdbnet_train_backward / _RRDBNetFn do — or zero_grad / mark_grads_stale between steps)')
            torch.add(ad[0], flat_new, out=gs['flat'])
            to_store()
            return
        if adopt and (none or ((on_store or on_adopted) and gs['stale'])):
            if ad is None or ad[0] is not flat_new:
                ad = gs['adopted'] = (flat_new, [t.view(p.shape) for t, p in zip(flat_new.split(gs['sizes']), params)])
            if not all(p.grad is v for p, v in zip(params, ad[1])):
                for p_, v in zip(params, ad[1]):
                    p_.grad = v
            gs['stale'] = False
            return
        if on_adopted:          # stale, and the caller does not vouch for the buffer's lifetime: overwrite the STORE
            gs['flat'].copy_(flat_new)
            gs['stale'] = False
            to_store()
            return
        if on_store:
            if gs['stale']:
                gs['flat'].copy_(flat_new)
                gs['stale'] = False
            else:
                gs['flat'].add_(flat_new)
            return
        if none:
            gs['flat'].copy_(flat_new)
            gs['stale'] = False
            to_store()
            return
        new = flat_new.clone().split(gs['sizes'])
        for p_, g, v in zip(params, new, views):
            g = g.view(v.shape)
            p_.grad = g if p_.grad is None else p_.grad + g

    def _release_adopted(self, buf=None):
        """Called by whoever is about to REWRITE a plan's flat gradient buffer (the zero fill in front of a backward):
        when the parameters' `.grad` are adopted views of that buffer (`buf=None`: of any buffer) and hold gradients
        that still count — not marked stale — they move into the module's own store first, so that the coming delivery
        accumulates onto them (a second backward without zero_grad, zero_grad(set_to_none=False), a loop that falls
        back from the hand-written step to autograd).  Stale gradients stay where they are: the delivery overwrites."""
        gs = self.__dict__.get('_gstore')
        if gs is None:
            return False
        ad = gs.get('adopted')
        if ad is None or (buf is not None and ad[0] is not buf and ad[0].data_ptr() != buf.data_ptr()):
            return False
        if gs['stale'] or not all(p.grad is v for p, v in zip(gs['params'], ad[1])):
            return False
        gs['flat'].copy_(ad[0])
        for p_, v in zip(gs['params'], gs['views']):
            p_.grad = v
        return True

    def mark_grads_stale(self):
        """Cheap stand-in for ``zero_grad`` in a training loop that owns this module's gradients (train.ESRGANPlusStep):
        the next backward OVERWRITES the store instead of adding to it; `.grad` keeps pointing at the (old) values
        until then.  Returns False — nothing marked — when the parameters' gradients are not the store's views."""
        gs = self.__dict__.get('_gstore')
        if gs is None:
            return False
        ad = gs.get('adopted')
        if not (all(p.grad is v for p, v in zip(gs['params'], gs['views']))
                or (ad is not None and all(p.grad is v for p, v in zip(gs['params'], ad[1])))):
            return False
        gs['stale'] = True
        return True

    def _flush_stale_grads(self):
        """A backward through the per-tensor autograd route ADDS into `.grad` (AccumulateGrad): a store that was only
        marked stale — `mark_grads_stale()` stood in for `zero_grad()` — must really be zeroed first, or the old
        gradients pile up (a parameter frozen after the first step, `flat_param_grads` switched off).  Adopted views
        are given up here as well: that backward rewrites the plan's buffer BEFORE AccumulateGrad adds to `.grad`, so
        `.grad` must not alias it — the gradients move to (stale: are zeroed in) the module's own store."""
        gs = self.__dict__.get('_gstore')
        if gs is None:
            return
        ad = gs.get('adopted')
        if ad is not None and all(p.grad is v for p, v in zip(gs['params'], ad[1])):
            if gs['stale']:
                ad[0].zero_()                     # (the stale gradients live in an adopted plan buffer)
                gs['flat'].zero_()
            else:
                gs['flat'].copy_(ad[0])
            for p_, v in zip(gs['params'], gs['views']):
                p_.grad = v
            gs['stale'] = False
            return
        if gs['stale']:
            gs['flat'].zero_()
            gs['stale'] = False

    def _conv_list(self):
        raise NotImplementedError

    def _dgrad_special(self):
        return {}

    def _dgrad_extra(self, device):
        """Constant operands of the backward plan that are not parameters."""
        return []

    def _dgrad_gathers(self):
        return []

    def _eye_operand(self, device):
        # 1x1 identity kernel: carries the incoming gradient of a stand-alone block through the conv
        # epilogue's noise / scale stages (engine.build_rrdbnet_train_plan, kind 'rdb' / 'rrdb')
        eye = getattr(self, '_eye64', None)
        if eye is None or eye.device != torch.device(device):
            eye = torch.eye(64, device=device).view(64, 64, 1, 1).contiguous()
            self._eye64 = eye
        return [('__eye', eye)]

    def _new_dgrad_pack(self, device):
        # dense-block convs get gather-form operands (one per channel slice); only the other
        # convs (head / tail of the generator) are packed as plain transposes
        convs = [(k, w) for k, w, _ in self._conv_list()
                 if 'RDB' not in k and not k.startswith('rdb')] + self._dgrad_extra(device)
        return E.DgradPack(convs, self.precision, device, self._dgrad_special(), self._dgrad_gathers())

    def _dgrad_weights(self, device):
        key = ('dgrad', self.precision, str(device))
        dp = self._wp.get(key)
        if dp is None:
            dp = self._wp[key] = self._new_dgrad_pack(device)
        return dp

    def _subpix_keys(self):
        """Keys of up-convs to run in the sub-pixel form (engine.WeightPack)."""
        return ()

    def prepack(self, fwd=True, dgrad=True):
        """Re-pack the kernel-side weight copies NOW, on the current stream (a training loop calls this right after
        ``optimizer.step()``, on a stream that runs next to other work): the next training forward then finds them
        fresh instead of packing at its start.  Only valid when nothing changes the parameters in between except
        through paths this module sees (load_state_dict / apply / .to set the force flag; in-place edits bump
        ``_version``).  No-op for packs that do not exist yet."""
        st = E.current_stream()
        for key, pk in list(self._wp.items()):
            if key[0] == 'dgrad':
                if dgrad and key[1] == self.precision:
                    pk.ensure(st, force=True, record_sig=True)
                    self.__dict__['_prepacked_dgrad'] = True
            elif fwd and key[0] == self.precision:
                pk.ensure(st, force=True, record_sig=True)
                self.__dict__['_prepacked_fwd'] = True

    def _dgrad_fresh(self):
        """True once after prepack(dgrad=True): the caller may skip its forced dgrad re-pack."""
        return self.__dict__.pop('_prepacked_dgrad', False)

    def _weights(self, device):
        key = (self.precision, str(device))
        wp = self._wp.get(key)
        if wp is None:
            wp = E.WeightPack(self._conv_list(), self.precision, device, self._subpix_keys())
            self._wp[key] = wp
        clean = self.__dict__.get('_weights_clean', False) or self.__dict__.pop('_prepacked_fwd', False)
        wp.ensure(E.current_stream(), force=self._force_repack or (self.training and not clean),
                  full=self._force_repack)
        self._force_repack = False
        return wp


def _rdb_gathers(prefix, m):
    """Gather-form input-gradient operands of one dense block (engine.DgradPack): the gradient of
    channel slice j (x4, x3, x2, x1, x) is ONE conv over Q = [g_t | g_a4 | g_a3 | g_a2 | g_a1 | g_x2]
    (the pre-activation gradients of the LATER convs, in that channel order; g_t stands for g_a5 / 0.2,
    hence the 0.2 on conv5's piece; g_x2, un-masked, feeds the transposed 1x1 at the centre tap)."""
    w = [None] + [getattr(m, 'conv%d' % k)[0].weight for k in range(1, 6)]
    later = lambda j: [(w[5], 0.2)] + [(w[k], 1.0) for k in range(4, j, -1)]
    out = []
    for j, co0 in ((4, 160), (3, 128), (2, 96), (1, 64)):
        out.append((prefix + '.g%d' % j, 32, [(t, co0, sc) for t, sc in later(j)]))
    out.append((prefix + '.g0', 64, [(t, 0, sc) for t, sc in later(0)] + [(m.conv1x1.weight, 0, 1.0)]))
    # operands of the fused backward chain (esr_rdb_backward, csrc/rdb_chain_kernel.h) where they differ from the
    # per-conv launches': the x2 slice with the identity path x4 = lrelu(a4) + x2 (block.py:266) folded into conv5's
    # piece (its x4 columns added to its x2 columns: no residual read of g_x4), the x slice without the 1x1 (K = 192),
    # and the transposed 1x1 in the K order of the chain's epilogue registers (esr_pack.one_t)
    out.append((prefix + '.c2', 32, [(w[5], 96, 0.2, 160)] + [(w[k], 96, 1.0) for k in (4, 3)]))
    out.append((prefix + '.c0', 64, [(t, 0, sc) for t, sc in later(0)]))
    out.append((prefix + '.o1', 'one_t', m.conv1x1.weight))
    return out


def _rdb_convs(prefix, m):
    out = [(prefix + '.conv1x1', m.conv1x1.weight, None)]
    for k in range(1, 6):
        c = getattr(m, 'conv%d' % k)[0]
        out.append((prefix + '.conv%d.0' % k, c.weight, c.bias))
    return out


class ResidualDenseBlock_5C(_PlannedModule):
    """block.py:232-268 (``gaussian_noise``) / test_image/block.py:195-232 (``noise_input``)."""

    def __init__(self, nc, kernel_size=3, gc=32, stride=1, bias=True, pad_type='zero',
                 norm_type=None, act_type='leakyrelu', mode='CNA', gaussian_noise=True,
                 noise_input=None):
        super().__init__()
        if noise_input is not None:
            gaussian_noise = noise_input
        if (nc, kernel_size, gc, stride, bias, norm_type, act_type.lower(), mode) != \
                (64, 3, 32, 1, True, None, 'leakyrelu', 'CNA'):
            raise NotImplementedError('HIP ResidualDenseBlock_5C supports the ESRGAN+ configuration '
                                      'nc=64, gc=32, 3x3, leakyrelu, CNA')
        self.noise = GaussianNoise() if gaussian_noise else None
        self.conv1x1 = conv1x1(nc, gc)
        for k in range(1, 5):
            setattr(self, 'conv%d' % k, conv_block(nc + (k - 1) * gc, gc, kernel_size, stride,
                                                   bias=bias, pad_type=pad_type, norm_type=norm_type,
                                                   act_type=act_type, mode=mode))
        self.conv5 = conv_block(nc + 4 * gc, nc, 3, stride, bias=bias, pad_type=pad_type,
                                norm_type=norm_type, act_type=None, mode=mode)
        self._init_planned()

    def _conv_list(self):
        return _rdb_convs('rdb', self)

    def _dgrad_gathers(self):
        return _rdb_gathers('rdb', self)

    def _dgrad_extra(self, device):
        return self._eye_operand(device)

    def forward(self, x, z=None):
        from .functional import run_block
        return run_block(self, 'rdb', x, z)


class RRDB(_PlannedModule):
    """block.py:271-291; ``extra_noise`` selects the test_image/block.py:250,256 variant."""

    def __init__(self, nc, kernel_size=3, gc=32, stride=1, bias=True, pad_type='zero',
                 norm_type=None, act_type='leakyrelu', mode='CNA', extra_noise=False):
        super().__init__()
        for j in (1, 2, 3):
            setattr(self, 'RDB%d' % j, ResidualDenseBlock_5C(nc, kernel_size, gc, stride, bias,
                                                             pad_type, norm_type, act_type, mode))
        if extra_noise:
            self.noise = GaussianNoise()
            self.variant = 'test_image'
        self._init_planned()

    def _conv_list(self):
        out = []
        for j in (1, 2, 3):
            out += _rdb_convs('rrdb.RDB%d' % j, getattr(self, 'RDB%d' % j))
        return out

    def _dgrad_gathers(self):
        out = []
        for j in (1, 2, 3):
            out += _rdb_gathers('rrdb.RDB%d' % j, getattr(self, 'RDB%d' % j))
        return out

    def _dgrad_extra(self, device):
        return self._eye_operand(device)

    def forward(self, x, z=None):
        from .functional import run_block
        return run_block(self, 'rrdb', x, z)
