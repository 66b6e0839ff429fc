"""Drop-in for the reference's ``architecture`` module on the ESRGAN+ hot path:
``RRDBNet`` (codes/models/modules/architecture.py:47-78) and ``RRDB_Net``
(test_image/architecture.py:7-38).  Same constructor signatures, module tree and state-dict keys
(SURVEY.md Appendix B); ``forward`` replays a fused HIP launch plan instead of walking the
``nn.Sequential``.  ``SRResNet`` (architecture.py:13-44) is a chain of per-conv HIP modules (``Conv2dHIP``).
"""
import os
import math

import torch.nn as nn

from . import block as B
from .functional import run_rrdbnet


class _RRDBNetBase(B._PlannedModule):
    # parameter gradients leave the backward node through the module's flat store (block._PlannedModule._grad_store)
    # instead of ~770 autograd outputs; ESR_FLAT_GRADS=0 or `net.flat_param_grads = False` restores the per-tensor
    # route (needed for torch.autograd.grad(y, params) and tensor hooks on the parameters)
    flat_param_grads = os.environ.get('ESR_FLAT_GRADS', '1') != '0'

    def _build(self, in_nc, out_nc, nf, nb, upscale, norm_type, act_type, mode, upsample_mode,
               extra_noise):
        if upsample_mode == 'pixelshuffle':
            raise NotImplementedError('the planned RRDBNet uses upconv (networks.py:99); pixelshuffle_block itself '
                                      'is available (block.pixelshuffle_block, SRResNet)')
        if upsample_mode != 'upconv':
            raise NotImplementedError('upsample mode [{:s}] is not found'.format(upsample_mode))
        if (nf, upscale, norm_type, act_type.lower(), mode) != (64, 4, None, 'leakyrelu', 'CNA'):
            raise NotImplementedError('HIP RRDBNet supports the ESRGAN+ configuration nf=64, x4, '
                                      'no norm, leakyrelu, CNA (train_ESRGANplus.json:36-45)')
        self.in_nc, self.out_nc, self.nb = in_nc, out_nc, nb
        n_up = int(math.log(upscale, 2))
        fea_conv = B.conv_block(in_nc, nf, kernel_size=3, norm_type=None, act_type=None)
        # NB: the reference ignores its ``gc`` argument and always builds gc=32
        # (architecture.py:56) — so do we.
        blocks = [B.RRDB(nf, kernel_size=3, gc=32, stride=1, bias=True, pad_type='zero',
                         norm_type=norm_type, act_type=act_type, mode='CNA',
                         extra_noise=extra_noise) for _ in range(nb)]
        lr_conv = B.conv_block(nf, nf, kernel_size=3, norm_type=norm_type, act_type=None, mode=mode)
        ups = [B.upconv_blcok(nf, nf, act_type=act_type) for _ in range(n_up)]
        hr0 = B.conv_block(nf, nf, kernel_size=3, norm_type=None, act_type=act_type)
        hr1 = B.conv_block(nf, out_nc, kernel_size=3, norm_type=None, act_type=None)
        self.model = B.sequential(fea_conv, B.ShortcutBlock(B.sequential(*blocks, lr_conv)),
                                  *ups, hr0, hr1)
        self._init_planned()
        self.max_cached_plans = 4
        self.variant = 'test_image' if extra_noise else 'codes'

    def _conv_list(self):
        m, nb = self.model, self.nb
        out = [('model.0', m[0].weight, m[0].bias)]
        for i in range(nb):
            rr = m[1].sub[i]
            for j in (1, 2, 3):
                out += B._rdb_convs('model.1.sub.%d.RDB%d' % (i, j), getattr(rr, 'RDB%d' % j))
        lr = m[1].sub[nb]
        out.append(('model.1.sub.%d' % nb, lr.weight, lr.bias))
        for idx in (3, 6, 8, 10):
            out.append(('model.%d' % idx, m[idx].weight, m[idx].bias))
        return out

    def attach_grad_sync(self, sync):
        """Data-parallel hook (dp.GradExchange): `sync(flat_slice)` starts the mean all-reduce of a finished
        slice of the backward's flat gradient buffer; the backward then runs segmented (no graph replay)."""
        self._grad_sync = sync

    def _subpix_keys(self):
        # upconv_blcok (block.py:315-322) in its 4-phase 2x2 form: 16 instead of 36 MACs per 4 outputs
        return () if os.environ.get('ESR_SUBPIX', '1') == '0' else ('model.3', 'model.6')

    def _dgrad_special(self):
        return {'model.3': {'ups': True}, 'model.6': {'ups': True}}

    def _dgrad_extra(self, device):
        return self._eye_operand(device)

    def _dgrad_gathers(self):
        out = []
        for i in range(self.nb):
            rr = self.model[1].sub[i]
            for j in (1, 2, 3):
                out += B._rdb_gathers('model.1.sub.%d.RDB%d' % (i, j), getattr(rr, 'RDB%d' % j))
        return out

    def forward(self, x, z=None):
        """x: NCHW float32 in [0,1] on the MI355X -> [B, out_nc, 4H, 4W] float32.
        ``z`` (training mode only): explicit N(0,1) tensors, one [B,64,H,W] per noise layer in
        execution order, for bit-parity tests; default = fused Philox stream."""
        self._join_pending()
        return run_rrdbnet(self, x, z)


class RRDBNet(_RRDBNetBase):
    """codes/models/modules/architecture.py:47-78."""

    def __init__(self, in_nc, out_nc, nf, nb, gc=32, upscale=4, norm_type=None,
                 act_type='leakyrelu', mode='CNA', upsample_mode='upconv'):
        super().__init__()
        self._build(in_nc, out_nc, nf, nb, upscale, norm_type, act_type, mode, upsample_mode, False)


class RRDB_Net(_RRDBNetBase):
    """test_image/architecture.py:7-38 (inference copy: extra noise layer per RRDB in train
    mode, test_image/block.py:250,256; unused ``res_scale``)."""

    def __init__(self, in_nc, out_nc, nf, nb, gc=32, upscale=4, norm_type=None,
                 act_type='leakyrelu', mode='CNA', res_scale=1, upsample_mode='upconv'):
        super().__init__()
        self._build(in_nc, out_nc, nf, nb, upscale, norm_type, act_type, mode, upsample_mode, True)


# =================================================================================================
# Discriminator_VGG_128 and VGGFeatureExtractor (the rest of the ESRGAN+ train step)
# =================================================================================================
import torch  # noqa: E402

from . import _lib as L  # noqa: E402
from . import convnet as CN  # noqa: E402
from . import engine as E  # noqa: E402


class _SeqNet(B._PlannedModule):
    """Shared forward/backward driver of the feed-forward plans."""

    def _spec(self):
        raise NotImplementedError

    def _head(self):
        return None

    def _input_affine(self):
        return None

    def _pspec(self):
        raise NotImplementedError

    def _dgrad_special(self):
        return {s['conv']: {'ts2': True} for s in self._spec() if 'conv' in s and s['stride'] == 2}

    def _wants_param_grads(self):
        # by attribute access (`_pspec`), not `parameters()`: that is empty on a nn.DataParallel replica
        return any(t.requires_grad for _, t in self._pspec())

    def _run_forward(self, x, need_bwd, groups=1, bwd_B=None, dual=None):
        E.require_cuda(x, 'input')
        xin = x.detach().contiguous().float()
        Bn, C_, H, W = xin.shape
        dev = xin.device
        order = E.StreamOrder.of(self) if not need_bwd else None      # inference calls: one at a time per module
        cur = order.enter() if order else None
        st = E.current_stream()
        wp = self._weights(dev)
        dp = None
        want_w = dual is not None or self._wants_param_grads()
        # _per_call_weights (Discriminator_VGG_128_SN): the weights a backward needs are those of ITS forward, and a later
        # forward rewrites the module's weight buffers (a new power iteration) before that backward runs — such plans own
        # their input-gradient operands and head weights, snapshotted at forward time
        own = need_bwd and getattr(self, '_per_call_weights', False)
        if need_bwd and not own:
            dp = self._dgrad_weights(dev)
            dp.ensure(st, force=(bool(self.training) or want_w) and not self.__dict__.get('_weights_clean', False)
                      and not self._dgrad_fresh())
        training = bool(self.training) and self._has_bn
        groups = groups if training else 1
        key = ('seq', Bn, H, W, self.precision, training, need_bwd, want_w, wp.generation, str(dev), groups, bwd_B, dual)
        pool = self._plans.setdefault(key, [])
        plan = next((p for p in pool if not p.busy), None)
        if plan is None:
            head = self._head()
            if own:
                dp = self._new_dgrad_pack(dev)
                head = {k: torch.empty_like(v) for k, v in head.items()} if head is not None else None
            plan = CN.build_seq_plan(self._spec(), wp, dp, self._pspec(), want_w, Bn, H, W, self.precision,
                                     dev, training, need_bwd, self._input_affine(), head,
                                     groups=groups, bwd_B=bwd_B if need_bwd else None, dual=dual)
            plan.own_dp, plan.own_head = (dp, head) if own else (None, None)
            if E.use_graphs() and dual is None:
                # hipGraph replay: the input lands in a fixed staging tensor (everything else these
                # plans touch — outputs, upstream gradients, BN sums — already lives in fixed buffers)
                plan.x_static = torch.empty_like(xin)
                plan.fwd.array()[plan.in_op].u.layout.nchw = plan.x_static.data_ptr()
                plan.graph = True
            pool.append(plan)
        plan.packs = (wp, dp)            # the lists hold raw pointers into the packs; a DataParallel replica dies before its backward
        lease = CN._Lease(plan) if need_bwd else None
        if own:
            plan.own_dp.ensure(st, force=True)
            if plan.own_head is not None:
                with torch.no_grad():
                    for k, v in self._head().items():
                        plan.own_head[k].copy_(v)
        if training:
            plan.sums_f.zero_()
        if getattr(plan, 'graph', False):
            plan.x_static.copy_(xin)
            plan.fwd.graph_launch(st)
        else:
            plan.fwd.array()[plan.in_op].u.layout.nchw = xin.data_ptr()
            plan.fwd.run(st)
        plan.keep_x = xin       # (num_batches_tracked is advanced by the BN finalize launches)
        y = plan.out_tensor.clone()
        if order:
            order.leave(cur)
        return y, lease

    # ---- the dual pair forward in two stages (train step, round 5): ``_pair_begin(b)`` runs the SECOND operand's half
    # (group 1: the train step's ``real``, known at the start of the step) and ``_pair_finish(handle, a)`` the first
    # operand's half once it exists; together they are ``_run_forward(cat([a, b]), need_bwd=True, groups=2, dual=n)``:
    # the same plan (same pool key), the same launches on half the batch each, the running-statistics updates in the
    # reference's order (the early half's update is deferred to plan.restat1, which the caller runs after the late half).
    def _pair_begin(self, b):
        E.require_cuda(b, 'input')
        xb = b.detach().contiguous().float()
        n, C_, H, W = xb.shape
        dev = xb.device
        st = E.current_stream()
        wp = self._weights(dev)
        dp = self._dgrad_weights(dev)
        dp.ensure(st, force=not self.__dict__.get('_weights_clean', False) and not self._dgrad_fresh())
        training = bool(self.training) and self._has_bn
        if not training or getattr(self, '_per_call_weights', False) or E.use_graphs():
            raise L.HipExtensionError('_pair_begin: BatchNorm network in train mode only')
        key = ('seq', 2 * n, H, W, self.precision, training, True, True, wp.generation, str(dev), 2, None, n)
        pool = self._plans.setdefault(key, [])
        plan = next((p for p in pool if not p.busy), None)
        if plan is None:
            plan = CN.build_seq_plan(self._spec(), wp, dp, self._pspec(), True, 2 * n, H, W, self.precision,
                                     dev, training, True, self._input_affine(), self._head(), groups=2, bwd_B=None, dual=n)
            plan.own_dp, plan.own_head = None, None
            pool.append(plan)
        if getattr(plan, 'fwd_half', None) is None:
            CN.split_forward_groups(plan, n)
        plan.packs = (wp, dp)
        lease = CN._Lease(plan)
        plan.sums_f.zero_()
        half = plan.fwd_half[1]
        half.array()[plan.in_op].u.layout.nchw = xb.data_ptr()
        half.run(st)
        plan.keep_x = (xb,)
        return plan, lease

    def _pair_finish(self, plan, a):
        xa = a.detach().contiguous().float()
        half = plan.fwd_half[0]
        half.array()[plan.in_op].u.layout.nchw = xa.data_ptr()
        half.run(E.current_stream())
        plan.keep_x = plan.keep_x + (xa,)
        return plan.out_tensor.clone()

    def forward(self, x):
        self._join_pending()
        need = torch.is_grad_enabled() and (x.requires_grad or self._wants_param_grads())
        if not need:
            return self._run_forward(x, need_bwd=False)[0]
        return CN.SeqNetFn.apply(x, self, *[t for _, t in self._pspec()])

    def forward_pair(self, a, b):
        """``(self(a), self(b))`` as ONE pass over the concatenated batch — the two calls the train step makes on
        every network (SRRaGAN_model.py:128-129 netF, 133-134 and 150-151 netD).  BatchNorm layers in train mode
        keep the two calls' semantics: batch statistics per half, running statistics updated once per half in call
        order (esr_bn.groups).  When no parameter requires a gradient and ``b`` does not either (the detached
        ``real`` operand of the G step), the backward runs on ``a``'s half only and ``self(b)`` comes back
        detached."""
        if a.shape != b.shape:
            raise ValueError('forward_pair: the two batches must have one shape')
        self._join_pending()
        n = a.shape[0]
        x = torch.cat([a, b])
        wantp = self._wants_param_grads()
        need = torch.is_grad_enabled() and (x.requires_grad or wantp)
        groups = 2 if self._has_bn else 1
        if not need:
            y = self._run_forward(x, need_bwd=False, groups=groups)[0]
            return y[:n], y[n:]
        half = not wantp and not b.requires_grad
        self._pair_opts = dict(groups=groups, bwd_B=n if half else None)
        try:
            y = CN.SeqNetFn.apply(x, self, *[t for _, t in self._pspec()])
        finally:
            self._pair_opts = {}
        return y[:n], (y[n:].detach() if half else y[n:])


    # One forward for the train step's TWO netD pairs (SRRaGAN_model.py:133-134 and 150-151): see forward_shared
    _shared_ok = False

    def forward_shared(self, a, b):
        """``self(a), self(b).detach()`` in one pass — as ``forward_pair`` — plus a handle whose ``second_pass()`` gives
        ``self(b), self(a.detach())`` again WITHOUT a second forward: between the G step's and the D step's pair of
        calls the weights do not change and the operands are the same values, so the activations are the same; what a
        real second pair adds — the BatchNorm running-statistics updates, in its own call order — is applied by the
        handle.  The first pair's backward gives dL/da (parameters frozen, as the G step has them); the second pair's
        gives the parameter gradients.  Training mode only."""
        if not (self._shared_ok and self.training and torch.is_grad_enabled() and a.requires_grad):
            raise RuntimeError('forward_shared: a training-mode pass whose first operand requires a gradient')
        if a.shape != b.shape:
            raise ValueError('forward_shared: the two batches must have one shape')
        self._join_pending()
        holder = []
        y = CN.SharedFirstFn.apply(a, b, self, holder)
        n = a.shape[0]
        return y[:n], y[n:].detach(), holder[0]


class _DiscriminatorVGG(_SeqNet):
    """The reference's VGG-style discriminators (architecture.py:87-129, 178-270): 3x3/s1 and 4x4/s2 conv pairs
    (64..512 ch) with BatchNorm2d (batch statistics in train mode) and LeakyReLU(0.2), flatten,
    Linear(F,100), LeakyReLU, Linear(100,1).  State-dict keys as the reference."""

    _has_bn = True
    _shared_ok = True
    _pairs = 5          # conv pairs (128 / 96: 5, 192: 6)
    _final = 4          # side of the final feature map at the nominal input size

    def __init__(self, in_nc, base_nf, norm_type='batch', act_type='leakyrelu', mode='CNA'):
        super().__init__()
        if (base_nf, norm_type, act_type.lower(), mode) != (64, 'batch', 'leakyrelu', 'CNA'):
            raise NotImplementedError('HIP %s supports base_nf=64, batch norm, leakyrelu, CNA '
                                      '(train_ESRGANplus.json:46-53)' % type(self).__name__)
        nf = base_nf
        widths = [nf, 2 * nf, 4 * nf, 8 * nf, 8 * nf, 8 * nf][:self._pairs]
        chans, cin = [], in_nc
        for i, w in enumerate(widths):
            chans.append((cin, w, 3, 1, None if i == 0 else norm_type))
            chans.append((w, w, 4, 2, norm_type))
            cin = w
        self.features = B.sequential(*[B.conv_block(ci, co, kernel_size=k, stride=s, norm_type=nt,
                                                    act_type=act_type, mode=mode)
                                       for ci, co, k, s, nt in chans])
        self.classifier = nn.Sequential(nn.Linear(512 * self._final * self._final, 100), nn.LeakyReLU(0.2, True),
                                        nn.Linear(100, 1))
        self.in_nc = in_nc
        self._init_planned()

    def _layers(self):
        out, mods, i = [], list(self.features.children()), 0
        while i < len(mods):
            conv = mods[i]
            bn = mods[i + 1] if isinstance(mods[i + 1], nn.BatchNorm2d) else None
            out.append((i, conv, (i + 1, bn) if bn is not None else None))
            i += 3 if bn is not None else 2
        return out

    def _conv_list(self):
        return [('features.%d' % i, c.weight, c.bias) for i, c, _ in self._layers()]

    def _spec(self):
        spec = []
        for i, c, bn in self._layers():
            d = {'conv': 'features.%d' % i, 'cin': c.in_channels, 'cout': c.out_channels,
                 'ks': c.kernel_size[0], 'stride': c.stride[0], 'act': L.ACT_LRELU, 'bn': None}
            if bn is not None:
                m = bn[1]
                d['bn'] = dict(weight=m.weight, bias=m.bias, rm=m.running_mean, rv=m.running_var,
                               nbt=m.num_batches_tracked)
            spec.append(d)
        return spec

    def _head(self):
        c = self.classifier
        return dict(w1=c[0].weight, b1=c[0].bias, w2=c[2].weight, b2=c[2].bias)

    def _pspec(self):
        ps, k = [], 0
        for i, c, bn in self._layers():
            ps += [('features.%d.weight' % i, c.weight), ('features.%d.bias' % i, c.bias)]
            if bn is not None:
                ps += [('bn%d.weight' % k, bn[1].weight), ('bn%d.bias' % k, bn[1].bias)]
                k += 1
        c = self.classifier
        ps += [('head1.weight', c[0].weight), ('head1.bias', c[0].bias),
               ('head2.weight', c[2].weight), ('head2.bias', c[2].bias)]
        return ps


class Discriminator_VGG_128(_DiscriminatorVGG):
    """codes/models/modules/architecture.py:87-129 — 128x128 inputs: 10 convs, Linear(512*4*4, 100)."""
    _pairs, _final = 5, 4


class Discriminator_VGG_96(_DiscriminatorVGG):
    """architecture.py:178-221 — 96x96 inputs: the same 10 convs, Linear(512*3*3, 100)."""
    _pairs, _final = 5, 3


class Discriminator_VGG_192(_DiscriminatorVGG):
    """architecture.py:224-270 — 192x192 inputs: 12 convs (one more 512-channel pair), Linear(512*3*3, 100)."""
    _pairs, _final = 6, 3


class _SNLayer(nn.Module):
    """Parameter holder of one spectrally normalised layer with the reference's state-dict entries
    (spectral_norm.py:55-75): ``weight_orig`` + ``bias`` (parameters), ``weight`` (buffer: the normalised weight the
    last training forward used; what eval-mode forwards use) and ``weight_u`` (buffer: the left singular vector
    estimate)."""

    def __init__(self, shape, eps=1e-12):
        super().__init__()
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        bound = 1.0 / math.sqrt(fan_in)
        self.weight_orig = nn.Parameter(torch.empty(shape).uniform_(-bound, bound))
        self.bias = nn.Parameter(torch.empty(shape[0]).uniform_(-bound, bound))
        self.register_buffer('weight', self.weight_orig.detach().clone())
        u = torch.randn(shape[0])
        self.register_buffer('weight_u', u / u.norm().clamp_min(eps))
        self.eps = eps

    def normalised(self):
        """One power iteration (spectral_norm.py:21-41): v = W^T u / |.|, u = W v / |.| without gradients (``weight_u``
        is updated in place), sigma = u . (W v) and W / sigma WITH the gradient through sigma.  Returns the
        autograd tensor and refreshes the ``weight`` buffer with its value."""
        w = self.weight_orig
        wm = w.reshape(w.shape[0], -1)
        with torch.no_grad():
            v = wm.t().mv(self.weight_u)
            v = v / v.norm().clamp_min(self.eps)
            u = wm.mv(v)
            u = u / u.norm().clamp_min(self.eps)
            self.weight_u.copy_(u)
        sigma = torch.dot(u, wm.mv(v))
        w_eff = w / sigma
        with torch.no_grad():
            self.weight.copy_(w_eff)
        return w_eff


class Discriminator_VGG_128_SN(_SeqNet):
    """codes/models/modules/architecture.py:131-175 (``which_model_D: discriminator_vgg_128_SN``, networks.py:130-131):
    the 128x128 VGG-style discriminator without BatchNorm, every conv / linear weight divided by its spectral norm
    (spectral_norm.py: one power iteration per training forward).  Same state-dict keys (``conv0.weight_orig``,
    ``conv0.bias``, ``conv0.weight``, ``conv0.weight_u``, ... ``linear1.*``).  The ten convs and two linears run on the
    discriminator plans; the normalisation itself — a handful of matrix-vector products on the weights — is torch
    code in front of them, and autograd carries the plans' weight gradients through ``W / sigma`` to ``weight_orig``."""

    _has_bn = False
    _per_call_weights = True

    def __init__(self):
        super().__init__()
        chans, cin = [], 3
        for w in (64, 128, 256, 512, 512):
            chans += [(cin, w, 3, 1), (w, w, 4, 2)]
            cin = w
        for i, (ci, co, k, s) in enumerate(chans):
            setattr(self, 'conv%d' % i, _SNLayer((co, ci, k, k)))
        self._geom = chans
        self.linear0 = _SNLayer((100, 512 * 4 * 4))
        self.linear1 = _SNLayer((1, 100))
        self.in_nc = 3
        self._eff = None
        self._init_planned()

    def _sn_layers(self):
        return [getattr(self, 'conv%d' % i) for i in range(10)] + [self.linear0, self.linear1]

    def _conv_list(self):
        return [('conv%d' % i, getattr(self, 'conv%d' % i).weight, getattr(self, 'conv%d' % i).bias) for i in range(10)]

    def _spec(self):
        return [{'conv': 'conv%d' % i, 'cin': ci, 'cout': co, 'ks': k, 'stride': s, 'act': L.ACT_LRELU, 'bn': None}
                for i, (ci, co, k, s) in enumerate(self._geom)]

    def _head(self):
        return dict(w1=self.linear0.weight, b1=self.linear0.bias, w2=self.linear1.weight, b2=self.linear1.bias)

    def _pspec(self):
        # the tensors the plan's gradients belong to: the normalised weights of this forward (autograd carries them on
        # to weight_orig) or, outside a training forward, the buffers
        ws = self._eff if self._eff is not None else [m.weight for m in self._sn_layers()]
        ps = []
        for i in range(10):
            ps += [('conv%d.weight' % i, ws[i]), ('conv%d.bias' % i, getattr(self, 'conv%d' % i).bias)]
        ps += [('head1.weight', ws[10]), ('head1.bias', self.linear0.bias),
               ('head2.weight', ws[11]), ('head2.bias', self.linear1.bias)]
        return ps

    def forward(self, x):
        if self.training:
            self._eff = [m.normalised() for m in self._sn_layers()]
            self.invalidate()                   # the buffers the plans pack from were rewritten in place
        try:
            return super().forward(x)
        finally:
            self._eff = None

    def forward_pair(self, a, b):
        # two calls, as the reference makes them: each runs its own power iteration
        return self(a), self(b)


VGG19_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M',
             512, 512, 512, 512, 'M']


class VGGFeatureExtractor(_SeqNet):
    """architecture.py:279-307: ``(x - mean)/std`` then torchvision VGG19 ``features[:feature_layer+1]``
    (cfg 'E'; feature_layer=34 -> conv5_4 before its ReLU), frozen.  The reference pulls ImageNet
    weights through ``torchvision.models.vgg19(pretrained=True)``; there is no network here, so the
    conv weights are whatever the caller loads (``load_state_dict`` with keys ``features.N.weight``)."""

    _has_bn = False

    def __init__(self, feature_layer=34, use_bn=False, use_input_norm=True, device=torch.device('cpu')):
        super().__init__()
        if use_bn:
            raise NotImplementedError('vgg19_bn is not used by the ESRGAN+ recipe (networks.py:145-150)')
        layers, cin = [], 3
        for v in VGG19_CFG:
            if v == 'M':
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        self.features = nn.Sequential(*layers[:feature_layer + 1])
        self.use_input_norm = use_input_norm
        if use_input_norm:
            self.register_buffer('mean', torch.Tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1).to(device))
            self.register_buffer('std', torch.Tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1).to(device))
        for p in self.features.parameters():       # architecture.py:300-301
            p.requires_grad = False
        self._init_planned()

    def _conv_list(self):
        return [('features.%d' % i, m.weight, m.bias) for i, m in enumerate(self.features)
                if isinstance(m, nn.Conv2d)]

    def _spec(self):
        spec, mods = [], list(self.features.children())
        for i, m in enumerate(mods):
            if isinstance(m, nn.Conv2d):
                act = L.ACT_RELU if (i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)) else L.ACT_NONE
                spec.append({'conv': 'features.%d' % i, 'cin': m.in_channels, 'cout': m.out_channels,
                             'ks': 3, 'stride': 1, 'act': act, 'bn': None})
            elif isinstance(m, nn.MaxPool2d):
                spec.append({'pool': True})
        return spec

    def _input_affine(self):
        if not self.use_input_norm:
            return None
        mean = [float(v) for v in self.mean.flatten().tolist()]
        inv = [1.0 / float(v) for v in self.std.flatten().tolist()]
        return (mean, inv)

    def _pspec(self):
        ps = []
        for k, w, b in self._conv_list():
            ps += [(k + '.weight', w), (k + '.bias', b)]
        return ps


class Conv2dHIP(_SeqNet, nn.Conv2d):
    """An nn.Conv2d (same parameters, same state-dict keys) whose forward AND backward run on the HIP kernels as a
    one-layer plan (convnet.build_seq_plan: layout in, fused conv, layout out; dgrad + wgrad + layout back) — the
    building block of the module families that are not planned as a whole.  3x3 / 1x1 stride 1 and 4x4 stride 2,
    zero padding (k-1)//2, like every conv of the reference's generators and discriminators."""

    _has_bn = False

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True):
        nn.Conv2d.__init__(self, in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                           padding=padding, dilation=1, groups=1, bias=bias)
        ks, st = self.kernel_size[0], self.stride[0]
        if (self.kernel_size[0] != self.kernel_size[1] or (ks, st) not in ((3, 1), (1, 1), (4, 2))
                or self.padding != ((ks - 1) // 2,) * 2):
            raise NotImplementedError('Conv2dHIP: 3x3/s1, 1x1/s1 or 4x4/s2 with padding (k-1)//2')
        self._init_planned()

    def _conv_list(self):
        return [('conv', self.weight, self.bias)]

    def _spec(self):
        return [{'conv': 'conv', 'cin': self.in_channels, 'cout': self.out_channels, 'ks': self.kernel_size[0],
                 'stride': self.stride[0], 'act': L.ACT_NONE, 'bn': None}]

    def _pspec(self):
        ps = [('conv.weight', self.weight)]
        if self.bias is not None:
            ps.append(('conv.bias', self.bias))
        return ps


class SRResNet(_SeqNet):
    """architecture.py:13-44 of the reference (networks.py:88-91 builds it with act 'relu', upsample_mode
    'pixelshuffle'; train_SRResNet.json:39-43: norm_type null, mode CNA, nb 16).  Same module tree and state-dict keys
    as the reference.  Round 5: planned as a WHOLE — one forward and one backward launch list per input shape
    (convnet.build_seq_plan), one autograd node — instead of a chain of per-conv modules with torch glue: ReLU and the
    ResNetBlock / ShortcutBlock residual adds are conv epilogues (block.py:199-232, 84-86), nearest-x2 up-sampling is
    folded into the conv's load (block.py:315-322), nn.PixelShuffle(2) is one index-remapping launch whose ReLU rides
    in the producing conv's epilogue (block.py:299-312; ReLU commutes with the shuffle), the backward mirrors it
    (skip gradients enter the dgrad convs as epilogue residuals).  x3 (PixelShuffle(3)) keeps the per-conv modules."""

    _has_bn = False

    def __init__(self, in_nc, out_nc, nf, nb, upscale=4, norm_type=None, act_type='relu', mode='CNA', res_scale=1,
                 upsample_mode='upconv'):
        super().__init__()
        if norm_type or mode != 'CNA':
            raise NotImplementedError('HIP SRResNet: norm_type null, mode CNA (train_SRResNet.json:40-41)')
        if act_type not in ('relu', None):
            raise NotImplementedError('HIP SRResNet: act_type relu (networks.py:88-91)')
        n_upscale = 1 if upscale == 3 else int(math.log(upscale, 2))
        self._planned = upscale != 3              # x3: per-conv modules + torch glue (nn.PixelShuffle(3))
        hip = not self._planned
        fea_conv = B.conv_block(in_nc, nf, kernel_size=3, norm_type=None, act_type=None, hip=hip)
        blocks = [B.ResNetBlock(nf, nf, nf, norm_type=None, act_type=act_type, mode=mode, res_scale=res_scale, hip=hip)
                  for _ in range(nb)]
        LR_conv = B.conv_block(nf, nf, kernel_size=3, norm_type=None, act_type=None, mode=mode, hip=hip)
        if upsample_mode == 'upconv':
            up = lambda *a, **k: B.upconv_blcok(*a, hip=hip, **k)
        elif upsample_mode == 'pixelshuffle':
            up = lambda *a, **k: B.pixelshuffle_block(*a, hip=hip, **k)
        else:
            raise NotImplementedError('upsample mode [{:s}] is not found'.format(upsample_mode))
        if upscale == 3:
            if upsample_mode == 'upconv':
                raise NotImplementedError('x3 upconv is outside the kernels (nearest x2 only)')
            upsampler = [up(nf, nf, 3, act_type=act_type)]
        else:
            upsampler = [up(nf, nf, act_type=act_type) for _ in range(n_upscale)]
        HR_conv0 = B.conv_block(nf, nf, kernel_size=3, norm_type=None, act_type=act_type, hip=hip)
        HR_conv1 = B.conv_block(nf, out_nc, kernel_size=3, norm_type=None, act_type=None, hip=hip)
        self.model = B.sequential(fea_conv, B.ShortcutBlock(B.sequential(*blocks, LR_conv)), *upsampler,
                                  HR_conv0, HR_conv1)
        self.in_nc, self.out_nc, self.nf, self.nb = in_nc, out_nc, nf, nb
        self.n_upscale, self.upsample_mode, self.res_scale = n_upscale, upsample_mode, float(res_scale)
        self.act = L.ACT_RELU if act_type == 'relu' else L.ACT_NONE
        self._init_planned()

    def set_precision(self, precision):
        super().set_precision(precision)
        for m in self.modules():
            if isinstance(m, Conv2dHIP):
                m.set_precision(precision)
        return self

    def _named_convs(self):
        return [('model.' + n, m) for n, m in self.model.named_modules() if isinstance(m, nn.Conv2d)]

    def _conv_list(self):
        return [(k, m.weight, m.bias) for k, m in self._named_convs()]

    def _pspec(self):
        ps = []
        for k, m in self._named_convs():
            ps += [(k + '.weight', m.weight), (k + '.bias', m.bias)]
        return ps

    def _dgrad_special(self):
        if self.upsample_mode != 'upconv':
            return {}
        convs = self._named_convs()
        return {k: {'ups': True} for k, _ in convs[2 + 2 * self.nb:2 + 2 * self.nb + self.n_upscale]}

    def _spec(self):
        convs = self._named_convs()
        assert len(convs) == 4 + 2 * self.nb + self.n_upscale

        def cv(i, act, **kw):
            k, m = convs[i]
            return dict({'conv': k, 'cin': m.in_channels, 'cout': m.out_channels, 'ks': 3, 'stride': 1, 'act': act,
                         'bn': None}, **kw)
        spec = [cv(0, L.ACT_NONE), {'save': 'fea'}]
        for i in range(self.nb):
            # ResNetBlock (block.py:229-232): x + res_scale * conv1(act(conv0(x)))
            spec += [{'save': 'b%d' % i}, cv(1 + 2 * i, self.act), cv(2 + 2 * i, L.ACT_NONE, res='b%d' % i, alpha=self.res_scale)]
        spec.append(cv(1 + 2 * self.nb, L.ACT_NONE, res='fea', alpha=1.0))        # ShortcutBlock: x + sub(x)
        for u in range(self.n_upscale):
            i = 2 + 2 * self.nb + u
            if self.upsample_mode == 'upconv':
                spec.append(cv(i, self.act, ups=True))
            else:
                spec += [cv(i, self.act), {'shuffle': 2}]
        i = 2 + 2 * self.nb + self.n_upscale
        spec += [cv(i, self.act), cv(i + 1, L.ACT_NONE)]
        return spec

    def forward(self, x):
        if not self._planned:
            return self.model(x)
        return super().forward(x)


