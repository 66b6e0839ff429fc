"""Launch plans (forward + backward) for the two feed-forward conv nets of the ESRGAN+ train step:
``Discriminator_VGG_128`` (codes/models/modules/architecture.py:87-129) and the VGG19 feature
extractor (architecture.py:279-307; body = torchvision cfg 'E' ``features[:35]``).

Same mechanics as engine.py: G32 buffers, fused-conv launches, one ``esr_run_ops`` per pass.
Convs keep their LeakyReLU / ReLU in the epilogue; BatchNorm2d runs as stats -> finalize -> apply
passes over the conv output; the backward applies activation masks inside the producing kernels
(conv epilogue ``out2``, BN backward, max-pool backward).
"""
import os

import torch

from . import _lib as L
from . import engine as E

BN_MOMENTUM, BN_EPS = 0.1, 1e-5      # nn.BatchNorm2d defaults (block.py:31)
# ESR_FUSE_BN=0: the five-launch BatchNorm of round 3 (stats, finalize, apply / reduce, final, apply) for A/B runs
def fuse_bn():
    return os.environ.get('ESR_FUSE_BN', '1') != '0'


class _Lease(object):
    def __init__(self, plan):
        self.plan = plan
        plan.busy = True

    def release(self):
        if self.plan is not None:
            self.plan.busy = False
            self.plan = None

    def __del__(self):
        self.release()


def s2_ksplit(B, wo, ho, cin, cout, dt_e):
    """K split of a 4x4/s2 conv that leaves a square map of 4 / 8 / 16 columns (the discriminators' deep layers:
    include/esrgan_hip.h, esr_conv.ksplit): enough workgroups to fill the chip, at least two K steps each; 0 = the
    plain one-image-per-tile launch (other shapes, fp32, ESR_S2_SPLIT=0)."""
    if dt_e != L.ESR_F16 or wo != ho or wo not in (4, 8, 16) or B < 2 or os.environ.get('ESR_S2_SPLIT', '1') == '0':
        return 0
    nchunks = (cin + 15) // 16
    per_tile = (32 // wo) * (2 if wo <= 4 else 1)
    tyn = 1 if wo <= 4 else (wo + 7) // 8
    base = ((B + per_tile - 1) // per_tile) * tyn * ((cout + 127) // 128)
    k = 1
    while k * 2 <= nchunks // 2 and k * 2 * base <= 256:
        k *= 2
    return k if k >= 2 else 0


def _lin(mode, B_, I, O, act, **kw):
    o = L.esr_linear()
    o.mode, o.B, o.I, o.O, o.act = mode, B_, I, O, act
    for k, v in kw.items():
        setattr(o, k, v)
    return o


class SeqPlan:
    """Forward/backward launch lists of a conv -> [bn] -> act -> [pool] chain (+ optional linear
    head) for one input shape."""

    def __init__(self):
        self.fwd, self.bwd = L.OpList(), L.OpList()
        self.bufs, self.keep = [], []
        self.busy = False
        self.in_op = self.out_tensor = None
        self.gy_tensor = None           # fp32 tensor the upstream gradient is copied into
        self.gx_tensor = None           # fp32 NCHW gradient w.r.t. the input image
        self.sums_f = self.sums_b = None
        self.grad_flat, self.grad_views = None, None
        self.bn_layers = []
        self.second = None              # BwdPass of a dual plan (build_seq_plan(dual=...)): the G step's input-gradient pass
        self.restat = None              # OpList replaying the BatchNorm running-statistics updates (dual plans)


class BwdPass:
    """One backward launch list over a plan's saved activations with its own scratch buffers."""

    def __init__(self):
        self.bwd = L.OpList()
        self.bufs, self.keep = [], []
        self.gy_tensor = self.gx_tensor = self.sums_b = None
        self.grad_flat = self.grad_views = self.tapmajor = self.wgrad_arena = None


def _g32(plan, B, C_, H, W, dtype, dev):
    b = E.G32(B, C_, H, W, dtype, dev)
    plan.bufs.append(b)
    return b


def _layout(ops, dt_e, to_g32, B, C_, buf, nchw_ptr=None, affine=None):
    lo = L.esr_layout()
    lo.dtype, lo.to_g32 = dt_e, to_g32
    lo.B, lo.C, lo.H, lo.W = B, C_, buf.H, buf.W
    lo.g32 = buf.view(0, C_)
    if nchw_ptr is not None:
        lo.nchw = nchw_ptr
    if affine is not None:
        lo.use_affine = 1
        for i in range(len(affine[0])):
            lo.mean_c[i] = affine[0][i]
            lo.inv_std_c[i] = affine[1][i]
    return ops.add(L.OP_LAYOUT, 'layout', lo)


def build_seq_plan(spec, wp, dp, pspec, want_wgrad, B, H, W, dtype, dev, training, need_bwd,
                   input_affine=None, head=None, groups=1, bwd_B=None, dual=None):
    """dual (None | images): ONE forward whose saved activations serve TWO backward passes with their own scratch
    buffers — P (full batch, parameter gradients: the D step) and P.second (the first `dual` images only, input
    gradient only: the G step's pass through the frozen discriminator) — plus P.restat, the launch list that applies
    the BatchNorm running-statistics updates of a second forward call over the same batch in reverse group order
    (SRRaGAN_model.py:133-134 then 150-151: netD sees fake, real, then real, fake with unchanged weights, i.e. the
    same activations twice).
    groups: BatchNorm statistics groups of the batch (``forward_pair``: two reference forward calls as one
    pass; esr_bn.groups).  bwd_B: the backward covers only the first ``bwd_B`` images (the second half of a pair
    that is detached: its saved activations are the batch suffix of every buffer, so the backward launches simply
    run on the prefix) — with groups == 2 that prefix is statistics group 0.

       spec: list of dicts, in execution order:
         {'conv': key, 'cin', 'cout', 'ks', 'stride', 'act': ACT_*, 'bn': None | dict(weight,bias,rm,rv)}
             optional (round 5, SRResNet: architecture.py:13-44, block.py:199-232,299-322 as ONE launch plan):
             'ups': True        nearest x2 up-sampling folded into the conv's load (upconv_blcok)
             'res': tag, 'alpha': a    out = conv(x) * a + saved[tag]   (ResNetBlock / ShortcutBlock adds, in the epilogue)
         {'save': tag}          the current tensor is a residual source (no launch)
         {'shuffle': 2}         nn.PixelShuffle(2) (its ReLU rides in the producing conv's epilogue: it commutes)
         {'pool': True}
       head: None | dict(w1,b1,w2,b2) — flatten + Linear(.,100) + LeakyReLU + Linear(100,1)
       pspec: [(name, tensor)] in autograd-argument order; names '<convkey>.weight|bias',
       'bn<k>.weight|bias', 'head1|head2.weight|bias'.  want_wgrad: emit parameter-gradient launches.
    """
    dt_e, tdtype, cpg = E._dt(dtype)
    P = SeqPlan()
    f, bk = P.fwd, P.bwd
    e = wp.entries
    P.tapmajor = None
    P.grad_views = [(t.numel(), tuple(t.shape)) for _, t in pspec]

    # ---------------------------------------------------------------- forward
    cin0 = spec[0]['cin']
    xin = _g32(P, B, cin0, H, W, dtype, dev)
    P.in_op = _layout(f, dt_e, 1, B, cin0, xin, affine=input_affine)
    nbn = sum(1 for s in spec if s.get('bn'))
    maxc = max([s['cout'] for s in spec if 'conv' in s] + [1])
    Bb0 = B if bwd_B is None else bwd_B
    assert B % groups == 0 and (Bb0 == B or (groups > 1 and Bb0 == B // groups) or groups == 1)
    P.sums_f = torch.zeros(max(nbn, 1) * 2 * maxc * groups, dtype=torch.float64, device=dev)
    stats = torch.zeros(max(nbn, 1) * 2 * maxc * groups, dtype=torch.float32, device=dev)   # mean | invstd, [groups][C] each
    P.keep += [stats]
    cur, ch, h, w = xin, cin0, H, W
    recs = []          # per layer record for the backward
    ibn = 0
    saved = {}
    for s in spec:
        if 'save' in s:
            saved[s['save']] = (cur, ch)
            recs.append(dict(kind='save', tag=s['save']))
            continue
        if 'shuffle' in s:
            if s['shuffle'] != 2 or ch % (4 * cpg):
                raise L.HipExtensionError('pixel shuffle: factor 2 on a multiple of %d channels' % (4 * cpg))
            y = _g32(P, B, ch // 4, 2 * h, 2 * w, dtype, dev)
            pl = L.esr_pool()
            pl.dtype, pl.mode, pl.B, pl.C, pl.H, pl.W = dt_e, L.POOL_SHUFFLE, B, ch // 4, h, w
            pl.x, pl.y = cur.view(0, ch), y.view(0, ch // 4)
            f.add(L.OP_POOL, 'pool', pl)
            recs.append(dict(kind='shuffle', x=cur, y=y, ch=ch // 4, h=h, w=w))
            cur, ch, h, w = y, ch // 4, 2 * h, 2 * w
            continue
        if 'pool' in s:
            y = _g32(P, B, ch, h // 2, w // 2, dtype, dev)
            pl = L.esr_pool()
            pl.dtype, pl.mode, pl.B, pl.C, pl.H, pl.W = dt_e, 0, B, ch, h // 2, w // 2
            pl.x, pl.y = cur.view(0, ch), y.view(0, ch)
            f.add(L.OP_POOL, 'pool', pl)
            recs.append(dict(kind='pool', x=cur, y=y, ch=ch, h=h // 2, w=w // 2))
            cur, h, w = y, h // 2, w // 2
            continue
        key, cout, ks, st = s['conv'], s['cout'], s['ks'], s['stride']
        pad = (ks - 1) // 2
        ho, wo = (h + 2 * pad - ks) // st + 1, (w + 2 * pad - ks) // st + 1
        bn = s.get('bn')
        ups, res = bool(s.get('ups')), s.get('res')
        if (ups or res is not None) and (bn is not None or ks != 3 or st != 1 or (res is not None and s['act'] != L.ACT_NONE)):
            raise L.HipExtensionError('conv %s: up-sampling / residual epilogues are for plain 3x3 stride-1 convs' % key)
        if ups:
            ho, wo = 2 * h, 2 * w
        cpad = ((cout + 31) // 32) * 32
        yb = _g32(P, B, cpad, ho, wo, dtype, dev)
        if bn is None:
            c = E._conv(dt_e, B, ho, wo, cur.view(0), ch, yb.view(0), e[key], s['act'], stride=st, upsample=1 if ups else 0)
            if res is not None:
                src, src_ch = saved[res]
                assert src_ch == cout and (src.H, src.W) == (ho, wo), 'residual source of %s has another shape' % key
                c.res1, c.alpha = src.view(0, cout), float(s.get('alpha', 1.0))
            ksp = s2_ksplit(B, wo, ho, ch, cout, dt_e) if (st == 2 and ks == 4) else 0
            if ksp:
                ws = torch.empty(ksp * B * ho * wo * cpad, dtype=torch.float32, device=dev)
                P.keep.append(ws)
                c.ksplit, c.split_ws = ksp, ws.data_ptr()
            f.add_conv(c)
            recs.append(dict(kind='conv', key=key, x=cur, cin=ch, y=yb, c=None, cout=cout, ks=ks, st=st,
                             act=s['act'], h=ho, w=wo, hin=h, win=w, bn=None, ups=ups, res=res,
                             alpha=float(s.get('alpha', 1.0))))
        else:
            cb = _g32(P, B, cpad, ho, wo, dtype, dev)
            cv = E._conv(dt_e, B, ho, wo, cur.view(0), ch, cb.view(0), e[key], L.ACT_NONE, stride=st)
            base = ibn * 2 * maxc * groups
            ksp = s2_ksplit(B, wo, ho, ch, cout, dt_e) if (st == 2 and ks == 4) else 0
            stats_in_conv = False
            if ksp:
                # deep stride-2 layer: packed tiles + split K; its finishing pass also takes the BatchNorm statistics
                ws = torch.empty(ksp * B * ho * wo * cpad, dtype=torch.float32, device=dev)
                P.keep.append(ws)
                cv.ksplit, cv.split_ws = ksp, ws.data_ptr()
                if training and fuse_bn():
                    cv.stat_sums, cv.stat_groups, cv.stat_C = P.sums_f.data_ptr() + 8 * base, groups, cout
                    stats_in_conv = True
            f.add_conv(cv)
            mk = dict(sums_f=P.sums_f.data_ptr() + 8 * base, base=base,
                      mean=stats.data_ptr() + 4 * base, invstd=stats.data_ptr() + 4 * (base + maxc * groups))

            def bnop(mode, x=cb, y=yb, g=None, gx=None, sums=mk['sums_f'], act=s['act'], bn=bn, mk=mk,
                     cout=cout, ho=ho, wo=wo, Bb=None, gb=None):
                o = L.esr_bn()
                fwd_op = mode in (L.BN_STATS, L.BN_FINALIZE, L.BN_APPLY, L.BN_RESTAT, L.BN_FIN_APPLY)
                o.dtype, o.mode, o.B, o.C, o.H, o.W = dt_e, mode, (B if fwd_op else Bb), cout, ho, wo
                o.groups = groups if fwd_op else gb
                if mode in (L.BN_FINALIZE, L.BN_RESTAT, L.BN_FIN_APPLY) and training and bn.get('nbt') is not None:
                    o.num_batches_tracked = bn['nbt'].data_ptr()
                o.training, o.act, o.momentum, o.eps = int(training), act, BN_MOMENTUM, BN_EPS
                o.x, o.y = x.view(0, cout), y.view(0, cout)
                if g is not None:
                    o.g = g.view(0, cout)
                if gx is not None:
                    o.gx = gx.view(0, cout)
                o.sums, o.mean, o.invstd = sums, mk['mean'], mk['invstd']
                o.gamma, o.beta = bn['weight'].data_ptr(), bn['bias'].data_ptr()
                o.running_mean, o.running_var = bn['rm'].data_ptr(), bn['rv'].data_ptr()
                return o
            if training and fuse_bn():
                # statistics pass (unless the conv's finishing pass took them), then ONE pass that finalizes and
                # applies (ESR_BN_FIN_APPLY)
                if not stats_in_conv:
                    f.add(L.OP_BN, 'bn', bnop(L.BN_STATS))
                f.add(L.OP_BN, 'bn', bnop(L.BN_FIN_APPLY))
            else:
                if training:
                    f.add(L.OP_BN, 'bn', bnop(L.BN_STATS))
                f.add(L.OP_BN, 'bn', bnop(L.BN_FINALIZE))
                f.add(L.OP_BN, 'bn', bnop(L.BN_APPLY))
            recs.append(dict(kind='conv', key=key, x=cur, cin=ch, y=yb, c=cb, cout=cout, ks=ks, st=st,
                             act=s['act'], h=ho, w=wo, hin=h, win=w, bn=bn, bnop=bnop, mk=mk, ibn=ibn))
            P.bn_layers.append(bn)
            ibn += 1
        cur, ch, h, w = yb, cout, ho, wo
    if head is None:
        P.out_tensor = torch.empty(B, ch, h, w, dtype=torch.float32, device=dev)
        _layout(f, dt_e, 0, B, ch, cur, nchw_ptr=P.out_tensor.data_ptr())
    else:
        F_ = torch.empty(B, ch * h * w, dtype=torch.float32, device=dev)
        H1 = torch.empty(B, head['w1'].shape[0], dtype=torch.float32, device=dev)
        P.out_tensor = torch.empty(B, head['w2'].shape[0], dtype=torch.float32, device=dev)
        P.keep += [F_, H1]
        _layout(f, dt_e, 0, B, ch, cur, nchw_ptr=F_.data_ptr())
        lin = _lin
        I1, O1, O2 = ch * h * w, head['w1'].shape[0], head['w2'].shape[0]
        f.add(L.OP_LINEAR, 'linear', lin(0, B, I1, O1, L.ACT_LRELU, x=F_.data_ptr(), w=head['w1'].data_ptr(),
                                         b=head['b1'].data_ptr(), y=H1.data_ptr()))
        f.add(L.OP_LINEAR, 'linear', lin(0, B, O1, O2, L.ACT_NONE, x=H1.data_ptr(), w=head['w2'].data_ptr(),
                                         b=head['b2'].data_ptr(), y=P.out_tensor.data_ptr()))
    if not need_bwd:
        return P

    def build_backward(Q, Bb, want_wgrad):
        """Backward launch list over the first Bb images into the BwdPass / SeqPlan Q (own scratch buffers)."""
        gb = groups if Bb == B else 1           # statistics groups the backward sees
        bk = Q.bwd
        params_grad, poff = {}, {}
        Q.tapmajor, Q.grad_flat = None, None
        Q.grad_views = [(t.numel(), tuple(t.shape)) for _, t in pspec]
        Q.sums_b = torch.zeros(max(nbn, 1) * 2 * maxc * groups, dtype=torch.float64, device=dev)
        if want_wgrad:
            Q.grad_flat = torch.zeros(sum(t.numel() for _, t in pspec), dtype=torch.float32, device=dev)
            if dt_e == L.ESR_F16:
                Q.tapmajor = E.TapMajorGrads(Q.grad_flat)
            ptr, off = {}, 0
            for name, t in pspec:
                ptr[name] = Q.grad_flat.data_ptr() + 4 * off
                poff[name] = off
                off += t.numel()
            for name in ptr:
                base = name.rsplit('.', 1)[0]
                if base not in params_grad:
                    params_grad[base] = (ptr.get(base + '.weight'), ptr.get(base + '.bias'))

        def g32q(C_, h_, w_):
            b_ = E.G32(Bb, C_, h_, w_, dtype, dev)
            Q.bufs.append(b_)
            return b_

        # ---------------------------------------------------------------- backward
        de = dp.entries
        gcur = None      # G32 gradient w.r.t. the current layer's OUTPUT (post-activation)
        if head is None:
            Q.gy_tensor = torch.empty(Bb, ch, h, w, dtype=torch.float32, device=dev)
            gcur = g32q(ch, h, w)
            _layout(bk, dt_e, 1, Bb, ch, gcur, nchw_ptr=Q.gy_tensor.data_ptr())
        else:
            Q.gy_tensor = torch.empty(Bb, O2, dtype=torch.float32, device=dev)
            gH1 = torch.empty(Bb, O1, dtype=torch.float32, device=dev)
            gF = torch.empty(Bb, I1, dtype=torch.float32, device=dev)
            Q.keep += [gH1, gF]
            lin = _lin
            g2 = params_grad.get('head2')
            g1 = params_grad.get('head1')
            if g2 is not None:
                bk.add(L.OP_LINEAR, 'linear', lin(2, Bb, O1, O2, L.ACT_NONE, x=H1.data_ptr(), g=Q.gy_tensor.data_ptr(),
                                                  dw=g2[0], db=g2[1], w=head['w2'].data_ptr()))
            bk.add(L.OP_LINEAR, 'linear', lin(1, Bb, O1, O2, L.ACT_NONE, g=Q.gy_tensor.data_ptr(),
                                              w=head['w2'].data_ptr(), gx=gH1.data_ptr()))
            if g1 is not None:
                bk.add(L.OP_LINEAR, 'linear', lin(2, Bb, I1, O1, L.ACT_LRELU, x=F_.data_ptr(), g=gH1.data_ptr(),
                                                  ysaved=H1.data_ptr(), dw=g1[0], db=g1[1], w=head['w1'].data_ptr()))
            # a last conv that feeds the head through its activation without a norm layer (Discriminator_VGG_128_SN):
            # the head's input gradient is masked by that activation here (F_ holds the activation's output)
            last = recs[-1]
            head_masks = last['kind'] == 'conv' and last['bn'] is None and last['act'] != L.ACT_NONE
            o = lin(1, Bb, I1, O1, L.ACT_LRELU, g=gH1.data_ptr(), ysaved=H1.data_ptr(), w=head['w1'].data_ptr(), gx=gF.data_ptr())
            if head_masks:
                o.x, o.in_act = F_.data_ptr(), last['act']
            bk.add(L.OP_LINEAR, 'linear', o)
            gcur = g32q(ch, h, w)
            _layout(bk, dt_e, 1, Bb, ch, gcur, nchw_ptr=gF.data_ptr())

        # gcur_masked: True when gcur already is the gradient w.r.t. the producing conv's pre-activation
        masked = head is not None and head_masks
        skips = {}       # residual tag -> gradient buffers that flow back to the saved tensor over the skip

        def before(li):
            """(the layer in front of recs[li] in execution order, the residual tags saved in between)"""
            k, tags = li - 1, []
            while k >= 0 and recs[k]['kind'] == 'save':
                tags.append(recs[k]['tag'])
                k -= 1
            return (recs[k] if k >= 0 else None), tags

        for li in range(len(recs) - 1, -1, -1):
            r = recs[li]
            if r['kind'] == 'save':
                continue
            if r['kind'] == 'shuffle':
                prev, tags = before(li)
                if tags:
                    raise L.HipExtensionError('a residual source right in front of a pixel shuffle is not supported')
                gx = g32q(4 * r['ch'], r['h'], r['w'])
                pl = L.esr_pool()
                pl.dtype, pl.mode, pl.B, pl.C, pl.H, pl.W = dt_e, L.POOL_UNSHUFFLE, Bb, r['ch'], r['h'], r['w']
                pl.x, pl.g, pl.gx = r['x'].view(0, 4 * r['ch']), gcur.view(0, r['ch']), gx.view(0, 4 * r['ch'])
                act_prev = prev['act'] if (prev and prev['kind'] == 'conv' and prev['bn'] is None) else L.ACT_NONE
                if act_prev not in (L.ACT_NONE, L.ACT_RELU):
                    raise L.HipExtensionError('pixel shuffle behind a LeakyReLU conv: only ReLU / no activation')
                pl.relu_mask = 1 if act_prev == L.ACT_RELU else 0
                bk.add(L.OP_POOL, 'pool', pl)
                gcur, masked = gx, bool(pl.relu_mask)
                continue
            if r['kind'] == 'pool':
                gx = g32q(r['ch'], r['h'] * 2, r['w'] * 2)
                pl = L.esr_pool()
                pl.dtype, pl.mode, pl.B, pl.C, pl.H, pl.W = dt_e, 1, Bb, r['ch'], r['h'], r['w']
                pl.x, pl.y, pl.g, pl.gx = r['x'].view(0, r['ch']), r['y'].view(0, r['ch']), gcur.view(0, r['ch']), gx.view(0, r['ch'])
                prev = before(li)[0]
                pl.relu_mask = 1 if (prev and prev['kind'] == 'conv' and prev['act'] == L.ACT_RELU and prev['bn'] is None) else 0
                bk.add(L.OP_POOL, 'pool', pl)
                gcur, masked = gx, bool(pl.relu_mask)
                continue
            cout, cin_ = r['cout'], r['cin']
            alpha = r.get('alpha', 1.0) if r.get('res') is not None else 1.0
            if r.get('res') is not None:
                skips.setdefault(r['res'], []).append(gcur)       # d(out)/d(saved) = 1: the skip carries gcur as it is
            if r['bn'] is not None:
                gconv = g32q(((cout + 31) // 32) * 32, r['h'], r['w'])
                bnop = r['bnop']
                bk.add(L.OP_BN, 'bn', bnop(L.BN_BWD_REDUCE, g=gcur, sums=Q.sums_b.data_ptr() + 8 * r['mk']['base'], Bb=Bb, gb=gb))
                gbn = params_grad.get('bn%d' % r['ibn'])
                if gbn is not None and not fuse_bn():
                    o = bnop(L.BN_BWD_FINAL, sums=Q.sums_b.data_ptr() + 8 * r['mk']['base'], Bb=Bb, gb=gb)
                    o.dgamma, o.dbeta = gbn
                    bk.add(L.OP_BN, 'bn', o)
                o = bnop(L.BN_BWD_APPLY, g=gcur, gx=gconv, sums=Q.sums_b.data_ptr() + 8 * r['mk']['base'], Bb=Bb, gb=gb)
                if gbn is not None and fuse_bn():
                    o.dgamma, o.dbeta = gbn          # BWD_FINAL folded into the apply pass
                bk.add(L.OP_BN, 'bn', o)
                gpre = gconv
            else:
                if r['act'] != L.ACT_NONE and not masked:
                    raise RuntimeError('internal: activation mask of %s not applied' % r['key'])
                gpre = gcur
            # weight gradient
            gw = params_grad.get(r['key'])
            if gw is not None:
                wg = L.esr_wgrad()
                wg.dtype, wg.ks, wg.stride, wg.upsample = dt_e, r['ks'], r['st'], 1 if r.get('ups') else 0
                wg.B, wg.H, wg.W, wg.cout, wg.cin = Bb, r['h'], r['w'], cout, cin_
                wg.g, wg.in_ = gpre.view(0, cout), r['x'].view(0, cin_)
                wg.dw, wg.dbias, wg.scale = gw[0], gw[1], alpha
                if Q.tapmajor is not None and r['ks'] in (3, 4):
                    wg.dw, wg.tap_major = Q.tapmajor.slot(poff[r['key'] + '.weight'], cout, cin_, r['ks'] ** 2), 1
                # every layer owns its gradient buffers, so the weight gradient can run on the side stream
                # next to the dgrad chain (joined before the unpermute / at the end of the plan)
                # (no waits between these runs, several in flight: ESR_OPF_SIDE_FREE; each gets its own partial region)
                bk.add(L.OP_WGRAD, 'wgrad', wg, flags=L.OPF_SIDE | L.OPF_SIDE_FREE)
            # input gradient
            prev, tags = before(li)
            resid = [g_ for t_ in tags for g_ in skips.get(t_, [])]
            if len(resid) > 2:
                raise L.HipExtensionError('more than two skip connections end at the input of %s' % r['key'])
            gx = g32q(((cin_ + cpg - 1) // cpg) * cpg, r['hin'], r['win'])
            if r.get('ups'):
                # adjoint of (nearest x2 + 3x3 conv): a 4x4 / stride-2 conv over the gradient (esr_pack.ups_dgrad operand)
                c = E._conv(dt_e, Bb, r['hin'], r['win'], gpre.view(0), cout, None, de[r['key']], L.ACT_NONE, ks=4, stride=2)
            elif r['st'] == 1:
                c = E._conv(dt_e, Bb, r['hin'], r['win'], gpre.view(0), cout, None, de[r['key']], L.ACT_NONE)
            else:
                c = E._conv(dt_e, Bb, r['hin'], r['win'], gpre.view(0), cout, None, de[r['key']], L.ACT_NONE,
                            ks=4, stride=1, upsample=2)
            c.bias = None
            need_mask = prev is not None and prev['kind'] == 'conv' and prev['bn'] is None and prev['act'] != L.ACT_NONE
            if r['st'] == 2 and r['ks'] == 4 and not need_mask and r['w'] <= 4:
                # the transposed conv of the DEEPEST stride-2 layer (8x8 output): packed tiles + split K (esr_conv.ksplit),
                # K = forward couts.  Only there: the fp32 slabs of the split grow with the OUTPUT map, and on the 16^2 /
                # 32^2 outputs their write + read (67 MB per launch) costs more than the split saves (measured:
                # 57 -> 102 us and 33 -> 100 us; profiles/r04_experiments.md)
                ksp = s2_ksplit(Bb, r['w'], r['h'], cout, cin_, dt_e)
                if ksp:
                    ws = torch.empty(ksp * Bb * r['hin'] * r['win'] * ((cin_ + 31) // 32) * 32, dtype=torch.float32, device=dev)
                    Q.keep.append(ws)
                    c.ksplit, c.split_ws = ksp, ws.data_ptr()
            c.alpha = alpha                                   # (backward epilogue: v = acc * alpha [+ res1] [+ res2])
            if resid:
                c.res1 = resid[0].view(0, cin_)
                if len(resid) > 1:
                    c.res2, c.beta = resid[1].view(0, cin_), 1.0
            if need_mask:
                c.mask, c.out2, c.mask_cb_begin = prev['y'].view(0, cin_), gx.view(0, cin_), 0
                c.mask_act = prev['act']
            else:
                c.out = gx.view(0, cin_)
            bk.add_conv(c)
            gcur, masked = gx, need_mask
        if Q.tapmajor is not None:
            up = Q.tapmajor.op()
            if up is not None:
                bk.add(L.OP_UNPERMUTE, 'unpermute', up)
        Q.gx_tensor = torch.empty(Bb, cin0, H, W, dtype=torch.float32, device=dev)
        aff = None
        if input_affine is not None:
            aff = (input_affine[0], input_affine[1])
        Q.gx_op = _layout(bk, dt_e, 0, Bb, cin0, gcur, nchw_ptr=Q.gx_tensor.data_ptr(), affine=aff)
        Q.has_bn = nbn > 0
        Q.wgrad_arena = E.attach_wgrad_arena(bk, dev, exclusive=True)
        return Q

    build_backward(P, Bb0, want_wgrad)
    if dual is not None:
        # the G step's pass: input gradient of the first `dual` images, parameters frozen
        P.second = build_backward(BwdPass(), dual, False)
        # what a SECOND forward call over the same batch (same weights) adds to the BatchNorm buffers: the groups'
        # momentum updates in reverse order (the reference's netD(real), netD(fake) after netD(fake), netD(real))
        P.restat = L.OpList()
        for r in recs:
            if r['kind'] == 'conv' and r['bn'] is not None:
                P.restat.add(L.OP_BN, 'bn', r['bnop'](L.BN_RESTAT))
    return P


def split_forward_groups(P, n):
    """The forward launch list of a two-group pair plan (batch 2n, BatchNorm groups (0, 1)) as TWO lists over the halves,
    so that the half whose input is known early (the train step's ``real`` operand, group 1) can run long before the
    other one exists (round 5: netD(real) under the generator's forward instead of behind it).  Every launch of the
    full list is copied with B = n, its buffers advanced to the half's images, BatchNorm launches with groups = 1 and
    the statistics arrays advanced to the group's [C] rows.  The reference's order of running-statistics updates is
    (group 0, group 1): the EARLY half (group 1) therefore leaves the running buffers alone and the update it owes is a
    third list, ``P.restat1`` (ESR_BN_RESTAT on group 1's sums), to be run once after the late half.
    Returns nothing; fills P.fwd_half = [list of group 0, list of group 1] and P.restat1."""
    import ctypes as C_
    src = P.fwd.array()
    halves = [L.OpList(), L.OpList()]
    restat1 = L.OpList()
    g32_fields = {L.OP_CONV: ('conv', ('in_', 'out', 'aux_out', 'res1', 'res2', 'z1', 'z2', 'z3', 'mask', 'out2', 'out3')),
                  L.OP_BN: ('bn', ('x', 'y', 'g', 'gx')), L.OP_POOL: ('pool', ('x', 'y', 'g', 'gx')),
                  L.OP_LAYOUT: ('layout', ('g32',))}

    def adv(v, g):
        if v.ptr:
            v.ptr = v.ptr + g * n * v.batch_stride

    for g in (0, 1):
        for i in range(len(P.fwd.ops)):
            o = L.esr_op.from_buffer_copy(src[i])
            k = o.kind
            if k in g32_fields:
                st = getattr(o.u, g32_fields[k][0])
                for fld in g32_fields[k][1]:
                    adv(getattr(st, fld), g)
                assert st.B == 2 * n, 'pair plan: every launch covers both halves'
                st.B = n
                if k == L.OP_CONV:
                    if st.nchw_out:
                        raise L.HipExtensionError('split pair forward: NCHW conv outputs are not supported')
                    if st.stat_sums:
                        assert st.stat_groups == 2
                        st.stat_sums = st.stat_sums + g * 2 * st.stat_C * 8
                        st.stat_groups = 1
                elif k == L.OP_BN:
                    assert st.groups == 2
                    st.groups = 1
                    st.sums = st.sums + g * 2 * st.C * 8
                    st.mean = st.mean + g * st.C * 4
                    st.invstd = st.invstd + g * st.C * 4
                    if g == 1:
                        st.running_mean = st.running_var = st.num_batches_tracked = None
                elif k == L.OP_LAYOUT:
                    if st.nchw and i != P.in_op:
                        st.nchw = st.nchw + g * n * st.C * st.H * st.W * 4
            elif k == L.OP_LINEAR:
                st = o.u.linear
                assert st.B == 2 * n and st.mode == 0
                st.B = n
                st.x = st.x + g * n * st.I * 4
                st.y = st.y + g * n * st.O * 4
            else:
                raise L.HipExtensionError('split pair forward: launch kind %d is not supported' % k)
            halves[g].ops.append(o)
    for i in range(len(P.restat.ops)):
        o = L.esr_op.from_buffer_copy(P.restat.array()[i])
        st = o.u.bn
        assert o.kind == L.OP_BN and st.groups == 2 and st.B == 2 * n
        st.B, st.groups = n, 1
        st.sums = st.sums + 2 * st.C * 8
        restat1.ops.append(o)
    P.fwd_half, P.restat1 = halves, restat1


def _run_pass(Q, graph, gy, n_total, want_gx, needs):
    """Replay one backward pass (a SeqPlan's own or its BwdPass) for the upstream gradient gy; returns
    (gx or None, [parameter gradients or None])."""
    st = E.current_stream()
    nb_ = Q.gy_tensor.shape[0]              # images the pass covers (a pair's first half, or all)
    Q.gy_tensor.copy_(gy.detach()[:nb_].reshape(Q.gy_tensor.shape))
    Q.sums_b.zero_()
    if Q.grad_flat is not None:
        Q.grad_flat.zero_()
        if Q.tapmajor is not None:
            Q.tapmajor.tm.zero_()
    if graph:
        Q.bwd.graph_launch(st)
    else:
        lo = Q.bwd.array()[Q.gx_op].u.layout
        lo.nchw, lo.accumulate = Q.gx_tensor.data_ptr(), 0
        Q.bwd.run(st)
    gx = None
    if want_gx:
        gx = Q.gx_tensor.clone()
        if nb_ < n_total:                    # detached second half: zero gradient
            gx = torch.cat([gx, gx.new_zeros((n_total - nb_,) + tuple(gx.shape[1:]))])
    grads = [None] * len(needs)
    if Q.grad_flat is not None:
        flat = Q.grad_flat.clone()
        off = 0
        for i, (numel, shape) in enumerate(Q.grad_views):
            if needs[i]:
                grads[i] = flat[off:off + numel].view(shape)
            off += numel
    return gx, grads


def run_pass_into(Q, gx_into=None, accumulate=False):
    """The hand-written train step's face of a backward pass: the upstream gradient is ALREADY in Q.gy_tensor (the loss
    kernel wrote it there), the input gradient goes to ``gx_into`` (NCHW fp32; ``accumulate``: added to what it holds)
    or is dropped into the pass's own buffer, the parameter gradients stay in Q.grad_flat (OIHW order of the plan's
    parameter list).  No copies, no clones."""
    st = E.current_stream()
    if getattr(Q, 'has_bn', True):
        Q.sums_b.zero_()
    if Q.grad_flat is not None:
        Q.grad_flat.zero_()
        if Q.tapmajor is not None:
            Q.tapmajor.tm.zero_()
    lo = Q.bwd.array()[Q.gx_op].u.layout
    lo.nchw = (gx_into if gx_into is not None else Q.gx_tensor).data_ptr()
    lo.accumulate = 1 if (accumulate and gx_into is not None) else 0
    Q.bwd.run(st)


class SeqNetFn(torch.autograd.Function):
    """One autograd node for a whole feed-forward plan."""

    @staticmethod
    def forward(ctx, x, mod, *params):
        out, lease = mod._run_forward(x, need_bwd=True, **getattr(mod, '_pair_opts', {}))
        ctx.mod, ctx.lease, ctx.n = mod, lease, len(params)
        return out

    @staticmethod
    def backward(ctx, gy):
        P = ctx.lease.plan
        if P is None:
            raise RuntimeError('backward called twice on the same forward (retain_graph unsupported)')
        gx, grads = _run_pass(P, getattr(P, 'graph', False), gy, gy.shape[0], ctx.needs_input_grad[0],
                              ctx.needs_input_grad[2:])
        ctx.lease.release()
        return (gx, None) + tuple(grads)


class SharedPass:
    """Handle of ONE forward over [a; b] whose activations serve two backward passes (build_seq_plan(dual=...)):
    the pass that gave `a` its gradient (first_fn) and `second_pass()`, which re-issues the same outputs attached to
    the module's parameters — what a second pair of forward calls with unchanged weights would compute
    (SRRaGAN_model.py:150-151 after 133-134) — and applies that second pair's BatchNorm buffer updates."""

    def __init__(self, mod, lease, out, n):
        self.mod, self.lease, self.out, self.n = mod, lease, out, n
        self.pending = 2

    def done(self):
        self.pending -= 1
        if self.pending <= 0 and self.lease is not None:
            self.lease.release()
            self.lease = None

    def second_pass(self):
        """(pred_b, pred_a): the second pair in the reference's order (real first), differentiable w.r.t. the module's
        parameters only."""
        if self.lease is None or self.lease.plan is None:
            raise RuntimeError('SharedPass.second_pass: the forward\'s activations were released')
        y = SharedSecondFn.apply(self, *[t for _, t in self.mod._pspec()])
        return y[self.n:], y[:self.n]

    def __del__(self):
        if self.lease is not None:
            self.lease.release()


class SharedFirstFn(torch.autograd.Function):
    """netD over [a; b] in one pass (BatchNorm statistics per half); backward: input gradient of `a` only."""

    @staticmethod
    def forward(ctx, a, b, mod, holder):
        n = a.shape[0]
        x = torch.cat([a.detach(), b.detach()])
        out, lease = mod._run_forward(x, need_bwd=True, groups=2 if mod._has_bn else 1, dual=n)
        h = SharedPass(mod, lease, out, n)
        holder.append(h)
        ctx.h = h
        return out

    @staticmethod
    def backward(ctx, gy):
        h = ctx.h
        if h is None or h.lease is None or h.lease.plan is None:
            raise RuntimeError('backward called twice on the same forward (retain_graph unsupported)')
        P = h.lease.plan
        gx, _ = _run_pass(P.second, False, gy, h.n, True, ())
        ctx.h = None
        h.done()
        return gx, None, None, None


class SharedSecondFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, *params):
        P = h.lease.plan
        if P.restat is not None and P.restat.ops:
            P.restat.run(E.current_stream())
        ctx.h, ctx.n = h, len(params)
        return h.out.clone()

    @staticmethod
    def backward(ctx, gy):
        h = ctx.h
        if h is None or h.lease is None or h.lease.plan is None:
            raise RuntimeError('backward called twice on the same forward (retain_graph unsupported)')
        P = h.lease.plan
        _, grads = _run_pass(P, False, gy, gy.shape[0], False, ctx.needs_input_grad[1:])
        ctx.h = None
        h.done()
        return (None,) + tuple(grads)
