"""Glue between the drop-in nn.Modules and the recorded HIP launch plans."""
import os

import torch

from . import engine as E
from . import _lib as L


def _draw_seed():
    """Philox key of one training forward: torch's CPU generator (reproducible under torch.manual_seed,
    utils/util.py:43-47) mixed with the data-parallel rank — every rank seeds torch identically
    (set_random_seed), and identical keys would inject the SAME GaussianNoise field into the ranks' different
    local batches instead of independent noise over the global batch."""
    s = int(torch.randint(0, 2 ** 62, (1,)).item())
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        s = (s ^ (torch.distributed.get_rank() * 0x9E3779B97F4A7C15)) & (2 ** 62 - 1)
    return s


def _needs_grad(module, x):
    if not torch.is_grad_enabled():
        return False
    # (a nn.DataParallel replica holds its weights as plain attributes — `parameters()` is empty there — so ask the
    # attribute-access list the launches use)
    ps = module._convs()[1] if hasattr(module, '_convs') else list(module.parameters())
    return x.requires_grad or any(p.requires_grad for p in ps)


def _prep_input(x, what):
    E.require_cuda(x, what)
    if x.dim() != 4:
        raise ValueError('%s must be NCHW, got shape %s' % (what, tuple(x.shape)))
    if x.shape[2] == 0 or x.shape[3] == 0:
        # torch's Conv2d rejects these too ("Kernel size can't be greater than actual input size")
        raise ValueError('%s has an empty spatial extent: shape %s' % (what, tuple(x.shape)))
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
    if x.dim() == 4 and x.shape[0] == 0:
        E.require_cuda(x, 'input')
        return x.new_zeros(tuple(x.shape), dtype=torch.float32)
    if _needs_grad(mod, x):
        E.require_cuda(x, 'input')
        B, C_, H, W = x.shape
        if C_ != 64:
            raise ValueError('expected 64 input channels, got %d' % C_)
        has_noise = getattr(mod, 'noise', None) is not None if kind == 'rdb' else True
        noise = bool(mod.training and has_noise)
        n_noise = 0 if not noise else (1 if kind == 'rdb' else (4 if mod.variant == 'test_image' else 3))
        zs = _zs_list(z, n_noise, (B, 64, H, W), x.device) if noise else None
        return _BlockFn.apply(x, mod, kind, noise, zs, *mod._convs()[1])
    xin = _prep_input(x, 'input')
    B, C_, H, W = xin.shape
    if C_ != 64:
        raise ValueError('expected 64 input channels, got %d' % C_)
    order = E.StreamOrder.of(mod)
    cur = order.enter()
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
    order.leave(cur)
    return out


def _train_forward(tp, xin, st, seed, zs):
    if tp.graph:
        tp.x_static.copy_(xin)
        tp.seed_t.fill_(seed)
        if tp.fwd.streams is not None:          # the chain's weight streams are gathered outside the graph
            tp.fwd.streams.ensure(st)
        tp.fwd.ops.graph_launch(st)
        return tp.out_static.clone()
    out = torch.empty(tp.fwd.out_shape, dtype=torch.float32, device=xin.device)
    tp.fwd.run(xin, out, st, seed, zs)
    return out


def _train_backward(tp, gy, st, noise, explicit, seed, want_gx, sync=None, prepared=False, follow_wgs=None):
    """Runs the backward launch list; returns dL/dx (stand-alone blocks) or None.
    follow_wgs: workgroups of the follower pass of weight gradients for THIS run (None: the plan's default); the
    results do not depend on it.
    prepared: the gradient buffers are zeroed and the backward chain's weight streams gathered already
    (`rrdbnet_train_prepare`, on another stream the caller has ordered in front of this one)."""
    if not prepared:
        tp.grad_flat.zero_()
        if tp.tapmajor is not None:
            tp.tapmajor.tm.zero_()
    if tp.graph:
        tp.gy_static.copy_(gy)
        if tp.bwd_streams is not None:
            tp.bwd_streams.ensure(st)
        tp.bwd.graph_launch(st)                 # seed_t still holds this step's seed
        return tp.gx_static.clone() if (want_gx and tp.gx_op is not None) else None
    arr = tp.bwd.array()
    arr[tp.gy_op].u.layout.nchw = gy.data_ptr()
    gx = None
    n_ops = len(tp.bwd.ops)
    if tp.gx_begin is not None and not want_gx:
        n_ops = tp.gx_begin                         # whole generator: fea_conv's input gradient only on request
    elif tp.gx_op is not None:
        if arr[tp.gx_op].kind == L.OP_CONV:
            cv = arr[tp.gx_op].u.conv
            gx = torch.empty((cv.B, cv.nchw_out_c, cv.H, cv.W), dtype=torch.float32, device=gy.device)
            cv.nchw_out = gx.data_ptr()
        else:
            lo = arr[tp.gx_op].u.layout
            gx = torch.empty((lo.B, lo.C, lo.H, lo.W), dtype=torch.float32, device=gy.device)
            lo.nchw = gx.data_ptr()
    mode = L.NOISE_OFF
    if noise:
        mode = L.NOISE_EXPLICIT if explicit else L.NOISE_PHILOX
    for i in tp.bwd_noise_ops:
        arr[i].u.conv.noise_mode = mode
        arr[i].u.conv.seed = seed
    if tp.follow_op is not None:
        arr[tp.follow_op].u.rdb_wgrad.max_workgroups = (max(32, min(int(follow_wgs), tp.follow_spare)) if follow_wgs
                                                        else tp.follow_wgs)
    for i in tp.bwd_chain_ops:                      # fused backward chain (esr_rdb_backward): same Philox key as the forward
        arr[i].u.rdb_chain.noise_mode = L.NOISE_PHILOX if noise else L.NOISE_OFF
        arr[i].u.rdb_chain.seed = seed
    if tp.bwd_streams is not None and not prepared:
        tp.bwd_streams.ensure(st)
    if tp.segments is None or sync is None:
        tp.bwd.run_range(st, 0, n_ops)
        return gx
    # data-parallel: hand every finished slice of the flat gradient buffer to `sync` (an asynchronous
    # all-reduce, dp.GradExchange) as soon as the ops that produce it are enqueued — slices are merged up to
    # sync.bucket_elems so that a collective moves enough bytes per xGMI link — and wait for all of them
    # (stream-side) before the gradients leave the node
    i0, pend_lo, pend_hi, handles = 0, None, None, []
    limit = getattr(sync, 'bucket_elems', 0)
    for op_end, lo, hi in tp.segments:
        tp.bwd.run_range(st, i0, op_end)
        i0 = op_end
        if pend_lo is not None and (hi != pend_lo and lo != pend_hi):
            handles.append(sync(tp.grad_flat[pend_lo:pend_hi]))
            pend_lo = None
        pend_lo, pend_hi = (lo, hi) if pend_lo is None else (min(lo, pend_lo), max(hi, pend_hi))
        if pend_hi - pend_lo >= limit:
            handles.append(sync(tp.grad_flat[pend_lo:pend_hi]))
            pend_lo = None
    tp.bwd.run_range(st, i0, n_ops)
    if pend_lo is not None:
        handles.append(sync(tp.grad_flat[pend_lo:pend_hi]))
    if hasattr(sync, 'wait_handles'):
        sync.wait_handles(handles)                  # dp.GradExchange (optionally timed: bench.py dp_train)
    else:
        for h in handles:
            if h is not None:
                h.wait()
    return gx


def _grad_views(tp):
    """Parameter gradients as views of a fresh copy of the plan's flat gradient buffer (the plan's own
    buffer is overwritten by the next backward).  One split + one view per tensor: building ~770
    slices by hand cost 3 ms of host time per step."""
    meta = tp.__dict__.get('_grad_meta')
    if meta is None:
        sizes, shapes = [], []
        for gw, gb in tp.grad_views:
            sizes.append(gw.numel())
            shapes.append(gw.shape)
            if gb is not None:
                sizes.append(gb.numel())
                shapes.append(gb.shape)
        meta = tp._grad_meta = (sizes, shapes)
    flat = tp.grad_flat.clone()
    return [t.view(sh) for t, sh in zip(flat.split(meta[0]), meta[1])]


MAX_TRAIN_SHAPES = 4    # shapes whose training plans (saved activations) a module keeps


class _PlanLease:
    """Marks a TrainPlan busy while an autograd graph still references its saved activations."""

    def __init__(self, tp):
        self.tp = tp
        tp.busy = True

    def release(self):
        if self.tp is not None:
            self.tp.busy = False
            self.tp = None

    def __del__(self):
        self.release()


def _train_plan(net, wp, dp, B, H, W, dev, noise, explicit):
    sync = getattr(net, '_grad_sync', None)      # dp.GradExchange attached: segmented backward, no graph replay
    key = ('train', B, H, W, net.precision, noise, explicit, wp.generation, str(dev), sync is not None)
    pool = net._plans.get(key)
    if pool is None:
        # every shape keeps a full set of saved activations alive: bound the number of shapes (LRU over the
        # train keys; a pool with a plan still referenced by a live autograd graph is never dropped)
        tkeys = [k for k in net._plans if isinstance(k, tuple) and k and k[0] == 'train']
        while len(tkeys) >= MAX_TRAIN_SHAPES:
            victim = next((k for k in tkeys if not any(tp.busy for tp in net._plans[k])), None)
            if victim is None:
                break
            del net._plans[victim]
            tkeys.remove(victim)
        pool = []
    else:
        del net._plans[key]              # re-insert below: most recently used last
    net._plans[key] = pool
    for tp in pool:
        if not tp.busy:
            tp.packs = (wp, dp)
            return tp
    tp = E.build_rrdbnet_train_plan(net, wp, dp, net.nb, net.in_nc, net.out_nc, B, H, W,
                                    net.precision, dev, noise, net.variant, explicit, segmented=sync is not None)
    if not explicit and E.use_graphs() and sync is None:
        tp.enable_graph((B, net.in_nc, H, W), dev)
    # The launch lists hold RAW pointers into the packed operands: the plan keeps the packs alive.  An autograd graph
    # references the plan (through its lease), not the module — and a nn.DataParallel replica (networks.py:105-107) is
    # gone when its forward returns: without this its input-gradient operands were freed memory by the time the backward
    # ran (found in round 5 with two replicas on one GPU: wrong gradients whenever the allocator re-used the block)
    tp.packs = (wp, dp)
    pool.append(tp)
    return tp


class _RRDBNetFn(torch.autograd.Function):
    """RRDBNet forward/backward as ONE autograd node (architecture.py:76-78 + the implicit
    autograd backward triggered at SRRaGAN_model.py:140)."""

    @staticmethod
    def forward(ctx, x, net, zs, proxy, *params):
        # proxy (a 1-element leaf that requires grad) stands for ALL parameters: their gradients then leave the node
        # through net._deliver_flat_grads instead of one autograd output each; `params` is empty in that mode
        xin = _prep_input(x, 'input')
        B, _, H, W = xin.shape
        dev = xin.device
        st = E.current_stream()
        wp = net._weights(dev)
        dp = net._dgrad_weights(dev)
        dp.ensure(st, force=not net._dgrad_fresh())
        noise = bool(net.training)
        tp = _train_plan(net, wp, dp, B, H, W, dev, noise, zs is not None)
        ctx.lease = _PlanLease(tp)
        ctx.seed = _draw_seed() if (noise and zs is None) else 0
        ctx.explicit = zs is not None
        ctx.noise = noise
        ctx.n_params = len(params)
        ctx.net = net if proxy is not None else None
        ctx.sync = getattr(net, '_grad_sync', None)
        return _train_forward(tp, xin, st, ctx.seed, zs)

    @staticmethod
    def backward(ctx, gy):
        tp = ctx.lease.tp
        if tp is None:
            raise RuntimeError('RRDBNet backward called twice (retain_graph is not supported: the '
                               'saved activations live in a reusable launch plan)')
        gy = gy.detach().contiguous().float()
        if ctx.net is not None:
            ctx.net._release_adopted(tp.grad_flat)      # `.grad` may alias the buffer this backward is about to rewrite
        gx = _train_backward(tp, gy, E.current_stream(), ctx.noise, ctx.explicit, ctx.seed,
                             bool(ctx.needs_input_grad[0]), ctx.sync)
        gx = gx if ctx.needs_input_grad[0] else None
        if ctx.net is not None:
            ctx.net._deliver_flat_grads(tp.grad_flat)
            ctx.lease.release()
            return (gx, None, None, None)
        grads = _grad_views(tp)
        ctx.lease.release()
        assert len(grads) == ctx.n_params
        grads = [g if need else None for g, need in zip(grads, ctx.needs_input_grad[4:])]
        return (gx, None, None, None) + tuple(grads)


class TrainPass:
    """State of one RRDBNet training forward outside autograd (train.ESRGANPlusStep's hand-written step)."""
    __slots__ = ('lease', 'seed', 'noise', 'explicit', 'sync', 'prepared', 'pack_ev')


_ADOPT = os.environ.get('ESR_ADOPT_GRADS', '1') != '0'     # A/B knob: 0 = always copy into the module's store (round 4)


def rrdbnet_train_forward(net, x, z=None):
    """RRDBNet.forward in training form (every activation the backward needs is kept) WITHOUT an autograd node:
    returns (y, state) for ``rrdbnet_train_backward``.  Same plans, same launches as ``_RRDBNetFn``."""
    xin = _prep_input(x, 'input')
    B, C_, H, W = xin.shape
    if C_ != net.in_nc:
        raise ValueError('expected %d input channels, got %d' % (net.in_nc, C_))
    dev = xin.device
    st = E.current_stream()
    noise = bool(net.training)
    per = 4 if net.variant == 'test_image' else 3
    zs = _zs_list(z, per * net.nb, (B, 64, H, W), dev) if noise else None
    wp = net._weights(dev)
    dp = net._dgrad_weights(dev)
    n_packs = getattr(dp, 'pack_count', 0)
    dp.ensure(st, force=not net._dgrad_fresh())
    pack_ev = None
    if getattr(dp, 'pack_count', 0) != n_packs and not torch.cuda.is_current_stream_capturing():
        # the input-gradient operands were re-packed on THIS stream just now (first step, ESR_PREPACK=0, after
        # load_state_dict / resume): whoever gathers the backward chain's weight streams from dp.arena on another
        # stream (`rrdbnet_train_prepare` on the train step's side stream) has to wait for it — recorded here, in
        # front of the forward's launches, so that the wait does not cover the forward
        pack_ev = torch.cuda.Event()
        pack_ev.record(torch.cuda.current_stream())
    tp = _train_plan(net, wp, dp, B, H, W, dev, noise, zs is not None)
    s = TrainPass()
    s.pack_ev = pack_ev
    s.lease = _PlanLease(tp)
    s.seed = _draw_seed() if (noise and zs is None) else 0
    s.noise, s.explicit, s.sync = noise, zs is not None, getattr(net, '_grad_sync', None)
    s.prepared = False
    return _train_forward(tp, xin, st, s.seed, zs), s


def rrdbnet_train_prepare(net, s):
    """What the backward of the forward state `s` needs before its first launch and that depends on nothing of this
    step — the flat gradient buffer and the tap-major arena zeroed (67 + 67 MB of fills), the backward chain's weight
    streams gathered from the input-gradient packs — enqueued on the CURRENT stream: the train step runs it on its side
    stream under the generator's forward and makes the main stream wait for it in front of the backward (three launches
    and their dispatch gaps off the step's critical path)."""
    tp = s.lease.tp
    if tp is None or tp.graph or s.prepared:
        return
    if s.pack_ev is not None:
        torch.cuda.current_stream().wait_event(s.pack_ev)     # the operands the gather below reads (see the forward)
    net._release_adopted(tp.grad_flat)
    tp.grad_flat.zero_()
    if tp.tapmajor is not None:
        tp.tapmajor.tm.zero_()
    if tp.bwd_streams is not None:
        tp.bwd_streams.ensure(E.current_stream())
    s.prepared = True


def rrdbnet_train_backward(net, s, gy, follow_wgs=None):
    """The backward of ``rrdbnet_train_forward``: parameter gradients into the module's flat store (every parameter's
    ``.grad`` = its view of it, block._PlannedModule._deliver_flat_grads); dL/dx is not formed."""
    tp = s.lease.tp
    if tp is None:
        raise RuntimeError('rrdbnet_train_backward called twice on one forward')
    if not s.prepared:
        net._release_adopted(tp.grad_flat)
    _train_backward(tp, gy, E.current_stream(), s.noise, s.explicit, s.seed, False, s.sync, prepared=s.prepared,
                    follow_wgs=follow_wgs)
    net._deliver_flat_grads(tp.grad_flat, adopt=_ADOPT)
    s.lease.release()


def _block_train_plan(mod, kind, wp, dp, B, H, W, dev, noise, explicit):
    key = ('train', kind, B, H, W, mod.precision, noise, explicit, wp.generation, str(dev))
    pool = mod._plans.setdefault(key, [])
    for tp in pool:
        if not tp.busy:
            tp.packs = (wp, dp)
            return tp
    tp = E.build_rrdbnet_train_plan(mod, wp, dp, 1, 64, 64, B, H, W, mod.precision, dev, noise,
                                    mod.variant, explicit, kind=kind)
    if not explicit and E.use_graphs():
        tp.enable_graph((B, 64, H, W), dev)
    tp.packs = (wp, dp)              # (the lists hold raw pointers into the packs: see _train_plan)
    pool.append(tp)
    return tp


class _BlockFn(torch.autograd.Function):
    """Stand-alone ResidualDenseBlock_5C / RRDB (block.py:260-268, 287-291) forward + backward as one
    autograd node; unlike the whole generator it also returns the gradient w.r.t. its input."""

    @staticmethod
    def forward(ctx, x, mod, kind, noise, zs, *params):
        xin = _prep_input(x, 'input')
        B, _, H, W = xin.shape
        dev = xin.device
        st = E.current_stream()
        wp = mod._weights(dev)
        dp = mod._dgrad_weights(dev)
        dp.ensure(st)
        tp = _block_train_plan(mod, kind, wp, dp, B, H, W, dev, noise, zs is not None)
        ctx.lease = _PlanLease(tp)
        ctx.seed = _draw_seed() if (noise and zs is None) else 0
        ctx.explicit, ctx.noise, ctx.n_params = zs is not None, noise, len(params)
        return _train_forward(tp, xin, st, ctx.seed, zs)

    @staticmethod
    def backward(ctx, gy):
        tp = ctx.lease.tp
        if tp is None:
            raise RuntimeError('backward called twice (retain_graph is not supported: the saved '
                               'activations live in a reusable launch plan)')
        gy = gy.detach().contiguous().float()
        # (a stand-alone block never adopts the plan's gradient buffer: autograd accumulates copies of the views)
        gx = _train_backward(tp, gy, E.current_stream(), ctx.noise, ctx.explicit, ctx.seed, True)
        grads = _grad_views(tp)
        ctx.lease.release()
        assert len(grads) == ctx.n_params
        grads = [g if need else None for g, need in zip(grads, ctx.needs_input_grad[5:])]
        return (gx if ctx.needs_input_grad[0] else None, None, None, None, None) + tuple(grads)


def _flat_grad_route(net, params):
    """True when the backward may hand the parameter gradients over as ONE flat buffer (`_deliver_flat_grads`)."""
    return bool(net.flat_param_grads) and all(p.requires_grad and p.is_leaf for p in params)


def run_rrdbnet(net, x, z=None):
    """RRDBNet.forward (architecture.py:76-78) on the HIP path."""
    if x.dim() == 4 and x.shape[0] == 0:          # empty batch: torch returns an empty result
        E.require_cuda(x, 'input')
        return x.new_zeros((0, net.out_nc, 4 * x.shape[2], 4 * x.shape[3]), dtype=torch.float32)
    if _needs_grad(net, x):
        E.require_cuda(x, 'input')
        B, C_, H, W = x.shape
        if C_ != net.in_nc:
            raise ValueError('expected %d input channels, got %d' % (net.in_nc, C_))
        per = 4 if net.variant == 'test_image' else 3
        zs = _zs_list(z, per * net.nb, (B, 64, H, W), x.device) if net.training else None
        params = net._convs()[1]
        # the flat route assigns `.grad` itself, which only means something on LEAF parameters: under a multi-device
        # nn.DataParallel (networks.py:105-107) a replica's weights are Broadcast outputs, and their gradient has to
        # travel back through autograd to the originals -> per-tensor outputs
        if _flat_grad_route(net, params):
            proxy = net.__dict__.get('_grad_proxy')
            if proxy is None or proxy.device != x.device:
                proxy = net.__dict__['_grad_proxy'] = torch.zeros(1, device=x.device, requires_grad=True)
            return _RRDBNetFn.apply(x, net, zs, proxy)
        net._flush_stale_grads()
        return _RRDBNetFn.apply(x, net, zs, None, *params)
    xin = _prep_input(x, 'input')
    B, C_, H, W = xin.shape
    if C_ != net.in_nc:
        raise ValueError('expected %d input channels, got %d' % (net.in_nc, C_))
    order = E.StreamOrder.of(net)
    cur = order.enter()
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
    order.leave(cur)
    return out
