"""Fused multi-tensor Adam for the ESRGAN+ train step.

The reference builds two ``torch.optim.Adam`` instances over ~770 (G) and 42 (D) parameter tensors
(SRRaGAN_model.py:77-91) and steps them every iteration; on a GPU that is ~8 ``foreach`` launches per
optimizer plus a per-parameter gradient un-scaling pass under loss scaling.  ``FusedAdam`` is a
drop-in ``torch.optim.Optimizer`` (so ``lr_scheduler.MultiStepLR`` and ``state_dict`` plumbing keep
working, train.py:102 / base_model.py:35-40) whose ``step`` is ONE HIP launch (``esr_adam_step``):
moments live in two flat fp32 buffers laid out like the flat gradient buffer the fused backward nodes
emit, the loss-scale division is folded in (``grad_scale``), parameters stay ordinary leaf tensors.
Same arithmetic as torch.optim.Adam (amsgrad off, L2 weight decay)."""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib as L
from . import engine as E


def adam_tables(sizes):
    """Host-side launch tables: (goff per tensor, blocks [(entry, first_elem)]).  Every element of
    every tensor is covered by exactly one block of at most ADAM_BLOCK_ELEMS elements."""
    goff, blocks, off = [], [], 0
    for e, n in enumerate(sizes):
        goff.append(off)
        off += n
        for first in range(0, n, L.ADAM_BLOCK_ELEMS):
            blocks.append((e, first))
    return goff, blocks, off


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._g = {}          # per param group: device tables + flat moment buffers

    def _group_state(self, gi, params):
        st = self._g.get(gi)
        # cheap fingerprint first (three sampled storages + the count): the full per-tensor signature costs ~0.2 ms of
        # host time per step for the generator's ~770 tensors and only changes when the module moved (.to / .cuda)
        n = len(params)
        fp = (n, params[0].data_ptr(), params[n // 2].data_ptr(), params[-1].data_ptr())
        if st is not None and st.get('fp') == fp:
            st['calls'] = st.get('calls', 0) + 1
            if st['calls'] % 64:          # every 64th step the full signature is compared (a single parameter re-pointed
                return st                 # by hand would otherwise leave raw pointers to its old storage in the tables)
        sig = tuple((p.data_ptr(), p.numel()) for p in params)
        if st is not None and st['sig'] == sig:
            st['fp'] = fp
            return st
        dev = params[0].device
        E.require_cuda(params[0], 'FusedAdam parameter')
        for p in params:
            if p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                raise L.HipExtensionError('FusedAdam needs contiguous fp32 parameters on one device')
        sizes = [p.numel() for p in params]
        goff, blocks, total = adam_tables(sizes)
        ent = np.zeros((len(params), 3), dtype=np.int64)
        for i, p in enumerate(params):
            ent[i] = (p.data_ptr(), goff[i], sizes[i])
        blk = np.asarray(blocks, dtype=np.int32).reshape(-1, 2)
        old = st or {}
        st = dict(sig=sig, fp=fp, total=total, goff=goff, sizes=sizes,
                  entries=torch.from_numpy(ent).to(dev), blocks=torch.from_numpy(blk).to(dev),
                  nblocks=len(blocks),
                  exp_avg=old.get('exp_avg') if old.get('total') == total else torch.zeros(total, device=dev),
                  exp_avg_sq=old.get('exp_avg_sq') if old.get('total') == total else torch.zeros(total, device=dev),
                  step=old.get('step', 0) if old.get('total') == total else 0, stage=None,
                  applied=old.get('applied') if old.get('total') == total else None)
        self._g[gi] = st
        return st

    def _flat_grad(self, st, params):
        """The gradients as one flat fp32 tensor in parameter order: zero-copy when they already are
        views tiling one buffer in that order (what the fused backward nodes emit), else staged."""
        g0 = params[0].grad
        fpr = st.get('flat_fp')
        if (fpr is not None and g0 is fpr[0] and params[len(params) // 2].grad is fpr[1] and params[-1].grad is fpr[2]
                and g0.data_ptr() == fpr[3]):
            return fpr[4]       # the same persistent views as the last verified step (a module's flat gradient store)
        if g0 is None:
            # torch.optim.Adam skips a parameter whose grad is None; the flat kernel cannot leave single tensors
            # out of its tables, so this has to be said instead of silently moving them by their momentum
            raise L.HipExtensionError('FusedAdam: a parameter of the group has no gradient while others do; '
                                      'freeze it (requires_grad=False) or give it its own optimizer')
        base = g0.storage_offset()
        ok = g0.dtype == torch.float32
        if ok:
            sp = g0.untyped_storage().data_ptr()
            for p, off in zip(params, st['goff']):
                g = p.grad
                if (g is None or g.dtype != torch.float32 or not g.is_contiguous()
                        or g.untyped_storage().data_ptr() != sp or g.storage_offset() != base + off):
                    ok = False
                    break
        if ok:
            flat = torch.empty(0, dtype=torch.float32, device=g0.device).set_(
                g0.untyped_storage(), base, (st['total'],))
            st['flat_fp'] = (g0, params[len(params) // 2].grad, params[-1].grad, g0.data_ptr(), flat)
            return flat
        st['flat_fp'] = None
        if st['stage'] is None:
            st['stage'] = torch.zeros(st['total'], device=params[0].device)
        views = [st['stage'][o:o + n].view_as(p) for p, o, n in zip(params, st['goff'], st['sizes'])]
        if any(p.grad is None for p in params):
            raise L.HipExtensionError('FusedAdam: a parameter of the group has no gradient while others do; '
                                      'freeze it (requires_grad=False) or give it its own optimizer')
        grads = [p.grad for p in params]
        torch._foreach_copy_(views, [g.to(torch.float32) for g in grads])
        return st['stage']

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0, scaler=None, flat_grad=None):
        """``grad_scale`` multiplies every gradient on the fly (1/loss_scale under static loss scaling).
        ``scaler`` (DynamicLossScaler): the gradients are checked for inf / nan first and the whole update is
        skipped on the device when one is found; otherwise they are divided by the current scale.
        ``flat_grad`` (single parameter group): the gradients of all parameters as ONE fp32 tensor in parameter order —
        what a fused backward plan leaves behind — instead of the parameters' ``.grad`` (which need not be set)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        stream = E.current_stream()
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group['params'] if p.requires_grad]
            if not params or (flat_grad is None and not any(p.grad is not None for p in params)):
                continue
            st = self._group_state(gi, params)
            if flat_grad is not None:
                if len(self.param_groups) != 1 or flat_grad.numel() != st['total'] or flat_grad.dtype != torch.float32:
                    raise L.HipExtensionError('FusedAdam.step(flat_grad=...): one parameter group, %d fp32 elements' % st['total'])
                flat = flat_grad
            else:
                flat = self._flat_grad(st, params)
            st['step'] += 1                 # steps ATTEMPTED (== applied without a scaler)
            b1, b2 = group['betas']
            slot = 0
            if scaler is not None:
                # per-optimizer overflow flag and a DEVICE count of the steps really applied (torch.amp.GradScaler:
                # found_inf is tracked per optimizer, and a skipped step does not advance Adam's bias correction)
                slot = scaler.slot_of((id(self), gi))
                if st.get('applied') is None:
                    st['applied'] = torch.full((1,), float(st['step'] - 1), dtype=torch.float32, device=flat.device)
                scaler.check(flat, stream, slot)
                scaler.count(st['applied'], stream, slot)
            a = L.esr_adam()
            a.entries, a.blocks, a.nblocks = st['entries'].data_ptr(), st['blocks'].data_ptr(), st['nblocks']
            a.grad, a.exp_avg, a.exp_avg_sq = flat.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr()
            a.lr, a.beta1, a.beta2, a.eps = group['lr'], b1, b2, group['eps']
            a.bc1, a.bc2 = 1.0 - math.pow(b1, st['step']), 1.0 - math.pow(b2, st['step'])
            a.grad_scale, a.weight_decay = grad_scale, group['weight_decay']
            if scaler is not None:
                a.amp_state, a.amp_slot = scaler.state.data_ptr(), slot
                a.step_count, a.beta1_d, a.beta2_d = st['applied'].data_ptr(), float(b1), float(b2)
            L.check(L.lib().esr_adam_step(C.byref(a), C.c_void_p(stream)), 'esr_adam_step')
        return loss

    # ---- checkpoint plumbing: the reference stores `optimizer.state_dict()` of torch.optim.Adam in its
    # `.state` files and feeds them back through `load_state_dict` (base_model.py:66-85).  The moments
    # live in flat buffers here, so both directions translate to / from Adam's per-parameter layout.
    def state_dict(self):
        sd = super().state_dict()                       # param_groups with index lists; state = {}
        state, idx = {}, 0
        for gi, group in enumerate(self.param_groups):
            st = self._g.get(gi)
            live = [p for p in group['params'] if p.requires_grad]
            pos = {id(p): k for k, p in enumerate(live)}
            for p in group['params']:
                if st is not None and id(p) in pos and st['step'] > 0:
                    k = pos[id(p)]
                    o, n = st['goff'][k], st['sizes'][k]
                    # under dynamic loss scaling only the APPLIED steps count (read back here, at checkpoint time)
                    nstep = float(st['applied'].item()) if st.get('applied') is not None else float(st['step'])
                    state[idx] = {'step': torch.tensor(nstep),
                                  'exp_avg': st['exp_avg'][o:o + n].view_as(p).clone(),
                                  'exp_avg_sq': st['exp_avg_sq'][o:o + n].view_as(p).clone()}
                idx += 1
        sd['state'] = state
        return sd

    def load_state_dict(self, state_dict):
        state = state_dict.get('state', {})
        super().load_state_dict({'state': {}, 'param_groups': state_dict['param_groups']})
        idx = 0
        for gi, group in enumerate(self.param_groups):
            live = [p for p in group['params'] if p.requires_grad]
            pos = {id(p): k for k, p in enumerate(live)}
            st = self._group_state(gi, live) if live else None
            for p in group['params']:
                ent = state.get(idx, state.get(str(idx)))
                if ent is not None and st is not None and id(p) in pos:
                    k = pos[id(p)]
                    o, n = st['goff'][k], st['sizes'][k]
                    st['exp_avg'][o:o + n].copy_(ent['exp_avg'].reshape(-1).to(st['exp_avg'].device, torch.float32))
                    st['exp_avg_sq'][o:o + n].copy_(ent['exp_avg_sq'].reshape(-1).to(st['exp_avg'].device, torch.float32))
                    st['step'] = int(float(ent['step']))
                    st['applied'] = None           # re-seeded from `step` at the next scaled step
                idx += 1



class DynamicLossScaler:
    """Loss scaling for the fp16 training path with overflow skip, entirely on the device (no ``.item()``):
    ``loss * scaler.scale`` before ``backward()``; ``optimizer.step(scaler=scaler)`` checks the gradients and
    skips the update when they hold an inf / nan; ``update()`` once per iteration halves the scale after an
    overflow and doubles it after ``interval`` clean steps (the torch.amp.GradScaler policy)."""

    def __init__(self, device, init_scale=1024.0, growth=2.0, backoff=0.5, interval=2000):
        # {scale, -, good steps, -, found[0..3]}: one overflow flag per optimizer (include/esrgan_hip.h, esr_amp)
        self.state = torch.tensor([init_scale, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0], dtype=torch.float32, device=device)
        self.growth, self.backoff, self.interval = growth, backoff, interval
        self._slots = {}

    def slot_of(self, key):
        s = self._slots.get(key)
        if s is None:
            if len(self._slots) >= 4:
                raise L.HipExtensionError('DynamicLossScaler tracks at most 4 optimizers / parameter groups')
            s = self._slots[key] = len(self._slots)
        return s

    @property
    def scale(self):
        return self.state[0]            # a 0-dim DEVICE tensor: multiplying the loss by it needs no sync

    def check(self, flat, stream=None, slot=0):
        a = L.esr_amp()
        a.mode, a.state, a.grad, a.n, a.slot = L.AMP_CHECK, self.state.data_ptr(), flat.data_ptr(), flat.numel(), slot
        L.check(L.lib().esr_amp_step(C.byref(a), C.c_void_p(stream or E.current_stream())), 'esr_amp_step')

    def count(self, step_count, stream=None, slot=0):
        """step_count[0] += 1 unless this optimizer's gradients overflowed (the step esr_adam_step is about to apply)."""
        a = L.esr_amp()
        a.mode, a.state, a.slot, a.step_count = L.AMP_COUNT, self.state.data_ptr(), slot, step_count.data_ptr()
        L.check(L.lib().esr_amp_step(C.byref(a), C.c_void_p(stream or E.current_stream())), 'esr_amp_step')

    def update(self, stream=None):
        a = L.esr_amp()
        a.mode, a.state, a.interval = L.AMP_UPDATE, self.state.data_ptr(), self.interval
        a.growth, a.backoff = self.growth, self.backoff
        L.check(L.lib().esr_amp_step(C.byref(a), C.c_void_p(stream or E.current_stream())), 'esr_amp_step')
