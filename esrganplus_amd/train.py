"""ESRGAN+ training step on the HIP path — the call pattern of
``SRRaGANModel.optimize_parameters`` (codes/models/SRRaGAN_model.py:113-186) with the
hyper-parameters of codes/options/train/train_ESRGANplus.json:55-77: L1 pixel x0.01, L1 VGG-feature
x1, relativistic-average GAN (BCE-with-logits) x0.005, Adam(1e-4, betas (0.9, 0.999)) for G and D.

The three networks are the drop-in HIP modules; the losses are a handful of tiny reductions.  With
``torch.distributed`` initialised (one process per GPU) the step is data-parallel: gradient
exchanges are started right after each backward and waited for right before the matching
``optimizer.step()``, so the G exchange overlaps the whole D forward/backward (the D step only uses
``fake_H.detach()`` computed before the G update, exactly as in the reference).

Launch economy (the step is a few thousand small launches): each network sees its two operands in ONE pass
(``forward_pair``: per-operand BatchNorm statistics), the losses are single launches that also produce their
gradients (``losses``), and the D step is enqueued on a second stream under the G backward (one GPU and
data-parallel alike: the step that is measured on one GPU is the step that scales).
"""
import os

import torch

from . import dp as DP
from . import engine as E
from . import losses as LS
from .optim import FusedAdam, DynamicLossScaler


class ESRGANPlusStep:
    def __init__(self, netG, netD, netF, lr_G=1e-4, lr_D=1e-4, beta1_G=0.9, beta1_D=0.9,
                 pixel_weight=1e-2, feature_weight=1.0, gan_weight=5e-3, loss_scale=1.0, data_parallel=None):
        self.netG, self.netD, self.netF = netG, netD, netF
        self.l_pix_w, self.l_fea_w, self.l_gan_w = pixel_weight, feature_weight, gan_weight
        # data_parallel: None = follow torch.distributed (world size > 1); False inside a multi-rank job = this rank's
        # own step without any exchange (bench.py's no-exchange figure next to the data-parallel one)
        self.data_parallel = DP.active() if data_parallel is None else bool(data_parallel)
        # fp16 path: 'dynamic' (default policy of the scaler: start at 1024, halve on overflow and skip that step,
        # double after 2000 clean steps) or a fixed number (1.0 for fp32).  Nothing here synchronises with the host.
        self.scaler = None
        if loss_scale == 'dynamic':
            self.scaler = DynamicLossScaler(next(netG.parameters()).device)
            loss_scale = 1.0
        self.loss_scale = loss_scale
        # one fused launch per optimizer (optim.FusedAdam == torch.optim.Adam arithmetic; it stays a
        # torch.optim.Optimizer, so the reference's MultiStepLR schedulers attach unchanged)
        self.optimizer_G = FusedAdam([p for p in netG.parameters() if p.requires_grad],
                                     lr=lr_G, betas=(beta1_G, 0.999))
        self.optimizer_D = FusedAdam(netD.parameters(), lr=lr_D, betas=(beta1_D, 0.999))
        self.exG = DP.GradExchange(netG, enabled=self.data_parallel, measure=self.data_parallel)
        self.exD = DP.GradExchange(netD, enabled=self.data_parallel, measure=self.data_parallel)
        self.log = {}
        self.fake_H = None
        self._steps = 0
        # stream overlap of the step (ESR_TRAIN_OVERLAP): 0 = everything in sequence on the caller's stream (9.8 ms);
        # 1 (default) = netF(var_H) and the D step on a second stream (8.9 ms); 2 = additionally the G step's netD
        # pass on a third stream next to its netF pass, forward AND backward (autograd runs a node's backward on the
        # stream its forward ran on).  Measured, round 4 (profiles/r04_experiments.md): more concurrency is NOT faster
        # here — 2 costs 0.1-0.15 ms over 1, streams probed to be truly concurrent (engine.concurrent_streams,
        # ESR_STREAM_PROBE=1) 0.4-0.9 ms, GPU_MAX_HW_QUEUES=8 3-5 ms: a persistent chain launch (128 lock-stepped
        # workgroups exchanging halos) that shares the chip with other launches runs at the pace of its most-delayed
        # tile, and the small launches slow each other 2-5x.  What mode 1 overlaps is what the default stream's
        # hardware queue lets through.
        self.overlap = self._knob_int('ESR_TRAIN_OVERLAP', '1', (0, 1, 2))
        # ESR_SHARED_D=0: the D step runs its own forward pair (round 3) instead of re-using the G step's pass
        self.shared_d = os.environ.get('ESR_SHARED_D', '1') != '0'
        # ESR_PREPACK=0: every network packs its weights at the start of its next training forward (round 3)
        self.prepack = os.environ.get('ESR_PREPACK', '1') != '0'
        # ESR_TRAIN_MANUAL=0: the step is written with autograd (losses as autograd Functions, torch.autograd.backward
        # over the three networks' nodes: ~100 glue launches — clones, gradient copies / sums, zero fills — and their
        # host time per step).  Default: the same launch lists driven directly (`_step_manual`), when the networks
        # allow it (`_manual_ok`)
        self.manual = os.environ.get('ESR_TRAIN_MANUAL', '1') != '0'
        # netF(fake): forward, feature loss and input-gradient pass on the SIDE stream, next to netD's forward and its
        # G-step pass on the main stream (both hang off fake_H only; their two contributions to dL/d fake_H meet in one add)
        self.netf_side = os.environ.get('ESR_TRAIN_NETF_SIDE', '1') == '1'
        self.d_when = self._knob('ESR_TRAIN_DSTEP', 'last', ('first', 'mid', 'last'))       # see _step_manual
        # The logging form (sync_log=True: the host reads the losses every step and the call returns with the D-side tail
        # ordered on the caller's stream) wants the D step EARLY and the follower pass of G's weight gradients BIG: the main
        # stream waits for the side stream's tail at the end of the call, so what the D step gains by running late under the
        # backward chain is lost again.  Round 6, same box, ms per logging-form step (tools/train_marks.py sync, two runs):
        # last / 80 workgroups 7.06 / 7.02; first / 96: 6.90 / 7.02; first / 112: 6.65 / 6.68; mid / 112: 6.66 / 6.63; first /
        # 128: 6.59 / 6.66; mid / 128: 6.60 / 6.58.  An explicit ESR_TRAIN_DSTEP / ESR_BWD_FOLLOW_WGS rules both forms.
        self.d_when_sync = self._knob('ESR_TRAIN_DSTEP_SYNC', os.environ.get('ESR_TRAIN_DSTEP', 'mid'), ('first', 'mid', 'last'))
        fw = os.environ.get('ESR_TRAIN_SYNC_FOLLOW_WGS', '' if os.environ.get('ESR_BWD_FOLLOW_WGS') else '128')
        self.follow_wgs_sync = int(fw) if fw else None
        # host enqueue order of netD's forward vs netF(fake)
        self.order = self._knob('ESR_TRAIN_ORDER', 'main_first', ('main_first', 'side_first'))
        self.tail_side = os.environ.get('ESR_TRAIN_TAIL_SIDE', '1') == '1'
        self.prep_side = os.environ.get('ESR_TRAIN_PREP_SIDE', '1') == '1'      # A/B knob (round 5)
        # netD's forward in two stages: the `real` half on the side stream under the generator's forward (its input is
        # known when the step starts), the `fake` half alone behind the generator (A/B knob, round 5)
        self.d_split = os.environ.get('ESR_TRAIN_DSPLIT', '1') == '1'

        self.overlap_d_step = self.overlap >= 1

    @staticmethod
    def _knob(name, default, allowed):
        v = os.environ.get(name, default)
        if v not in allowed:          # (a misspelt ESR_TRAIN_DSTEP used to mean: the D step never runs)
            raise ValueError('%s=%r: expected one of %s' % (name, v, ', '.join(allowed)))
        return v

    @classmethod
    def _knob_int(cls, name, default, allowed):
        return int(cls._knob(name, default, tuple(str(a) for a in allowed)))

    def _side(self, dev, which=0):
        # ESR_STREAM_PROBE=1: streams PROBED to run concurrently with the caller's stream and with each other
        # (engine.concurrent_streams: the first stream a process creates shares the default stream's hardware queue);
        # default: plain new streams, which measured faster (see `overlap` above)
        return E.concurrent_streams(dev, 2, probe=os.environ.get('ESR_STREAM_PROBE', '0') == '1')[which]

    def _scale_t(self, dev):
        t = self.__dict__.get('_scale_tensor')
        if t is None or t.device != dev or float(self._scale_value) != float(self.loss_scale):
            t = self._scale_tensor = torch.full((), float(self.loss_scale), dtype=torch.float32, device=dev)
            self._scale_value = float(self.loss_scale)
        return t

    # ---- exchange accounting (bench.py dp_train) ----
    def comm_reset(self):
        self._steps = 0
        self.exG.reset_counters()
        self.exD.reset_counters()

    def comm_report(self):
        """Per step: all-reduce calls / bytes of the two gradient exchanges and the milliseconds the compute streams
        spent blocked on them (HIP events around the waits).  Synchronises the device."""
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        n = max(self._steps, 1)
        return {'calls_per_step': (self.exG.calls + self.exD.calls) / n,
                'bytes_per_step': (self.exG.bytes + self.exD.bytes) / n,
                'exposed_ms_per_step': (self.exG.exposed_ms() + self.exD.exposed_ms()) / n,
                'exposed_ms_per_step_G': self.exG.exposed_ms() / n, 'exposed_ms_per_step_D': self.exD.exposed_ms() / n}

    def _manual_ok(self):
        netG, netD, netF = self.netG, self.netD, self.netF
        return (self.manual and self.shared_d and getattr(netD, '_shared_ok', False) and netD.training
                and getattr(netG, 'flat_param_grads', False) and hasattr(netG, '_convs') and hasattr(netF, '_run_forward')
                and all(p.requires_grad for p in netG._convs()[1]) and all(p.requires_grad for p in netD.parameters()))

    def _step_manual(self, var_L, var_H, var_ref, z, sync_log):
        """optimize_parameters (SRRaGAN_model.py:113-168) as a hand-written forward / backward over the networks'
        launch lists — no autograd graph.  The same kernels in the same order as the autograd form of ``step`` (which
        stays for networks this path does not cover, and as the A/B reference: tests run both against the reference's
        golden steps); what disappears is the glue between them:
          * every loss kernel writes its gradient straight into the buffer the next backward list reads
            (losses.l1_raw / ragan_raw: the loss scale rides in the kernel);
          * dL/d fake_H = d l_pix + d l_fea + d l_gan is never summed by a launch: the pixel loss writes the buffer, the
            last layout ops of netF's and netD's input-gradient passes ADD into it (esr_layout.accumulate);
          * netD's parameter gradients stay in its plan's flat buffer (the parameters' .grad are persistent views of
            it: FusedAdam reads them in place); RRDBNet's leave the backward as one flat buffer which the parameters'
            `.grad` alias (`_deliver_flat_grads(adopt=True)`: no copy) and FusedAdam reads in place."""
        from . import functional as Fn
        from . import convnet as CN
        netG, netD, netF = self.netG, self.netD, self.netF
        dev = var_L.device
        S = float(self.loss_scale)
        sdev = self.scaler.state if self.scaler else None          # state[0] = the dynamic loss scale (device)
        mean = self.data_parallel
        ov = self.overlap
        n = var_L.shape[0]
        main = torch.cuda.current_stream()
        side = self._side(dev, 0) if ov >= 1 else None
        if not netG.mark_grads_stale():
            self.optimizer_G.zero_grad(set_to_none=True)
        marks = self.__dict__.get('_marks')          # measurement (tools/train_marks.py): timed events on the main stream

        def mark(name):
            if marks is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record(main)
                marks.append((name, e))
        mark('start')
        with torch.no_grad():
            if ov >= 1:
                ev0 = torch.cuda.Event()
                ev0.record(main)                              # var_H (and whatever the caller did before the step)
            # the generator's forward is enqueued FIRST: a loop that reads its losses every step has the host running
            # behind the GPU, and the step's critical path starts with this launch list, not with netF(real)'s
            fake, stG = Fn.rrdbnet_train_forward(netG, var_L, z)
            self.fake_H = fake
            mark('G forward')
            ev_prep = None
            dsplit = (self.d_split and ov >= 1 and netD._has_bn and netD.training and not E.use_graphs()
                      and not getattr(netD, '_per_call_weights', False))
            d_early = None
            if ov >= 1:
                side.wait_event(ev0)
                with torch.cuda.stream(side):
                    real_fea = netF._run_forward(var_H, need_bwd=False)[0]
                    if self.prep_side:
                        # the G backward's step-independent preliminaries (zero fills of the gradient buffers, the backward
                        # chain's weight-stream gather) under the generator's forward instead of in front of the backward
                        Fn.rrdbnet_train_prepare(netG, stG)
                        ev_prep = torch.cuda.Event()
                        ev_prep.record(side)
                    if dsplit:
                        # netD(real): the previous step's D-side tail (Adam, packs) sits on this stream, in front of it
                        var_ref.record_stream(side)
                        d_early = netD._pair_begin(var_ref) + (torch.cuda.Event(),)
                        d_early[2].record(side)
                real_fea.record_stream(main)
            gy = self.__dict__.get('_gy')
            if gy is None or gy.shape != fake.shape or gy.device != fake.device:
                gy = self._gy = torch.empty_like(fake)
            l_g_pix = LS.l1_raw(fake, var_H, self.l_pix_w, grad_out=gy, grad_scale=S, scale_dev=sdev)
            nf_side = self.netf_side and ov >= 1
            box = {}

            def netf_fake():
                fake_fea, leaseF = netF._run_forward(fake, need_bwd=True)
                PF = leaseF.plan
                box['leaseF'], box['PF'] = leaseF, PF
                box['l_g_fea'] = LS.l1_raw(fake_fea, real_fea, self.l_fea_w, grad_out=PF.gy_tensor, grad_scale=S, scale_dev=sdev)

            def netd_fwd():
                ev = self.__dict__.get('_ev_tail')
                if ev is not None:
                    main.wait_event(ev)                       # the previous step's D step, D's Adam and packs (side stream)
                # ONE netD forward for the step's four calls (forward_shared): groups (fake, real)
                if d_early is not None:
                    PD, leaseD = d_early[0], d_early[1]
                    main.wait_event(d_early[2])               # the real half (side stream)
                    out = netD._pair_finish(PD, fake)
                else:
                    out, leaseD = netD._run_forward(torch.cat([fake, var_ref]), need_bwd=True, groups=2 if netD._has_bn else 1, dual=n)
                    PD = leaseD.plan
                if side is not None:
                    out.record_stream(side)                   # (the D step reads it there)
                box['leaseD'], box['PD'], box['pg'], box['pr'] = leaseD, PD, out[:n], out[n:]
                # G step: BCE(pred_d_real - mean(pred_g_fake), 0) + BCE(pred_g_fake - mean(pred_d_real), 1), gradient to the fake half
                box['l_g_gan'], _ = LS.ragan_raw(out[n:], out[:n], False, True, self.l_gan_w, grad_x=None, grad_y=PD.second.gy_tensor,
                                                 grad_scale=S, scale_dev=sdev, global_mean=mean)

            if nf_side:
                # netF(fake) — forward, feature loss, input-gradient pass — on the side stream next to netD's forward and
                # G-step pass on the main stream.  The HOST enqueues the main stream's netD forward first (it is the longer
                # chain: the G backward hangs off it); the two input gradients meet in one add.
                gy2 = self.__dict__.get('_gy2')
                if gy2 is None or gy2.shape != fake.shape or gy2.device != fake.device:
                    gy2 = self._gy2 = torch.empty_like(fake)
                side.wait_stream(main)                        # fake_H
                fake.record_stream(side)
                if self.order == 'main_first':
                    netd_fwd()
                with torch.cuda.stream(side):
                    netf_fake()
                    CN.run_pass_into(box['PF'], gx_into=gy2, accumulate=False)
                    ev_f = torch.cuda.Event()
                    ev_f.record(side)
                if self.order != 'main_first':
                    netd_fwd()
            else:
                if ov >= 1:
                    main.wait_stream(side)
                else:
                    real_fea = netF._run_forward(var_H, need_bwd=False)[0]
                netf_fake()
                netd_fwd()
            leaseF, PF, l_g_fea = box['leaseF'], box['PF'], box['l_g_fea']
            leaseD, PD, pg, pr, l_g_gan = box['leaseD'], box['PD'], box['pg'], box['pr'], box['l_g_gan']
            mark('pixel loss, netD(fake) forward, GAN loss')
            ev_glog = None
            if sync_log and ov >= 1:
                ev_glog = torch.cuda.Event()          # the G step's three losses are enqueued (main; l_g_fea maybe on side)
                ev_glog.record(main)

            def d_step():
                # D step (SRRaGAN_model.py:143-168): the second pair of calls sees the first pair's values
                if d_early is not None and PD.restat1.ops:
                    PD.restat1.run(E.current_stream())            # the running-statistics update the early half still owes
                if PD.restat is not None and PD.restat.ops:
                    PD.restat.run(E.current_stream())
                gyt = PD.gy_tensor                                # plan order [fake; real]
                l_d, aux = LS.ragan_raw(pr, pg, True, False, 1.0, grad_x=gyt[n:], grad_y=gyt[:n],
                                        grad_scale=S, scale_dev=sdev, global_mean=mean)
                CN.run_pass_into(PD)
                views = PD.__dict__.get('_param_views')
                if views is None:
                    views, off = [], 0
                    for numel, shape in PD.grad_views:
                        views.append(PD.grad_flat[off:off + numel].view(shape))
                        off += numel
                    PD._param_views = views
                    PD._param_list = [t for _, t in netD._pspec()]
                for p_, v in zip(PD._param_list, views):
                    p_.grad = v
                self.exD.start()
                return aux

            # When the host enqueues the D step (~60 launches, ~0.5-1 ms of host time during which the main stream gets
            # nothing new).  Round 4, same box, ms per step with netF(fake) on the main / side stream: 'first' (before
            # the main stream's backward passes) 8.03 / 7.92, 'mid' 8.05 / 7.92, 'last' (behind the G backward's launch:
            # it then runs under the backward chain) 8.35 / 7.57.
            # Where the HOST enqueues it matters as much: the D step is ~60 launches (~1 ms of host time) during which the
            # main stream gets nothing new.  'mid': the main stream's netD / netF input-gradient passes (two C calls,
            # ~0.9 ms of GPU work) go out first and run while the host enqueues the D step; the G backward follows.
            d_when = self.d_when_sync if sync_log else self.d_when
            if ov >= 1:
                side.wait_stream(main)
                if d_when == 'first':
                    with torch.cuda.stream(side):
                        aux = d_step()
            # dL/d fake_H: + d l_gan (netD, first pair) + d l_fea (netF), added by the passes' last layout ops
            CN.run_pass_into(PD.second, gx_into=gy, accumulate=True)
            mark('netD input-gradient pass (G step)')
            if nf_side:
                main.wait_event(ev_f)
                gy.add_(gy2)
                mark('wait for netF(fake) pass, add')
            else:
                CN.run_pass_into(PF, gx_into=gy, accumulate=True)
            if ov >= 1 and d_when == 'mid':
                with torch.cuda.stream(side):
                    aux = d_step()
            if ev_prep is not None:
                main.wait_event(ev_prep)
            Fn.rrdbnet_train_backward(netG, stG, gy, follow_wgs=self.follow_wgs_sync if sync_log else None)
            mark('G backward (tail, chain, weight gradients, unpermute)')
            self.exG.start()
            if ov >= 1:
                if d_when == 'last':
                    with torch.cuda.stream(side):
                        aux = d_step()
            else:
                aux = d_step()
            leaseF.release()
            leaseD.release()
            inv = 1.0 / self.loss_scale
            # The step's tail.  tail_side (static loss scale): the main stream only carries what the NEXT generator forward
            # waits for — G's Adam and the pack of its forward weights; the D step's end, D's Adam, D's packs and G's
            # input-gradient packs stay on the side stream, and the main stream meets them again (ev_tail) in front of the
            # next netD forward.
            # Only in the pipelined form of the call (sync_log=False: the caller reads nothing before `finish()` / a device
            # synchronisation); the default call returns with everything ordered on the current stream.
            # sync_log (the reference reads seven .item()s per step, SRRaGAN_model.py:171-186): round 5 — the host no longer
            # waits for the END of the step.  The seven scalars are gathered and copied to pinned memory on the side stream
            # right behind the D step's loss kernel (ev_log); the tail below is enqueued first, THEN the host waits for
            # ev_log only: it returns ~1.3 ms of GPU work early and enqueues the next step under the rest of this one
            # (8.1 -> ~7 ms per step for a loop that logs every step).
            ev_log, log_keep = None, None
            if ev_glog is not None:
                with torch.cuda.stream(side):
                    side.wait_event(ev_glog)
                    host = self.__dict__.get('_log_host')
                    if host is None:
                        host = self._log_host = torch.empty(7, dtype=torch.float32).pin_memory()
                    log_keep = torch.stack([t.detach().reshape(()).float() for t in
                                            (l_g_pix, l_g_fea, l_g_gan, aux[2], aux[3], aux[0], aux[1])])
                    host.copy_(log_keep, non_blocking=True)
                    ev_log = torch.cuda.Event()
                    ev_log.record(side)
            tail_side = ov >= 1 and self.tail_side and self.scaler is None
            if not tail_side:
                if ov >= 1:
                    main.wait_stream(side)
                self.exG.wait()
                self.optimizer_G.step(grad_scale=inv, scaler=self.scaler)
                self.exD.wait()
                self.optimizer_D.step(grad_scale=inv, scaler=self.scaler)
                if self.scaler:
                    self.scaler.update()
                if self.prepack:
                    netG.prepack(fwd=True, dgrad=False)
                    if ov >= 1:
                        side.wait_stream(main)
                        with torch.cuda.stream(side):
                            netD.prepack()
                            netG.prepack(fwd=False, dgrad=True)
                    else:
                        netD.prepack()
                        netG.prepack(fwd=False, dgrad=True)
                if ov >= 1:
                    self._ev_tail = torch.cuda.Event()
                    self._ev_tail.record(side)
                    self._defer_networks(self._ev_tail)
            else:
                self.exG.wait()
                self.optimizer_G.step(grad_scale=inv, scaler=None)
                if self.prepack:
                    netG.prepack(fwd=True, dgrad=False)
                mark('Adam(G), forward weight pack')
                side.wait_stream(main)                        # G's new weights (its input-gradient packs read them)
                with torch.cuda.stream(side):
                    self.exD.wait()
                    self.optimizer_D.step(grad_scale=inv, scaler=None)
                    if self.prepack:
                        netD.prepack()
                        netG.prepack(fwd=False, dgrad=True)
                    self._ev_tail = torch.cuda.Event()
                    self._ev_tail.record(side)
                self._defer_networks(self._ev_tail)
        logs = dict(l_g_pix=l_g_pix, l_g_fea=l_g_fea, l_g_gan=l_g_gan, l_d_real=aux[2], l_d_fake=aux[3],
                    D_real=aux[0], D_fake=aux[1])
        if sync_log:
            if ev_log is not None:
                ev_log.synchronize()                              # the losses, not the end of the step
                self.log = dict(zip(('l_g_pix', 'l_g_fea', 'l_g_gan', 'l_d_real', 'l_d_fake', 'D_real', 'D_fake'),
                                    self._log_host.tolist()))
                del log_keep
                # the default call's contract: everything it enqueued is ordered on the CURRENT stream when it returns
                main.wait_event(self._ev_tail)
            else:
                if ov >= 1:
                    main.wait_stream(side)
                self.log = {k: float(v) for k, v in logs.items()}
        else:
            self.log = logs
        return self.log

    def _defer_networks(self, ev):
        # the networks' PUBLIC entry points (forward / forward_pair / state_dict) order this event in front of their
        # caller's stream (block._PlannedModule._join_pending): a validation forward or a checkpoint between two steps
        # sees the finished optimizer updates and weight packs without the loop having to call finish()
        for net in (self.netG, self.netD):
            if hasattr(net, '_defer_to'):
                net._defer_to(ev)

    def state_dict(self):
        """The two optimizers' states (base_model.py:65-74 `save_training_state`), ordered behind whatever a pipelined
        step left in flight."""
        self.finish()
        return {'optimizers': [self.optimizer_G.state_dict(), self.optimizer_D.state_dict()]}

    def finish(self):
        """Orders what a pipelined step (``step(..., sync_log=False)``) left on the side stream — the end of the D
        step, D's Adam, the weight packs — in front of the current stream: call it (or synchronise the device) before
        reading the networks' parameters, buffers or the logged losses after such a step."""
        ev = self.__dict__.get('_ev_tail')
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def step(self, var_L, var_H, var_ref=None, z=None, sync_log=True):
        """One optimisation step (SRRaGAN_model.py:113-168).  sync_log=False: the pipelined form for training loops —
        the logged losses stay device tensors and the step's discriminator-side tail may still be in flight on the side
        stream when the call returns (the next step, ``finish()`` and a device synchronisation order it)."""
        netG, netD, netF = self.netG, self.netD, self.netF
        var_ref = var_H if var_ref is None else var_ref
        self._steps += 1
        E.require_cuda(var_L, 'ESRGANPlusStep.step: var_L')      # (the networks and the fused losses have no CPU path)
        if self._manual_ok():
            return self._step_manual(var_L, var_H, var_ref, z, sync_log)
        # batch means of the relativistic terms: over ALL ranks when data-parallel (losses._RaGANGlobalFn: the fused
        # kernel + two scalar all-reduces), else inside the one fused loss launch
        mean = self.data_parallel
        cuda = True
        ov = self.overlap
        # ---------------- G ----------------
        for p in netD.parameters():
            p.requires_grad = False
        if not (hasattr(netG, 'mark_grads_stale') and netG.mark_grads_stale()):
            self.optimizer_G.zero_grad(set_to_none=True)     # (first step / gradients that are not the module's store)
        if ov >= 1:
            # netF(var_H) does not depend on G: on the second stream, under the generator's forward
            main = torch.cuda.current_stream()
            side = self._side(var_L.device, 0)
            side.wait_stream(main)
            with torch.cuda.stream(side), torch.no_grad():
                real_fea = netF(var_H)
            real_fea.record_stream(main)
        fake_H = netG(var_L, z=z) if z is not None else netG(var_L)
        self.fake_H = fake_H
        l_g_pix = LS.l1_loss(fake_H, var_H, self.l_pix_w)

        shared = self.shared_d and getattr(netD, '_shared_ok', False) and netD.training
        dh = []

        def d_pass():
            # both operands in ONE pass (forward_pair: per-half BatchNorm statistics, the detached ``real`` half
            # costs no backward) — the reference's call order fake, real is the group order.  shared: the same pass
            # also keeps what the D step's pair needs (forward_shared), so that pair costs no second forward
            if shared:
                pg, pr, h = netD.forward_shared(fake_H, var_ref)
                dh.append(h)
            else:
                pg, pr = netD.forward_pair(fake_H, var_ref)
            return LS.ragan_loss(pr, pg, False, True, self.l_gan_w, mean)[0]

        if ov >= 2:
            sideD = self._side(var_L.device, 1)
            sideD.wait_stream(main)
            sideD.wait_stream(side)               # (netD's weight packs of the previous step's end: prepack)
            with torch.cuda.stream(sideD):
                l_g_gan = d_pass()
        if ov >= 1:
            # join BEFORE netF runs on the main stream: the first netF call of a process packs its weights on the
            # side stream, and the side work finished under the generator's forward anyway
            main.wait_stream(side)
            fake_fea = netF(fake_H)
        else:
            fake_fea, real_fea = netF.forward_pair(fake_H, var_H)
        l_g_fea = LS.l1_loss(fake_fea, real_fea, self.l_fea_w)
        if ov < 2:
            l_g_gan = d_pass()
        scale = self.scaler.scale if self.scaler else self._scale_t(fake_H.device)

        def d_step():
            # ---------------- D ---------------- (SRRaGAN_model.py:143-168; only needs fake_H's VALUES and D as it is)
            for p in netD.parameters():
                p.requires_grad = True
            self.optimizer_D.zero_grad(set_to_none=True)
            if shared:
                pred_d_real, pred_d_fake = dh.pop().second_pass()
            else:
                with netD.weights_unchanged():        # no optimizer step since the G step's D pass
                    pred_d_real, pred_d_fake = netD.forward_pair(var_ref, fake_H.detach())
            l_d_total, aux = LS.ragan_loss(pred_d_real, pred_d_fake, True, False, 1.0, mean)
            torch.autograd.backward([l_d_total], [scale])
            return aux

        def g_backward():
            # d(scale * (pix + fea + gan)): one backward over the three terms, no sum / multiply launches
            torch.autograd.backward([l_g_pix, l_g_fea, l_g_gan], [scale, scale, scale])

        if ov >= 1:
            # The D step does not depend on the G backward: it runs on a second stream UNDER it (both are chains of
            # small launches).  Same arithmetic, same order of BatchNorm running-statistics updates (its forward
            # still follows the G step's D pass); the autograd graphs are disjoint.  Data-parallel runs take the
            # same route: every rank issues its collectives in the same program order (D's loss sums, D's gradient
            # buckets on the side stream; G's in-backward buckets on the main stream), each stream-ordered after
            # the kernels that feed it.
            if ov >= 2:
                # fake_H was complete when sideD started; the D step must follow the G step's D pass (BatchNorm
                # running statistics, the weight pack) but not netF(fake_H) on the main stream
                side.wait_stream(sideD)
            else:
                side.wait_stream(main)
            with torch.cuda.stream(side):
                aux = d_step()
                self.exD.start()
            g_backward()
            self.exG.start()
            main.wait_stream(side)
            if ov >= 2:
                main.wait_stream(sideD)
        else:
            g_backward()
            self.exG.start()                  # RCCL all-reduce of G grads overlaps the D pass below
            aux = d_step()
            self.exD.start()
        inv = 1.0 / self.loss_scale          # the loss-scale division rides inside the Adam kernel
        self.exG.wait()
        self.optimizer_G.step(grad_scale=inv, scaler=self.scaler)
        self.exD.wait()
        self.optimizer_D.step(grad_scale=inv, scaler=self.scaler)
        if self.scaler:
            self.scaler.update()
        if self.prepack and cuda:
            # the next step's weight packs, now: the generator's forward copy on the main stream (the step starts with
            # it anyway), everything that is only needed later — D's two packs, G's input-gradient operands — on the
            # side stream, next to it and to the start of the next generator forward
            netG.prepack(fwd=True, dgrad=False)
            if ov >= 1:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    netD.prepack()
                    netG.prepack(fwd=False, dgrad=True)
                # (joined at the next step's `main.wait_stream(side)` in front of netF(fake_H): before any consumer)
            else:
                netD.prepack()
                netG.prepack(fwd=False, dgrad=True)
        logs = dict(l_g_pix=l_g_pix, l_g_fea=l_g_fea, l_g_gan=l_g_gan, l_d_real=aux[2], l_d_fake=aux[3],
                    D_real=aux[0], D_fake=aux[1])
        if sync_log:      # the reference calls .item() on every loss each step (SRRaGAN_model.py:171-186)
            self.log = {k: float(v.detach()) for k, v in logs.items()}
        else:
            self.log = {k: v.detach() for k, v in logs.items()}
        return self.log
