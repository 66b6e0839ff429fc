#!/usr/bin/env python
"""bench.py — HR megapixels/s of the RRDBNet x4 hot path on MI355X.

Workload (BASELINE.json configs[1], SURVEY.md §8d "Config 2"): RRDBNet x4 (23 RRDB, nf=64, gc=32)
fp16 storage / fp32 accumulate, forward only, batch 16 of 128x128 LR tiles -> 512x512 HR, per GPU.
One "step" = one such forward over one synthetic batch already resident in HBM.
N>1 (launched by torch.distributed.run): tiles are independent, so every rank runs the same step
on its own batch (weak scaling, no data-path collective); value = all ranks' HR-Mpix / max time.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, measured with HIP events on the
launch stream) and `cpu_baseline` (the oracle restatement timed on the host cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

NB, BATCH, LR = 23, 16, 128
MAC_PER_LR_PIXEL = 18068160            # SURVEY.md §8d [probe]: conv MACs of RRDBNet x4 forward
PEAK_F16_TFLOPS = 2500.0               # MI355X dense fp16 MFMA (MI355X_MICROARCH.md)
# conv MACs per pixel of one ResidualDenseBlock_5C: conv1..conv5 3x3 + the 1x1 (block.py:239-258)
RDB_MAC_PER_PIXEL = 9 * 32 * (64 + 96 + 128 + 160) + 9 * 64 * 192 + 64 * 32


def committed_traffic(kernel):
    """(HBM bytes per launch, source note) of `kernel` from profiles/roofline_traffic.json — or (None, why) when the
    row is missing or was measured on another version of the kernel's sources (its `source_sha` stamp differs)."""
    try:
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import traffic_hashes as TH
        tj = json.load(open(os.path.join(ROOT, 'profiles', 'roofline_traffic.json')))
        if kernel not in tj:
            return None, None
        if not TH.fresh(tj, kernel):
            return None, 'profiles/roofline_traffic.json row is STALE (kernel sources changed since the PMC pass)'
        return tj[kernel]['read_bytes'] + tj[kernel]['write_bytes'], \
            'profiles/roofline_traffic.json (%s)' % tj.get('_source', 'rocprofv3 --pmc')
    except (OSError, ValueError, KeyError, ImportError):
        return None, None


def committed_profile(kernel):
    """The committed rocprofv3 --kernel-trace --stats figure of `kernel` (average launch duration on the PROFILED box,
    tools/update_traffic.py), or None: printed next to this run's own HIP-event figure so that the line itself says the
    two fractions come from different boxes."""
    try:
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import traffic_hashes as TH
        tj = json.load(open(os.path.join(ROOT, 'profiles', 'roofline_traffic.json')))
        row = tj.get(kernel, {})
        if 'profiled_avg_launch_us' not in row or not TH.fresh(tj, kernel):
            return None
        return row
    except (OSError, ValueError, KeyError, ImportError):
        return None


def conv_flops(c):
    """Algorithmic FLOPs of one fused-conv launch: 2 * MACs of the convolution(s) it replaces
    (true Cin/Cout, not the padded GEMM the kernel runs)."""
    return 2.0 * c['B'] * c['H'] * c['W'] * (c['cout'] * c['cin'] * c['ks'] ** 2 + c['extra_mac'])


def describe_plan(net, plan):
    """(kernel-variant name, algorithmic flops) for every op of the recorded plan."""
    from esrganplus_amd import _lib as L
    ent = {e.w_ptr: e for e in net._wp[(net.precision, str(next(net.parameters()).device))].entries.values()}
    out = []
    for o in plan.ops.ops:
        if o.kind == L.OP_RDB_CHAIN:
            # the fused dense-block chain (rdb_fused.hip): n_blocks x ResidualDenseBlock_5C, block.py:260-268
            ch = o.u.rdb_chain
            out.append(('rdb_chain', 2.0 * RDB_MAC_PER_PIXEL * ch.n_blocks * ch.B * ch.H * ch.W))
            continue
        if o.kind != L.OP_CONV:
            out.append(('layout', 0.0))
            continue
        c = o.u.conv
        e = ent[c.w]
        extra = 0
        if c.w1x1:
            extra = 32 * 64          # fused bias-free 1x1 64->32 (block.py:263)
        cbk = c.cout_blocks
        # kernel variants of conv_mfma.hip: <ks, cout per wave (32*NCW), extras>
        name = 'conv%dx%d_c%d' % (c.ks, c.ks, 32 if cbk == 1 else 64)
        if c.upsample:
            name += '_ups'
        if c.w1x1:
            name += '_1x1'
        fl = conv_flops(dict(B=c.B, H=c.H, W=c.W, cout=e.cout, cin=e.cin, ks=e.ks, extra_mac=extra))
        if c.upsample == 3:
            # nearest-x2 + 3x3 in its 4-phase 2x2 form: the kernel line counts the MACs it executes (16 of the
            # reference op's 36 per 4 outputs); the whole-net figure keeps the reference's count
            name, fl = 'upconv_subpix_c%d' % (32 if cbk == 1 else 64), fl * 4.0 / 9.0
        out.append((name, fl))
    return out


def host_cores():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota (a box that
    reports 256 CPUs but grants a few cores' worth of quota must not get 256 oneDNN threads)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, min(n, 64))


def mfma_sustained_probe(dev, seconds=1.6):
    """What the matrix pipe of THIS box sustains under the package power limit: every SIMD issues independent
    v_mfma_f32_32x32x16_f16 back to back from registers (esr_debug_mfma_probe), operands = random fp16 (weights-like A,
    activation-like B: the toggle rate of real data).  The quoted 2.5 PFLOP/s is reached with ZERO operands at 2.4 GHz;
    with real data the power management settles near 1.65-1.7 GHz (profiles/r05_mfma_power_probe.txt).  Context for
    `roofline.frac`, not a replacement of its peak."""
    import ctypes as C
    from esrganplus_amd import _lib as L, engine as E
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    g = torch.Generator(device='cpu').manual_seed(11)
    ops = torch.cat([torch.randn(4 * 256 * 8, generator=g) * 0.02, torch.randn(4 * 256 * 8, generator=g)]).half().to(dev)
    clk = torch.zeros(2, dtype=torch.int64, device=dev)
    sink = torch.zeros(1, dtype=torch.float32, device=dev)
    st = E.current_stream()
    iters = 60000                                    # ~20 ms per launch

    def launch():
        L.check(L.lib().esr_debug_mfma_probe(C.c_void_p(ops.data_ptr()), C.c_int32(iters), C.c_void_p(clk.data_ptr()),
                                             C.c_void_p(sink.data_ptr()), C.c_int32(cus), C.c_void_p(st)), 'esr_debug_mfma_probe')
    launch()
    torch.cuda.synchronize()
    t_end = time.perf_counter() + seconds            # load until the clock has settled, keep the last launches
    ms = []
    while time.perf_counter() < t_end:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1))
    last = sorted(ms[-5:])[len(ms[-5:]) // 2]
    c = clk.tolist()
    tf = cus * 4 * iters * 16 * 32768.0 / (last * 1e-3) / 1e12
    return {'tflops': round(tf, 1), 'shader_clock_ghz': round(c[0] / max(c[1], 1) * 0.1, 3), 'seconds_of_load': seconds,
            'what': 'independent v_mfma_f32_32x32x16_f16 from registers on every SIMD, random fp16 operands, after %.1f s of '
                    'load (esr_debug_mfma_probe): the sustained dense fp16 rate of this box under its power limit' % seconds}


def cpu_baseline(seconds_budget=25.0):
    """Oracle (torch restatement == reference bit-for-bit, oracle/gen_golden.py) on the host cores:
    RRDBNet x4 fp32 eval forward.  Calibrates on a 32x32 LR tile, then times the largest LR tile
    (128, 64 or 32 square — same MACs per pixel) whose forwards fit the time budget."""
    from esrganplus_amd import synth
    from oracle import ref_torch as RT
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = synth.rrdbnet_state_dict(NB, 0)

    def timed(n):
        x = synth.image_batch(0, 1, 3, n, n, name='bench.cpu')
        t0 = time.perf_counter()
        with torch.no_grad():
            RT.rrdbnet_forward(x, sd, NB)
        return time.perf_counter() - t0

    timed(32)                                   # warm-up (oneDNN primitive creation)
    t32 = timed(32)
    size = 128 if t32 * 16 * 3 <= seconds_budget else (64 if t32 * 4 * 3 <= seconds_budget else 32)
    est = t32 * (size // 32) ** 2
    n = max(1, min(5, int(seconds_budget / max(est, 1e-3)) - 1))
    timed(size)
    ts = sorted(timed(size) for _ in range(n))
    med = ts[len(ts) // 2]
    # the single-thread figure, on a 32x32 tile so it stays within a few seconds
    torch.set_num_threads(1)
    timed(32)
    t1 = timed(32)
    torch.set_num_threads(cores)
    model = ''
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                model = line.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    return {'value': round((4 * size) ** 2 / 1e6 / med, 4), 'unit': 'HR-Mpix/s', 'cores': cores,
            'kind': 'port',
            'sample': '%d forwards of one 1x3x%dx%d LR tile, fp32 eval, median %.3f s' % (n, size, size, med),
            'gflops': round(2 * MAC_PER_LR_PIXEL * size * size / med / 1e9, 1),
            'one_thread': {'value': round(128 ** 2 / 1e6 / t1, 4), 'unit': 'HR-Mpix/s',
                           'sample': 'one forward of a 1x3x32x32 LR tile, %.3f s' % t1},
            'cpu_model': model}


TRAIN_STEP_MAC = 1.515e12               # conv/linear MACs of one batch-16 ESRGAN+ step (G + D + VGG, fwd + bwd)


def _max_over_ranks(elapsed, dist, dev):
    if dist is None:
        return elapsed
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == 'nccl' else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _sync_all(dist):
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()


def make_train_step(args, rank, dev, data_parallel):
    """The three networks + optimizers of BASELINE configs[2]/[3] (train_ESRGANplus.json) on synthetic weights."""
    from esrganplus_amd import architecture as arch, synth, train, dp as DP
    prec = args.precision
    netG = arch.RRDBNet(3, 3, 64, NB).to(dev).train().set_precision(prec)
    netD = arch.Discriminator_VGG_128(3, 64).to(dev).train().set_precision(prec)
    netF = arch.VGGFeatureExtractor(34, False, True, dev).to(dev).eval().set_precision(prec)
    netG.load_state_dict(synth.rrdbnet_state_dict(NB, 0, gain=0.5))
    netD.load_state_dict(synth.discriminator_state_dict(0))
    netF.load_state_dict(synth.vgg19_state_dict(0, 34), strict=False)
    if data_parallel:
        for n in (netG, netD):
            DP.broadcast_parameters(n)
    st = train.ESRGANPlusStep(netG, netD, netF, loss_scale=1024.0 if prec == 'fp16' else 1.0,
                              data_parallel=data_parallel)
    tb = args.train_batch
    lr = synth.image_batch(200 + rank, tb, 3, 32, 32, name='bench.lr').to(dev)
    hr = synth.image_batch(300 + rank, tb, 3, 128, 128, name='bench.hr').to(dev)
    return st, lr, hr


def measure_train(args, world, rank, dev, dist, steps, warmup, data_parallel=None):
    """ms per full ESRGAN+ step (max over ranks) + the step object (kept for the per-kernel pass)."""
    dp_on = (world > 1) if data_parallel is None else data_parallel
    st, lr, hr = make_train_step(args, rank, dev, dp_on)
    for _ in range(max(warmup, 1)):
        st.step(lr, hr, sync_log=False)
    _sync_all(dist if dp_on or dist is not None else None)
    if dp_on:
        st.comm_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        log = st.step(lr, hr, sync_log=False)
    torch.cuda.synchronize()
    elapsed = _max_over_ranks(time.perf_counter() - t0, dist, dev)
    assert all(torch.isfinite(v).all() for v in log.values())
    return elapsed / steps, st, (lr, hr)


def train_object(args, world, rank, dev, dist, steps, warmup, kernels=True):
    """BASELINE configs[2]: the full ESRGAN+ train step on ONE GPU's batch (no exchange), as an object of the
    default bench line: ms/step, TFLOP/s against SURVEY §8a's 1.515 TMAC per batch-16 step, and the generator's
    fused kernels (the dominant launches of the step) timed with HIP events."""
    dt, st, (lr, hr) = measure_train(args, 1, rank, dev, None, steps, warmup, data_parallel=False)
    tb = args.train_batch
    fl = 2.0 * TRAIN_STEP_MAC * tb / 16
    res = {'ms_per_step': round(dt * 1e3, 3), 'value': round(tb * 128 * 128 / 1e6 / dt, 3), 'unit': 'HR-Mpix/s',
           'tflops': round(fl / dt / 1e12, 1), 'frac_of_f16_mfma_peak': round(fl / dt / 1e12 / PEAK_F16_TFLOPS, 4),
           'steps': steps, 'warmup': warmup,
           'what': 'full ESRGAN+ train step (RRDBNet nb=23 noise on + Discriminator_VGG_128 + VGG19[:35] feature loss, '
                   'Adam x2, loss scale 1024), batch %d of 32x32 LR -> 128x128 HR, %s (BASELINE configs[2]); pipelined calls '
                   '(step(sync_log=False): nothing read back between steps), K steps to a device synchronise'
                   % (tb, args.precision)}
    # the reference's loop reads seven .item()s per step (SRRaGAN_model.py:171-186): the same step with the default,
    # synchronous call (everything ordered on the current stream, the losses read back as floats every step)
    st.finish()
    torch.cuda.synchronize()
    for _ in range(3):
        st.step(lr, hr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        st.step(lr, hr)
    torch.cuda.synchronize()
    res['ms_per_step_sync_log'] = round((time.perf_counter() - t0) / steps * 1e3, 3)
    if kernels and args.precision == 'fp16':
        k = generator_kernel_times(st.netG, lr, tb, 32)
        if k:
            res.update(k)
    return res, st


def generator_kernel_times(netG, lr, batch, size, reps=5):
    """HIP-event times of the generator's recorded training launch lists (same buffers as a real step): the fused
    dense-block kernels by name, the slowest of them as the `roofline` object."""
    from esrganplus_amd import engine as E, functional as Fn, _lib as L
    tps = [t for k, pool in netG._plans.items() if isinstance(k, tuple) and k and k[0] == 'train' for t in pool]
    tps = [t for t in tps if t.fwd.out_shape[0] == batch and t.fwd.out_shape[2] == 4 * size and not t.graph]
    if not tps:
        return None
    tp = tps[0]
    dev = lr.device
    st = E.current_stream()
    out = torch.empty(tp.fwd.out_shape, dtype=torch.float32, device=dev)
    gy = torch.full(tp.fwd.out_shape, 1024.0 / out.numel(), dtype=torch.float32, device=dev)
    blk_fl = 2.0 * RDB_MAC_PER_PIXEL * batch * size * size          # per dense block, any of the three passes

    def name_of(o):
        if o.kind == L.OP_RDB_CHAIN:
            return 'rdb_chain_train', blk_fl * o.u.rdb_chain.n_blocks
        if o.kind == L.OP_RDB_CHAIN_BWD:
            return 'rdb_chain_bwd', blk_fl * o.u.rdb_chain.n_blocks
        if o.kind == L.OP_RDB_WGRAD:
            return 'rdb_wgrad', blk_fl * o.u.rdb_wgrad.n_blocks
        if o.kind == L.OP_WGRAD:
            w = o.u.wgrad
            return 'wgrad', 2.0 * w.B * w.H * w.W * w.cout * w.cin * w.ks * w.ks
        if o.kind == L.OP_CONV:
            return 'conv (head / tail / their dgrad)', 0.0
        return 'other (layout, pack, unpermute)', 0.0

    agg = {}
    for rep in range(reps + 1):
        tp.fwd.run(lr, out, st, 1234, None)                  # binds I/O + seed (untimed), then the timed replay
        ms_f = tp.fwd.ops.run_timed(st)
        Fn._train_backward(tp, gy, st, True, False, 1234, False)
        ms_b = tp.bwd.run_timed(st, tp.gx_begin)     # (the ops behind gx_begin give dL/dx: not part of a training step)
        if rep == 0:
            continue
        for ops, ms in ((tp.fwd.ops.ops, ms_f), (tp.bwd.ops, ms_b)):
            for o, t in zip(ops, ms):
                nm, f = name_of(o)
                a = agg.setdefault(nm, [0.0, 0.0, 0])
                a[0] += t
                a[1] += f
                a[2] += 1
    res = {'kernels': {k: {'launches': a[2] // reps, 'ms_per_step': round(a[0] / reps, 3),
                           'tflops': round(a[1] / (a[0] * 1e-3) / 1e12, 1) if a[1] else 0.0}
                       for k, a in sorted(agg.items())}}
    fused = [k for k in ('rdb_chain_train', 'rdb_chain_bwd', 'rdb_wgrad') if k in agg]
    if fused:
        dom = max(fused, key=lambda k: agg[k][0])
        t_ms, f, n = agg[dom]
        ach = f / (t_ms * 1e-3) / 1e12
        # committed PMC rows: the bench shape (profile_fwdbwd.sh) and the train step's shape (tools/pmc_train.sh)
        traffic = committed_traffic(dom)[0] if (batch == BATCH and size == LR) else (
            committed_traffic(dom + '@train')[0] if (batch == 16 and size == 32) else None)
        res['roofline'] = {'bound': 'mfma', 'kernel': dom, 'achieved': round(ach, 1), 'peak': PEAK_F16_TFLOPS,
                           'unit': 'TFLOP/s', 'frac': round(ach / PEAK_F16_TFLOPS, 4), 'traffic': traffic,
                           'launches_per_step': n // reps, 'avg_launch_us': round(t_ms / n * 1e3, 2),
                           'share_of_kernel_time': round(t_ms / reps / (sum(a[0] for a in agg.values()) / reps), 3)}
    return res


def dp_train_object(args, world, rank, dev, dist, steps=20, warmup=3):
    """BASELINE configs[3] at world > 1: the SAME train step data-parallel over the process group (RCCL; gloo only
    for one-GPU dry runs) next to this process's own no-exchange figure — bytes all-reduced per step, the time the
    compute streams spent blocked on the exchanges (HIP events around the waits), per-GPU step time with and
    without the exchange."""
    dt1, st1, _ = measure_train(args, 1, rank, dev, dist, steps, warmup, data_parallel=False)
    del st1
    torch.cuda.empty_cache()
    dtn, st, _ = measure_train(args, world, rank, dev, dist, steps, warmup, data_parallel=True)
    comm = st.comm_report()
    nG = sum(p.numel() for p in st.netG.parameters() if p.requires_grad)
    nD = sum(p.numel() for p in st.netD.parameters())
    tb = args.train_batch
    t = torch.tensor([comm['exposed_ms_per_step']], dtype=torch.float64,
                     device=dev if dist.get_backend() == 'nccl' else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return {'n_ranks': dist.get_world_size(), 'backend': dist.get_backend(),
            'ms_per_step': round(dtn * 1e3, 3), 'ms_per_step_no_exchange': round(dt1 * 1e3, 3),
            'value': round(world * tb * 128 * 128 / 1e6 / dtn, 3), 'unit': 'HR-Mpix/s', 'scaling': 'weak',
            'allreduce_bytes_per_step': 4 * (nG + nD) + 4 * 10,
            'allreduce_calls_per_step': comm['calls_per_step'],
            'exposed_comm_ms_per_step': round(float(t.item()), 3),
            'steps': steps, 'warmup': warmup,
            'what': 'ESRGAN+ train step, batch %d of 32x32 LR per rank, G gradients all-reduced in buckets inside the '
                    'backward (per RRDB), D gradients under the G backward, global-batch RaGAN means '
                    '(BASELINE configs[3])' % tb}


def dp_gtrain_object(args, world, rank, dev, dist, steps=5, warmup=2):
    """BASELINE configs[4] at world > 1: the mixed-tile generator step (16x128^2 + 8x192^2 + 4x256^2 LR per rank and step,
    noise on, L1, Adam) data-parallel over the process group — G's gradients all-reduced in buckets INSIDE each bucket's
    backward (per RRDB), three exchanges of 67.4 MB per step — next to this process's own no-exchange figure: per-bucket
    ms, bytes all-reduced, the time the compute stream spent blocked on the exchanges (HIP events around the waits)."""
    dt1, per1, lr_pix = measure_gtrain(args, world, rank, dev, dist, steps, warmup, data_parallel=False)
    torch.cuda.empty_cache()
    rep = {}
    dtn, pern, _ = measure_gtrain(args, world, rank, dev, dist, steps, warmup, data_parallel=True, report=rep)
    t = torch.tensor([rep['exposed_ms_per_step']], dtype=torch.float64, device=dev if dist.get_backend() == 'nccl' else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    f = gtrain_fields(dtn, pern, lr_pix, world)
    return {'n_ranks': dist.get_world_size(), 'backend': dist.get_backend(),
            'ms_per_step': f['ms_per_step'], 'ms_per_step_no_exchange': round(dt1 * 1e3, 3),
            'value': f['value'], 'unit': 'HR-Mpix/s', 'scaling': 'weak',
            'buckets_ms': {k: v['ms'] for k, v in f['buckets'].items()},
            'buckets_ms_no_exchange': {'%dx%d^2' % b: round(ms, 3) for b, ms in zip(GTRAIN_BUCKETS, per1)},
            'allreduce_bytes_per_step': int(rep['bytes_per_step']), 'allreduce_calls_per_step': rep['calls_per_step'],
            'expected_allreduce_bytes_per_step': 4 * rep['n_params'] * len(GTRAIN_BUCKETS),
            'exposed_comm_ms_per_step': round(float(t.item()), 3), 'steps': steps, 'warmup': warmup,
            'what': 'nESRGAN+ generator (noise on) fwd+bwd+Adam, L1 loss, 16x128^2 + 8x192^2 + 4x256^2 LR tiles per rank and '
                    'step, fp16, gradients all-reduced per bucket inside the backward (BASELINE configs[4])'}


def train_bench(args, world, rank, dev, dist):
    """BASELINE configs[2]/[3]: full ESRGAN+ train step (RRDBNet + Discriminator_VGG_128 + VGG19
    feature loss, Adam x2; train_ESRGANplus.json), per-GPU batch 16 of 32x32 LR -> 128x128 HR, data
    parallel over RCCL with the gradient exchange overlapped on the other network's pass."""
    dt, st, _ = measure_train(args, world, rank, dev, dist, args.steps, args.warmup)
    tb = args.train_batch
    if rank == 0:
        step_flops = 2.0 * TRAIN_STEP_MAC * tb / 16
        res = {'metric': 'HR megapixels/sec (x4 SR) full ESRGAN+ train step', 'unit': 'HR-Mpix/s',
               'value': round(world * tb * 128 * 128 / 1e6 / dt, 3), 'n_gpus': world,
               'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt * 1e3, 3),
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
               'dtype': 'f16' if args.precision == 'fp16' else 'f32', 'data': 'synthetic',
               'config': {'workload': 'ESRGAN+ train step (RRDBNet nb=23 + Discriminator_VGG_128 + VGG19[:35] '
                                      'feature loss, Adam x2), batch %d of 32x32 LR per GPU (BASELINE configs[2]/[3])' % tb,
                          'global_batch': world * tb, 'parallelism': 'dp%d, RCCL grad all-reduce overlapped' % world},
               'tflops_per_gpu': round(step_flops / dt / 1e12, 1)}
        if world > 1:
            res['comm'] = st.comm_report()
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


GTRAIN_BUCKETS = ((16, 128), (8, 192), (4, 256))


def measure_gtrain(args, world, rank, dev, dist, steps, warmup, data_parallel=None, report=None):
    """BASELINE configs[4]: the noise-injection generator in TRAIN mode (noise
    on), forward + backward + Adam with an L1 pixel loss, on mixed LR tiles bucketed by size
    (128/192/256 -> HR 512/768/1024).  The reference's discriminators only accept HR 96/128/192
    crops, so at these sizes there is no GAN step to reproduce: generator-only, as SURVEY.md reads it.
    One bench "step" = one optimizer iteration per bucket (16x128^2, 8x192^2, 4x256^2 LR per GPU: ~45 GB of saved
    activations, sized for 288 GB HBM).  Returns (seconds per step, per-bucket ms from HIP events, lr pixels)."""
    from esrganplus_amd import architecture as arch, synth, dp as DP, losses as LS, functional as Fn
    prec = args.precision
    dp_on = (world > 1) if data_parallel is None else bool(data_parallel)
    netG = arch.RRDBNet(3, 3, 64, NB).to(dev).train().set_precision(prec)
    netG.load_state_dict(synth.rrdbnet_state_dict(NB, 0, gain=0.5))
    if dp_on:
        DP.broadcast_parameters(netG)
    from esrganplus_amd.optim import FusedAdam
    opt = FusedAdam(netG.parameters(), lr=1e-4, betas=(0.9, 0.999))
    ex = DP.GradExchange(netG, enabled=dp_on, measure=dp_on)
    S = 1024.0 if prec == 'fp16' else 1.0
    scale = torch.full((), S, dtype=torch.float32, device=dev)
    inv = 1.0 / S
    # ESR_GTRAIN_MANUAL=0: the bucket loop through autograd (netG(lr), losses.l1_loss, torch.autograd.backward) — the
    # same launch lists plus ~290 torch glue launches per step; default: driven directly, as train.ESRGANPlusStep does
    manual = os.environ.get('ESR_GTRAIN_MANUAL', '1') != '0' and netG.flat_param_grads
    buckets = []
    for k, (n, sz) in enumerate(GTRAIN_BUCKETS):
        lr = synth.image_batch(400 + 10 * rank + k, n, 3, sz, sz, name='bench.glr').to(dev)
        hr = synth.image_batch(500 + 10 * rank + k, n, 3, 4 * sz, 4 * sz, name='bench.ghr').to(dev)
        buckets.append((lr, hr))
    lr_pix = sum(l.shape[0] * l.shape[2] * l.shape[3] for l, _ in buckets)
    marks = []

    gys = [torch.empty_like(h) for _, h in buckets] if manual else None

    def step(timed=False):
        for k, (lr, hr) in enumerate(buckets):
            if timed:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                marks.append(e)
            if not netG.mark_grads_stale():
                opt.zero_grad(set_to_none=True)
            if manual:
                with torch.no_grad():
                    fake, stG = Fn.rrdbnet_train_forward(netG, lr)
                    loss = LS.l1_raw(fake, hr, 1.0, grad_out=gys[k], grad_scale=S)     # loss + its gradient: one launch
                    del fake
                    Fn.rrdbnet_train_backward(netG, stG, gys[k])
            else:
                loss = LS.l1_loss(netG(lr), hr)
                torch.autograd.backward([loss], [scale])
            ex.start()
            ex.wait()
            opt.step(grad_scale=inv)
            netG.prepack(fwd=True, dgrad=True)           # the next bucket's weight packs, right behind the update
        if timed:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append(e)
        return loss

    for _ in range(max(warmup, 1)):
        step()
    _sync_all(dist)
    ex.reset_counters()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step(True)
    torch.cuda.synchronize()
    elapsed = _max_over_ranks(time.perf_counter() - t0, dist, dev)
    assert torch.isfinite(loss).all()
    if report is not None:
        report.update(calls_per_step=ex.calls / steps, bytes_per_step=ex.bytes / steps,
                      exposed_ms_per_step=ex.exposed_ms() / steps, manual=bool(manual),
                      n_params=sum(p.numel() for p in netG.parameters()))
    nbk = len(buckets) + 1
    per = [0.0] * len(buckets)
    for s_ in range(steps):
        ev = marks[s_ * nbk:(s_ + 1) * nbk]
        for k in range(len(buckets)):
            per[k] += ev[k].elapsed_time(ev[k + 1]) / steps
    return elapsed / steps, per, lr_pix


def gtrain_fields(dt, per, lr_pix, world=1):
    step_flops = 3.0 * 2.0 * MAC_PER_LR_PIXEL * lr_pix      # fwd + dgrad + wgrad (first-layer dgrad omitted: <0.1 %)
    bk = {}
    for (n, sz), ms in zip(GTRAIN_BUCKETS, per):
        fl = 3.0 * 2.0 * MAC_PER_LR_PIXEL * n * sz * sz
        tiles = ((sz + 15) // 16) * ((sz + 31) // 32)
        bk['%dx%d^2' % (n, sz)] = {'ms': round(ms, 3), 'tflops': round(fl / (ms * 1e-3) / 1e12, 1),
                                   'frac_of_f16_mfma_peak': round(fl / (ms * 1e-3) / 1e12 / PEAK_F16_TFLOPS, 4),
                                   'tiles_16x32_per_image': tiles, 'workgroups': n * tiles}
    return {'ms_per_step': round(dt * 1e3, 3), 'value': round(world * 16 * lr_pix / 1e6 / dt, 3), 'unit': 'HR-Mpix/s',
            'buckets_per_gpu_step': ' + '.join('%dx%d^2' % b for b in GTRAIN_BUCKETS),
            'tflops_per_gpu': round(step_flops / dt / 1e12, 1),
            'frac_of_f16_mfma_peak': round(step_flops / dt / 1e12 / PEAK_F16_TFLOPS, 4), 'buckets': bk}


def gtrain_bench(args, world, rank, dev, dist):
    dt, per, lr_pix = measure_gtrain(args, world, rank, dev, dist, args.steps, args.warmup)
    if rank == 0:
        f = gtrain_fields(dt, per, lr_pix, world)
        res = {'metric': 'HR megapixels/sec (x4 SR) generator train step (noise on, L1)', 'unit': 'HR-Mpix/s',
               'value': f['value'], 'n_gpus': world,
               'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': f['ms_per_step'],
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
               'dtype': 'f16' if args.precision == 'fp16' else 'f32', 'data': 'synthetic',
               'config': {'workload': 'nESRGAN+ generator (RRDBNet nb=23, GaussianNoise on) fwd+bwd+Adam, L1 loss, mixed LR '
                                      'tiles bucketed by size: 16x128^2 + 8x192^2 + 4x256^2 per GPU per step (BASELINE configs[4])',
                          'lr_pixels_per_gpu_step': lr_pix, 'parallelism': 'dp%d, RCCL grad all-reduce per bucket' % world},
               'tflops_per_gpu': f['tflops_per_gpu'], 'buckets': f['buckets']}
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def fwd_bwd_probe(args, dev, steps=20):
    """BASELINE.json's metric string says "fwd+bwd": the same workload as `value` (batch 16 of 128x128 LR, fp16)
    with the generator in training mode — GaussianNoise on, every activation kept, L1 loss against a synthetic
    HR target, loss-scaled backward through every conv (dgrad + wgrad); no optimizer step.  Reported next to
    the forward headline, not instead of it (north_star's roofline target is on the forward).  Carries its own
    `roofline` object: the three fused dense-block kernels (training forward chain, backward chain, weight
    gradients) timed with HIP events on the launch stream, the slowest of them named as the dominant kernel."""
    from esrganplus_amd import architecture as arch, synth, losses as LS
    netG = arch.RRDBNet(3, 3, 64, NB).to(dev).train().set_precision('fp16')
    netG.load_state_dict(synth.rrdbnet_state_dict(NB, 0, gain=0.5))
    lr = synth.image_batch(300, args.batch, 3, args.lr, args.lr, name='bench.fb.lr').to(dev)
    hr = synth.image_batch(301, args.batch, 3, 4 * args.lr, 4 * args.lr, name='bench.fb.hr').to(dev)

    scale = torch.full((), 1024.0, dtype=torch.float32, device=dev)

    def step():
        if not netG.mark_grads_stale():
            for q in netG.parameters():
                q.grad = None
        loss = LS.l1_loss(netG(lr), hr)                  # the library's loss kernel (csrc/loss_kernels.hip), one launch
        torch.autograd.backward([loss], [scale])
        return loss

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    assert torch.isfinite(loss).all()
    fl = 3.0 * 2.0 * MAC_PER_LR_PIXEL * args.batch * args.lr * args.lr      # fwd + dgrad + wgrad
    res = {'ms_per_step': round(dt * 1e3, 3), 'value': round(args.batch * (4 * args.lr) ** 2 / 1e6 / dt, 2),
           'unit': 'HR-Mpix/s', 'tflops': round(fl / dt / 1e12, 1),
           'frac_of_f16_mfma_peak': round(fl / dt / 1e12 / PEAK_F16_TFLOPS, 4), 'steps': steps, 'warmup': 3,
           'what': 'RRDBNet x4 train-mode forward (noise on) + backward (dgrad + wgrad, loss scale 1024), '
                   'batch %d of %dx%d LR, fp16, no optimizer step' % (args.batch, args.lr, args.lr)}

    k = generator_kernel_times(netG, lr, args.batch, args.lr)
    if k:
        res.update(k)
    return res


def main():
    global GTRAIN_BUCKETS
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--train-last', action='store_true', help='measure the train_step object behind the forward / fwd+bwd passes (rounds 1-5) instead of first')
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=BATCH)
    ap.add_argument('--lr', type=int, default=LR)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-mfma-probe', action='store_true', help='forward mode: skip the sustained-MFMA-rate probe (1.6 s)')
    ap.add_argument('--no-fwd-bwd', action='store_true', help='forward mode: skip the fwd+bwd side measurement')
    ap.add_argument('--dp-steps', type=int, default=20, help='timed steps of the dp_train object (N > 1); 3 warm-up steps')
    ap.add_argument('--no-train', action='store_true',
                    help='forward mode: skip the train_step / gtrain objects (N = 1) and the dp_train object (N > 1)')
    ap.add_argument('--mode', choices=['forward', 'train', 'gtrain'], default='forward',
                    help="'forward' = BASELINE configs[1] (the headline metric); 'train' = configs[2]/[3]: "
                         'full ESRGAN+ step, batch 16 of 32x32 LR per GPU, DP over RCCL; '
                         "'gtrain' = configs[4]: noise-on generator fwd+bwd+Adam on mixed 128/192/256 LR tiles")
    ap.add_argument('--gtrain-buckets', default=','.join('%dx%d' % b for b in GTRAIN_BUCKETS),
                    help='configs[4] buckets as COUNTxLR_SIZE per GPU and step (default: the benchmarked 16x128,8x192,4x256; '
                         'smaller ones only for dry runs with several ranks on one GPU)')
    ap.add_argument('--precision', choices=['fp16', 'fp32'], default='fp16')
    ap.add_argument('--train-batch', type=int, default=16,
                    help="--mode train: LR tiles per GPU per step (the reference's config uses 16)")
    args = ap.parse_args()
    GTRAIN_BUCKETS = tuple(tuple(int(v) for v in b.split('x')) for b in args.gtrain_buckets.split(','))

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world == 1:
        # not under torchrun: re-launch one process per GPU over RCCL
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.execvp(sys.executable, [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
                                   '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
                                   '--master-port', os.environ.get('MASTER_PORT', '29533'),
                                   os.path.abspath(__file__)] + sys.argv[1:])
    assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs'
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    # ESR_DP_FORCE=1 (esrganplus_amd.dp.forced): a ONE-rank process group over the real RCCL backend, and the dp_train /
    # dp_gtrain objects at N = 1 — every data-parallel branch runs (AVG all-reduce of the gradient buckets inside the
    # segmented backward, global RaGAN means, stream-side waits), only the wire is missing
    dp_forced = world == 1 and os.environ.get('ESR_DP_FORCE', '0') == '1'
    if dp_forced:
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        os.environ.setdefault('MASTER_PORT', '29543')
    if world > 1 or dp_forced:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        backend = os.environ.get('ESR_BENCH_BACKEND', 'nccl')     # 'gloo' only for single-GPU dry runs
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    from esrganplus_amd import architecture as arch, synth, engine as E
    if args.mode == 'train':
        return train_bench(args, world, rank, dev, dist)
    if args.mode == 'gtrain':
        return gtrain_bench(args, world, rank, dev, dist)
    # The train step (BASELINE configs[2]) is measured FIRST, on the box as the driver hands it over: it is a chain of small,
    # latency-bound launches whose time follows the shader clock, and behind the seconds of full-chip load of the forward /
    # fwd+bwd passes below the same step measured 4 % slower (6.60 vs 6.31 ms, same box, same process; --train-last: A/B).
    # It is a light load (0.18 of the MFMA peak): it does not warm the chip for the headline that follows.
    train_first = None
    if world == 1 and not args.no_train and not args.train_last:
        train_first, st_ = train_object(args, 1, 0, dev, None, steps=30, warmup=5)
        del st_
        torch.cuda.empty_cache()
    net = arch.RRDBNet(3, 3, 64, NB).to(dev).eval().set_precision('fp16')
    net.load_state_dict(synth.rrdbnet_state_dict(NB, 0), strict=True)
    x = synth.image_batch(100 + rank, args.batch, 3, args.lr, args.lr, name='bench.x').to(dev)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # exactly W warm-up steps, as the contract says (the first forward builds the launch plan and packs the weights; the
    # first ~10 forwards of a process run 1-2 % below the steady state, tools/step_trend.py — a caller who asks for
    # fewer warm-up steps than that measures that)
    with torch.no_grad():
        for _ in range(max(args.warmup, 1) if args.warmup > 0 else 0):
            y = net(x)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            y = net(x)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == 'nccl' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        assert torch.isfinite(y).all()

    ms_per_step = elapsed / args.steps * 1e3
    hr_mpix_per_step = world * args.batch * (4 * args.lr) ** 2 / 1e6
    value = hr_mpix_per_step / (elapsed / args.steps)
    step_flops = 2.0 * MAC_PER_LR_PIXEL * args.batch * args.lr * args.lr

    res = {'metric': 'HR megapixels/sec (x4 SR) RRDBNet forward (fwd+bwd: the fwd_bwd object of this line)', 'value': round(value, 2),
           'unit': 'HR-Mpix/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
           'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'weak',
           'vs_baseline': None, 'dtype': 'f16', 'data': 'synthetic',
           'config': {'workload': 'RRDBNet x4 (23 RRDB, nf=64, gc=32) fp16 forward-only, batch %d of '
                                  '%dx%d LR tiles per GPU -> %dx%d HR (BASELINE configs[1])'
                                  % (args.batch, args.lr, args.lr, 4 * args.lr, 4 * args.lr),
                      'global_batch': world * args.batch, 'parallelism': 'dp%d (independent tiles, no collective)' % world,
                      'weights': 'synthetic default-init seed 0'},
           'whole_net_tflops_per_gpu': round(step_flops / (elapsed / args.steps) / 1e12, 1),
           'whole_net_frac_of_f16_mfma_peak': round(step_flops / (elapsed / args.steps) / 1e12 / PEAK_F16_TFLOPS, 4)}

    if rank == 0:
        # ---- roofline of the dominant kernel: per-op HIP-event timing on the launch stream
        with torch.no_grad():
            plan = next(iter(net._plans.values()))
            desc = describe_plan(net, plan)
            st = E.current_stream()
            agg = {}
            reps = 8
            for _ in range(2):                       # untimed: the clocks settle again after the host-side gap above
                plan.ops.run_timed(st)
            for _ in range(reps):
                ms = plan.ops.run_timed(st)
                for (name, fl), t in zip(desc, ms):
                    a = agg.setdefault(name, [0.0, 0.0, 0])
                    a[0] += t
                    a[1] += fl
                    a[2] += 1
        dom = max((k for k in agg if k != 'layout'), key=lambda k: agg[k][0])
        tot_ms = sum(a[0] for a in agg.values()) / reps
        t_ms, fl, n = agg[dom]
        ach = fl / (t_ms * 1e-3) / 1e12
        # PMC-measured HBM-side bytes per launch: NOT measured by this run — the committed summary of separate
        # rocprofv3 --pmc passes over this same command (MI355X_MICROARCH.md HBM section), quoted only while the
        # kernel's sources are the ones it was measured on (tools/traffic_hashes.py)
        traffic, traffic_src = committed_traffic(dom) if (args.batch == BATCH and args.lr == LR) else (None, None)
        res['roofline'] = {'bound': 'mfma', 'kernel': dom, 'achieved': round(ach, 1), 'peak': PEAK_F16_TFLOPS,
                           'unit': 'TFLOP/s', 'frac': round(ach / PEAK_F16_TFLOPS, 4), 'traffic': traffic,
                           'traffic_source': traffic_src,
                           'launches_per_step': n // reps, 'avg_launch_us': round(t_ms / n * 1e3, 2),
                           'share_of_step_time': round(t_ms / reps / tot_ms, 3)}
        prow = committed_profile(dom) if (args.batch == BATCH and args.lr == LR) else None
        if prow:
            # the committed rocprofv3 figure of the same command — ANOTHER box than this run's (box spread +-3.5 %)
            res['roofline']['profiled_avg_launch_us'] = prow['profiled_avg_launch_us']
            res['roofline']['profiled_frac'] = round(fl / n / (prow['profiled_avg_launch_us'] * 1e-6) / 1e12 / PEAK_F16_TFLOPS, 4)
            res['roofline']['profiled_source'] = '%s (rocprofv3 --kernel-trace --stats, %d launches incl. the cold one; a different box than this line)' % (
                prow.get('profiled_source', 'profiles/'), prow.get('profiled_calls', 0))
        if traffic:
            # secondary bound: HBM-side bytes of that launch over its duration vs 8 TB/s
            gbps = traffic / (t_ms / n * 1e-3) / 1e9
            res['roofline']['hbm_achieved_GBps'] = round(gbps, 1)
            res['roofline']['hbm_frac'] = round(gbps / 8000.0, 4)
        res['kernels'] = {k: {'launches': a[2] // reps, 'ms_per_step': round(a[0] / reps, 3),
                              'tflops': round(a[1] / (a[0] * 1e-3) / 1e12, 1) if a[1] else 0.0}
                          for k, a in sorted(agg.items())}
        if world == 1 and not args.no_fwd_bwd:
            del y
            net = None
            torch.cuda.empty_cache()
            res['fwd_bwd'] = fwd_bwd_probe(args, dev)
            # BASELINE.json's metric string says "fwd+bwd": the same figure at the top level of the line
            res['fwd_bwd_value'] = res['fwd_bwd'].get('value')
            res['fwd_bwd_ms_per_step'] = res['fwd_bwd'].get('ms_per_step')
        if world == 1 and not args.no_train:
            # BASELINE configs[2] and configs[4] next to the headline, so that the driver's line carries them
            torch.cuda.empty_cache()
            if train_first is not None:
                res['train_step'] = train_first
            else:
                res['train_step'], st_ = train_object(args, 1, 0, dev, None, steps=30, warmup=5)
                del st_
            torch.cuda.empty_cache()
            dt, per, lr_pix = measure_gtrain(args, 1, 0, dev, None, steps=5, warmup=2)
            res['gtrain'] = dict(gtrain_fields(dt, per, lr_pix), steps=5, warmup=2,
                                 what='nESRGAN+ generator (noise on) fwd+bwd+Adam, L1 loss, 16x128^2 + 8x192^2 + 4x256^2 '
                                      'LR tiles per step, fp16 (BASELINE configs[4], one GPU)')
            torch.cuda.empty_cache()
        if world == 1 and not args.no_mfma_probe:
            # (last of the GPU work: 1.6 s at the power limit would otherwise warm the box for the objects above)
            torch.cuda.empty_cache()
            pr = mfma_sustained_probe(dev)
            res['mfma_sustained_probe'] = pr
            res['roofline']['frac_of_sustained_probe'] = round(ach / pr['tflops'], 4)
            fb = res.get('fwd_bwd', {}).get('roofline')
            if fb and fb.get('unit') == 'TFLOP/s':       # (the training chains fill the chip as well; the train step's do not)
                fb['frac_of_sustained_probe'] = round(fb['achieved'] / pr['tflops'], 4)
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline()
    if dist is not None:
        # proof of what ran: ranks in the group, backend, the RCCL the collectives went through
        res['dist'] = {'world_size': dist.get_world_size(), 'backend': dist.get_backend(),
                       'rccl_version': '.'.join(str(v) for v in torch.cuda.nccl.version()) if dist.get_backend() == 'nccl' else None,
                       'forced_one_rank': bool(dp_forced)}
    if (world > 1 or dp_forced) and not args.no_train:
        # BASELINE configs[3]: every rank runs the data-parallel train step (the collective path of this repo); the
        # forward headline above has no data-path collective (independent tiles)
        net = None
        torch.cuda.empty_cache()
        dpo = dp_train_object(args, world, rank, dev, dist, steps=args.dp_steps)
        torch.cuda.empty_cache()
        dpg = dp_gtrain_object(args, world, rank, dev, dist, steps=max(2, args.dp_steps // 4))
        if rank == 0:
            res['dp_train'] = dpo
            res['dp_gtrain'] = dpg
    if rank == 0:
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
