"""Process-per-GPU data parallelism over RCCL/xGMI (torch.distributed backend "nccl" == RCCL).

The reference's only multi-GPU mechanism is single-process ``nn.DataParallel``
(codes/models/networks.py:105-107,136-137,152-153), which re-broadcasts parameters every forward,
gathers outputs to device 0 and cannot run with the noise layers (block.py:115 pins them to cuda:0).
Here every rank owns one MI355X and a full replica; per optimizer there is exactly ONE exchange:
the mean all-reduce of that network's gradients.

* ``RRDBNet`` / ``Discriminator`` backward are single autograd nodes that emit ALL parameter
  gradients as views of one flat fp32 tensor, so the exchange all-reduces slices ("buckets") of that
  tensor in place — no gather/scatter copies.
* ``RRDBNet``'s node runs its backward list in SEGMENTS (tail, RRDB 22 .. 0, first conv) and hands every
  finished slice to the exchange while the remaining segments are still being enqueued: the all-reduce of
  the deep layers' gradients runs under the backward of the shallow ones (``GradExchange(overlap=True)``,
  ``functional._train_backward``).  Other modules' buckets are issued right after ``backward()`` and only
  waited for before ``optimizer.step()``, overlapping the other network's pass.
* xGMI is point-to-point (7 links x ~153 GB/s per GPU): bucket size defaults to 32 MiB so each
  ring/tree step moves >= 4 MiB per link — large enough to be bandwidth- rather than latency-bound.
* The relativistic-average GAN terms use the mean of D's logits over the GLOBAL batch
  (SRRaGAN_model.py:136-137,151-152; under DataParallel the loss is formed on gathered outputs):
  ``global_mean`` all-reduces [sum, count] and is differentiable.
* D's BatchNorm keeps per-replica statistics, exactly as under DataParallel (no SyncBN).
"""
import os

import torch
import torch.distributed as dist


def forced():
    """ESR_DP_FORCE=1: take every data-parallel branch (bucketed exchange inside the segmented backward, global RaGAN
    means, async handles + stream-side waits) even at world size 1, over whatever backend the process group has.  On
    the ONE GPU a box here has this is the only way the real RCCL backend ('nccl') ever runs under this code:
    ``ReduceOp.AVG``, ``device_id=``, the Work handles' stream semantics and RCCL's launch path next to the resident
    chains are exercised — everything except the wire (tests/test_gpu_dp.py, bench.py with the knob set)."""
    return os.environ.get('ESR_DP_FORCE', '0') == '1'


def active():
    """True when the data-parallel paths run: a process group of more than one rank, or a forced one (``forced``)."""
    return dist.is_initialized() and (dist.get_world_size() > 1 or forced())


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run).  World size 1 needs
    no process group — unless ESR_DP_FORCE=1 asks for the data-parallel paths anyway (a one-rank group)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if dist.is_initialized() or (world == 1 and not forced()):
        return world
    if world == 1:
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        os.environ.setdefault('MASTER_PORT', '29541')
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    kw = {}
    if backend == 'nccl':
        local = int(os.environ.get('LOCAL_RANK', '0'))
        torch.cuda.set_device(local)
        kw['device_id'] = torch.device('cuda', local)
    dist.init_process_group(backend, **kw)
    return world


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def broadcast_parameters(module, src=0):
    """One-time parameter/buffer sync at start-up (replicas then stay identical because every rank
    applies the same averaged gradients) — replaces DataParallel's per-forward broadcast."""
    if not active():
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t, src)


def flat_grad_spans(params):
    """Gradients grouped by storage: for every group of ``.grad`` tensors that are views tiling one
    contiguous span of a flat buffer (what the fused backward nodes emit) return that span as ONE 1-D
    tensor; all other grads are returned individually.  -> (spans, loose)"""
    groups = {}
    for p in params:
        if p.grad is not None:
            groups.setdefault(p.grad.untyped_storage().data_ptr(), []).append(p.grad)
    spans, loose = [], []
    for gs in groups.values():
        if len(gs) > 1 and all(g.is_contiguous() for g in gs):
            lo = min(g.storage_offset() for g in gs)
            hi = max(g.storage_offset() + g.numel() for g in gs)
            if sum(g.numel() for g in gs) == hi - lo:
                spans.append(torch.empty(0, dtype=gs[0].dtype, device=gs[0].device).set_(
                    gs[0].untyped_storage(), lo, (hi - lo,)))
                continue
        loose.extend(gs)
    return spans, loose


class GradExchange:
    """Mean all-reduce of a module's gradients, bucketed, asynchronous."""

    def __init__(self, module, bucket_bytes=32 << 20, overlap=True, enabled=True, measure=False):
        # enabled=False: a no-op stand-in (one rank's own step inside a multi-rank job: bench.py's no-exchange figure)
        # measure=True: HIP events around every wait on the compute stream -> exposed_ms() (bench.py dp_train)
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.bucket_bytes = bucket_bytes
        self.handles = []
        self._staged = []
        self.enabled = bool(enabled)
        self.measure = bool(measure)
        self.calls = 0
        self.bytes = 0
        self._events = []
        if not self.enabled:
            self.inline = False
            self.bucket_elems = max(1, bucket_bytes // 4)
            if getattr(module, '_grad_sync', None) is not None:
                module.attach_grad_sync(None)
            return
        # In-backward exchange (networks.py:105-107 reduces implicitly inside DataParallel's backward): a module
        # whose backward is one fused node (RRDBNet) runs it in segments and calls `reduce_slice` on every
        # finished span of its flat gradient buffer, so the all-reduce of the last layers' gradients runs
        # under the backward of the first ones.  The gradients that reach `.grad` are then already averaged:
        # start() / wait() skip them.
        self.inline = False
        self.bucket_elems = max(1, bucket_bytes // 4)
        if overlap and active() and hasattr(module, 'attach_grad_sync'):
            module.attach_grad_sync(self)
            self.inline = True

    def __call__(self, flat_slice):
        """Mean all-reduce of one finished slice (called from inside the backward node, stream-ordered after
        the kernels that wrote it).  Returns the work handle (None on one rank)."""
        if not active():
            return None
        self.calls += 1
        self.bytes += flat_slice.numel() * flat_slice.element_size()
        if dist.get_backend() == 'nccl':
            return dist.all_reduce(flat_slice, op=dist.ReduceOp.AVG, async_op=True)
        flat_slice.div_(float(world_size()))            # gloo has no AVG
        return dist.all_reduce(flat_slice, async_op=True)

    def wait_handles(self, handles):
        """Make the current stream wait for these exchanges; with measure=True the wait is bracketed by HIP events on
        that stream: their distance is the time the compute stream spent blocked on communication."""
        handles = [h for h in handles if h is not None]
        if not handles:
            return
        if self.measure and torch.cuda.is_available():
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        for h in handles:
            h.wait()
        if self.measure and torch.cuda.is_available():
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self._events.append((e0, e1))

    def reset_counters(self):
        self.calls, self.bytes, self._events = 0, 0, []

    def exposed_ms(self):
        """Sum over the measured waits (call after a device synchronize)."""
        return sum(a.elapsed_time(b) for a, b in self._events)

    def _flat_groups(self):
        """Group parameter grads by underlying storage: grads that are views of one flat tensor
        (our fused backward nodes) are reduced in place, slice by slice."""
        groups = {}
        for p in self.params:
            if p.grad is None:
                continue
            st = p.grad.untyped_storage()
            groups.setdefault(st.data_ptr(), []).append(p)
        return groups

    def start(self):
        """Issue the all-reduces (async).  Call right after ``loss.backward()``."""
        self.handles, self._staged = [], []
        if not active() or self.inline or not self.enabled:
            return
        ws = float(world_size())
        for _, ps in self._flat_groups().items():
            g0 = ps[0].grad
            esz = g0.element_size()
            if len(ps) > 1:
                # contiguous span covered by these views inside the shared storage
                lo = min(p.grad.storage_offset() for p in ps)
                hi = max(p.grad.storage_offset() + p.grad.numel() for p in ps)
                covered = sum(p.grad.numel() for p in ps)
                if covered == hi - lo:
                    flat = torch.empty(0, dtype=g0.dtype, device=g0.device).set_(
                        g0.untyped_storage(), lo, (hi - lo,))
                    step = max(1, self.bucket_bytes // esz)
                    # last slices first: they belong to the layers whose backward finished first
                    for s in range(((hi - lo - 1) // step) * step, -1, -step):
                        chunk = flat[s:s + step]
                        self.calls += 1
                        self.bytes += chunk.numel() * esz
                        if dist.get_backend() == 'nccl':
                            self.handles.append(dist.all_reduce(chunk, op=dist.ReduceOp.AVG, async_op=True))
                        else:
                            chunk.div_(ws)
                            self.handles.append(dist.all_reduce(chunk, async_op=True))
                    continue
            # generic path: stage into a bucket, reduce, copy back on wait()
            bucket, size = [], 0
            for p in reversed(ps):
                bucket.append(p)
                size += p.grad.numel() * esz
                if size >= self.bucket_bytes:
                    self._launch_bucket(bucket, ws)
                    bucket, size = [], 0
            if bucket:
                self._launch_bucket(bucket, ws)

    def _launch_bucket(self, ps, ws):
        flat = torch.cat([p.grad.reshape(-1) for p in ps]).div_(ws)
        self.calls += 1
        self.bytes += flat.numel() * flat.element_size()
        self.handles.append(dist.all_reduce(flat, async_op=True))
        self._staged.append((flat, ps))

    def wait(self):
        """Block the current stream on the exchange; call right before ``optimizer.step()``."""
        self.wait_handles(self.handles)
        for flat, ps in self._staged:
            off = 0
            for p in ps:
                n = p.grad.numel()
                p.grad.copy_(flat[off:off + n].view_as(p.grad))
                off += n
        self.handles, self._staged = [], []


class _GlobalMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        # torch.full, not new_tensor(scalar): the latter is a synchronous host->device copy (2 ms stall)
        s = torch.stack([x.sum(), torch.full((), float(x.numel()), dtype=x.dtype, device=x.device)])
        if active():
            dist.all_reduce(s)
        ctx.save_for_backward(s[1:2])
        ctx.shape = x.shape
        return s[0] / s[1]

    @staticmethod
    def backward(ctx, g):
        (n,) = ctx.saved_tensors
        # d(mean_global)/dx_local = 1/N_global for every local element, and EVERY rank's loss depends
        # on the mean, so the upstream gradient is summed over ranks first.  With per-rank losses =
        # local means and parameter gradients averaged over ranks (GradExchange) this reproduces the
        # gradient of the reference's single global-batch loss.
        if active():
            g = g.clone()
            dist.all_reduce(g)
        return (g / n).expand(ctx.shape)


def global_mean(x):
    """mean of ``x`` over the batch of ALL ranks (== torch.mean on one rank), differentiable."""
    return _GlobalMean.apply(x)
