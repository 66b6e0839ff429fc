"""Host-side launch planner for the HIP hot path.

Turns one pass of a drop-in module (``RRDBNet.forward`` — reference
codes/models/modules/architecture.py:76-78 — or a single ``ResidualDenseBlock_5C`` / ``RRDB``,
block.py:260-268 / 287-291) into a recorded list of fused-conv launches over G32 buffers
(include/esrgan_hip.h) that ONE C call replays on torch's current HIP stream.

torch is used here for device memory (buffers are torch tensors) and the stream handle only.
"""
import ctypes as C

import os

import torch

from . import _lib as L

SIGMA = 0.1   # GaussianNoise sigma (block.py:111)

_STORE_FLAVOUR = int(os.environ.get('ESR_STORE_FLAVOUR', '0'))   # experiment knob: 1 = nt epilogue stores


def _dt(dtype):
    if dtype in ('fp16', torch.float16, L.ESR_F16):
        return L.ESR_F16, torch.float16, 16
    if dtype in ('fp32', torch.float32, L.ESR_F32):
        return L.ESR_F32, torch.float32, 8
    raise ValueError('dtype must be fp16 or fp32, got %r' % (dtype,))


def require_cuda(t, what):
    if not t.is_cuda:
        raise L.HipExtensionError(
            'esrganplus_amd: %s is on %s — the HIP path needs a CUDA/HIP tensor on an MI355X; '
            'there is no CPU fallback (use oracle/ for CPU reference results).' % (what, t.device))


# ESR_SIDE=0: weight-gradient runs on the launch stream instead of the side stream (A/B)
_SIDE = L.OPF_SIDE if os.environ.get('ESR_SIDE', '1') != '0' else 0
_SIDE_FREE = L.OPF_SIDE_FREE if os.environ.get('ESR_TAIL_WGRAD_FREE', '1') != '0' else 0     # (A/B knob, round 5)

class G32:
    """[B][ngroups][Hp][Wp][cpg] activation buffer with a physical zero halo."""

    def __init__(self, B, C_, H, W, dtype, device):
        self.esr_dtype, self.tdtype, self.cpg = _dt(dtype)
        self.B, self.C, self.H, self.W = B, C_, H, W
        self.ng = (C_ + self.cpg - 1) // self.cpg
        self.Hp, self.Wp = L.g32_dims(H, W)
        self.t = torch.zeros(B, self.ng, self.Hp, self.Wp, self.cpg, dtype=self.tdtype, device=device)
        self.gs = self.Hp * self.Wp * 32
        self.bs = self.ng * self.gs

    def view(self, c0=0, nch=None, row0=0):
        """esr_g32 view starting at channel c0 (must be group aligned) and, for the taller buffers of the banded
        chain, at row row0 (the rows above / below the view's image are zero padding that nothing writes)."""
        assert c0 % self.cpg == 0, (c0, self.cpg)
        g0 = c0 // self.cpg
        ng = self.ng - g0 if nch is None else (nch + self.cpg - 1) // self.cpg
        v = L.esr_g32()
        v.ptr = self.t.data_ptr() + g0 * self.gs + row0 * self.Wp * 32
        v.batch_stride = self.bs
        v.group_stride = self.gs
        v.wp = self.Wp
        v.ngroups = ng
        return v

    def groups(self, nch):
        return (nch + self.cpg - 1) // self.cpg


class ConvW:
    """Packed weights of one conv (a slice of a WeightPack arena)."""
    __slots__ = ('key', 'cout', 'cin', 'ks', 'w_ptr', 'bias_ptr', 'has_bias', 'subpix')


class WeightPack:
    """MFMA-fragment-ordered copies of a module's conv weights (fp32 OIHW nn.Parameters stay the
    master copy — networks.py:30-44 pokes ``m.weight.data``, Adam updates them in place).
    ``ensure()`` re-packs (one launch per tensor, recorded once) whenever a parameter's storage
    or version changed, or unconditionally when ``force``."""

    def __init__(self, convs, dtype, device, subpix=()):
        # convs: list of (key, weight_param, bias_param_or_None); subpix: keys of up-convs (nearest x2 + 3x3,
        # block.py:315-322) packed in the 4-phase 2x2 form (esr_pack.ups_fwd, run with esr_conv.upsample = 3)
        self.esr_dtype, self.tdtype, self.cpg = _dt(dtype)
        self.subpix = frozenset(subpix)
        self.device = device
        self.convs = convs
        self.entries = {}
        total = 0
        offs = []
        for key, w, b in convs:
            cout, cin, ks, _ = w.shape
            offs.append(total)
            if key in self.subpix:      # 4 phases x cout blocks, 2x2 taps
                total += L.packed_weight_bytes(4 * 32 * ((cout + 31) // 32), cin, 2, self.esr_dtype)
            else:
                total += L.packed_weight_bytes(cout, cin, ks, self.esr_dtype)
        self.arena = torch.zeros(total, dtype=torch.uint8, device=device)
        nb_pad = sum(((w.shape[0] + 31) // 32) * 32 for _, w, b in convs if b is not None and w.shape[0] % 32)
        self.bias_arena = torch.zeros(max(nb_pad, 1), dtype=torch.float32, device=device)
        self._bias_copies = []
        bo = 0
        for (key, w, b), off in zip(convs, offs):
            e = ConvW()
            e.key, (e.cout, e.cin, e.ks) = key, w.shape[:3]
            e.subpix = key in self.subpix
            e.w_ptr = self.arena.data_ptr() + off
            e.has_bias = b is not None
            e.bias_ptr = None
            if b is not None:
                if e.cout % 32 == 0:
                    e.bias_ptr = None      # filled in _rebuild (points at the parameter itself)
                else:
                    n = ((e.cout + 31) // 32) * 32
                    e.bias_ptr = self.bias_arena.data_ptr() + 4 * bo
                    self._bias_copies.append((b, self.bias_arena[bo:bo + e.cout]))
                    bo += n
            self.entries[key] = e
        self._sig = None
        self._ptrs = None
        self.ops = None
        self.generation = 0      # bumped when any pointer handed out may have changed
        self.pack_count = 0      # bumped every time the arena is re-packed (RdbStreams re-gathers)

    def _check_params(self):
        for key, w, b in self.convs:
            for p in (w, b):
                if p is None:
                    continue
                require_cuda(p, 'parameter ' + key)
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise L.HipExtensionError('parameter %s must be contiguous fp32 (master weights)' % key)

    def _rebuild(self):
        self._check_params()
        packs = []
        for key, w, b in self.convs:
            e = self.entries[key]
            pk = L.esr_pack()
            pk.src = w.data_ptr()
            pk.dst = e.w_ptr
            pk.cout, pk.cin, pk.ks = e.cout, e.cin, e.ks
            pk.dtype = self.esr_dtype
            pk.transpose_flip = 0
            pk.ups_fwd = 1 if e.subpix else 0
            packs.append(pk)
            if b is not None and e.cout % 32 == 0:
                e.bias_ptr = b.data_ptr()
        ops = L.OpList()                      # ONE launch re-packs every conv of the network
        bp, self._pack_keep = L.batch_pack_op(packs, self.device)
        ops.add(L.OP_PACK_BATCH, 'pack_batch', bp)
        self.ops = ops
        self.generation += 1

    def _ptr_fingerprint(self):
        # a few sampled storages + the count: parameters move together (.to / .cuda / DataParallel replicas); the full
        # per-tensor tuple (2 x ~390 data_ptr calls for the generator) costs ~0.2 ms of host time per training step
        c, n = self.convs, len(self.convs)
        idx = sorted({0, n // 3, (2 * n) // 3, n - 1})
        return (n,) + tuple(c[i][1].data_ptr() for i in idx) + tuple(c[i][2].data_ptr() for i in idx if c[i][2] is not None)

    FULL_CHECK_EVERY = 64     # calls between full pointer comparisons (an unsampled parameter re-pointed by hand:
    #                            `p.data = t`, per-layer re-init, load_state_dict(assign=True) — ADVICE r04)

    def ensure(self, stream, force=False, record_sig=False, full=False):
        """full=True (the module saw load_state_dict / _apply / replicate): compare EVERY parameter's storage, not the
        sampled fingerprint; the same happens on every FULL_CHECK_EVERY-th call."""
        fp = self._ptr_fingerprint()
        self._calls = getattr(self, '_calls', 0) + 1
        if full or self._calls % self.FULL_CHECK_EVERY == 0 or fp != getattr(self, '_fp', None):
            ptrs = tuple(p.data_ptr() for _, w, b in self.convs for p in (w, b) if p is not None)
            if ptrs != self._ptrs:
                self._rebuild()
                self._ptrs = ptrs
                self._sig = None
            self._fp = fp
        # training passes re-pack unconditionally (FusedAdam updates through raw pointers: no version bump to see)
        # (record_sig: a forced pack whose result a later non-forced call may rely on — _PlannedModule.prepack)
        sig = None if (force and not record_sig) else tuple(p._version for _, w, b in self.convs for p in (w, b) if p is not None)
        if force or sig != self._sig:
            self.ops.run(stream)
            with torch.no_grad():
                for src, dst in self._bias_copies:
                    dst.copy_(src)
            self._sig = sig
            self.pack_count += 1


class RdbStreams:
    """Fused weight streams of dense blocks for esr_rdb_forward (include/esrgan_hip.h, esr_rdb_block): per
    block the 1 KB MFMA fragments of conv1..conv5 + conv1x1 in the order the kernel's units consume them
    (esr_rdb_block.w in include/esrgan_hip.h) — a pure gather of fragments out of the per-conv packed arena
    of `wp` (WeightPack), run as ONE esr_gather_fragments launch right after every re-pack."""

    def __init__(self, wp, prefixes):
        self.wp, self.prefixes = wp, list(prefixes)
        self.cpg = wp.cpg
        self.stream_bytes = L.lib().esr_rdb_weight_stream_bytes(wp.esr_dtype)
        self.arena = torch.zeros(len(self.prefixes) * self.stream_bytes, dtype=torch.uint8, device=wp.device)
        self.bias = torch.zeros(len(self.prefixes) * 192, dtype=torch.float32, device=wp.device)   # [block][192]
        self._gen = None
        self.ops = None

    def w_ptr(self, i):
        return self.arena.data_ptr() + i * self.stream_bytes

    def bias_ptr(self, i):
        return self.bias.data_ptr() + i * 192 * 4

    def _table(self):
        cpg, base = self.cpg, self.wp.arena.data_ptr()
        kx, kd = 64 // cpg, 32 // cpg
        offs = []
        for p in self.prefixes:
            ent = [self.wp.entries[p + '.conv%d.0' % k] for k in range(1, 6)]
            nch = [(64 + 32 * k) // cpg for k in range(5)]            # input chunks of conv1..conv5

            def frag(blk, c, kh, kw):                                # cout block 0..5 = conv1..conv4, conv5[0:32], conv5[32:64]
                k, cb = (blk, 0) if blk < 4 else (4, blk - 4)
                return ent[k].w_ptr - base + (((cb * nch[k] + c) * 3 + kh) * 3 + kw) * 1024
            e1 = self.wp.entries[p + '.conv1x1']
            one = [e1.w_ptr - base + c * 1024 for c in range(kx)]
            for ph in range(1, 6):                                   # phase = input slice x, x1..x4
                c0 = 0 if ph == 1 else kx + (ph - 2) * kd
                ks = kx if ph == 1 else kd
                if cpg == 8:
                    # fp32: units (K step, column tap) over conv ph..5, the 1x1 at the end of the stream
                    for c in range(ks):
                        for kw in range(3):
                            offs += [frag(blk, c0 + c, kh, kw) for blk in range(ph - 1, 6) for kh in range(3)]
                    continue
                # fp16 (rdb_fused.hip, Sched): crit_p = conv_p alone, one unit per K step (3 kw x 3 kh); bulk_p =
                # conv_{p+1}..conv5, one unit per (K step, kw); the 1x1 after bulk_1; phase 5 = conv5's two blocks
                if ph < 5:
                    for c in range(ks):
                        offs += [frag(ph - 1, c0 + c, kh, kw) for kw in range(3) for kh in range(3)]
                    for c in range(ks):
                        for kw in range(3):
                            offs += [frag(blk, c0 + c, kh, kw) for blk in range(ph, 6) for kh in range(3)]
                    if ph == 1:
                        offs += one
                else:
                    for c in range(ks):
                        for kw in range(3):
                            offs += [frag(blk, c0 + c, kh, kw) for blk in (4, 5) for kh in range(3)]
            if cpg == 8:
                offs += one
        assert len(offs) * 1024 == len(self.prefixes) * self.stream_bytes, (len(offs), self.stream_bytes)
        return offs

    def ensure(self, stream, force=False):
        """Call after wp.ensure(): re-gathers when the packed arena was rewritten."""
        if self.ops is None:
            self._tab = torch.tensor(self._table(), dtype=torch.int64, device=self.wp.device)
            g = L.esr_frag_gather()
            g.src_off, g.src_base, g.dst, g.n = (self._tab.data_ptr(), self.wp.arena.data_ptr(),
                                                 self.arena.data_ptr(), self._tab.numel())
            self.ops = L.OpList()
            self.ops.add(L.OP_FRAG_GATHER, 'frag_gather', g)
            # the biases of a block as one [192] vector: 128-byte pieces straight from the nn.Parameters
            boffs = []
            for p in self.prefixes:
                for k in range(1, 6):
                    bp = self.wp.entries[p + '.conv%d.0' % k].bias_ptr
                    boffs += [bp + 128 * q for q in range(2 if k == 5 else 1)]
            self._btab = torch.tensor(boffs, dtype=torch.int64, device=self.wp.device)
            gb = L.esr_frag_gather()
            gb.src_off, gb.src_base, gb.dst, gb.n, gb.piece_bytes = self._btab.data_ptr(), None, self.bias.data_ptr(), len(boffs), 128
            self.ops.add(L.OP_FRAG_GATHER, 'frag_gather', gb)
            self._bias_src = tuple(boffs)
        gen = (self.wp.generation, self.wp.pack_count)
        if force or gen != self._gen:
            self.ops.run(stream)
            self._gen = gen


class RdbBwdStreams:
    """Fused weight streams of the BACKWARD chain (esr_rdb_backward; include/esrgan_hip.h, "Backward weight stream"):
    per block the 1 KB fragments of its gather-form operands (DgradPack entries .g4 .g3 .c2 .g1 .c0 = cout blocks
    0..3, 4/5) in the forward stream's crit / bulk unit order, the transposed 1x1 (.o1) behind crit_3 — a pure gather
    out of the DgradPack arena, re-run after every re-pack."""

    def __init__(self, dp, prefixes):
        self.dp, self.prefixes = dp, list(prefixes)
        assert dp.esr_dtype == L.ESR_F16
        self.cpg = dp.cpg
        self.stream_bytes = L.lib().esr_rdb_weight_stream_bytes(dp.esr_dtype)
        self.arena = torch.zeros(len(self.prefixes) * self.stream_bytes, dtype=torch.uint8, device=dp.arena.device)
        self._gen, self.ops = None, None

    def w_ptr(self, i):
        return self.arena.data_ptr() + i * self.stream_bytes

    def _table(self):
        cpg, base = self.cpg, self.dp.arena.data_ptr()
        kx, kd = 64 // cpg, 32 // cpg
        offs = []
        for p in self.prefixes:
            ent = [self.dp.entries[p + sfx] for sfx in ('.g4', '.g3', '.c2', '.g1', '.c0')]
            nch = [(64 + 32 * k) // cpg for k in range(5)]            # K chunks of the five slice convs

            def frag(blk, c, kh, kw):
                k, cb = (blk, 0) if blk < 4 else (4, blk - 4)
                return ent[k].w_ptr - base + (((cb * nch[k] + c) * 3 + kh) * 3 + kw) * 1024
            one = [self.dp.entries[p + '.o1'].w_ptr - base + f * 1024 for f in range(4)]
            for ph in range(1, 6):
                c0 = 0 if ph == 1 else kx + (ph - 2) * kd
                ks = kx if ph == 1 else kd
                if ph < 5:
                    for c in range(ks):
                        offs += [frag(ph - 1, c0 + c, kh, kw) for kw in range(3) for kh in range(3)]
                    if ph == 3:
                        offs += one
                    for c in range(ks):
                        for kw in range(3):
                            offs += [frag(blk, c0 + c, kh, kw) for blk in range(ph, 6) for kh in range(3)]
                else:
                    for c in range(ks):
                        for kw in range(3):
                            offs += [frag(blk, c0 + c, kh, kw) for blk in (4, 5) for kh in range(3)]
        assert len(offs) * 1024 == len(self.prefixes) * self.stream_bytes, (len(offs), self.stream_bytes)
        return offs

    def ensure(self, stream, force=False):
        """Call after dp.ensure(): re-gathers when the packed arena was rewritten."""
        if self.ops is None:
            self._tab = torch.tensor(self._table(), dtype=torch.int64, device=self.arena.device)
            g = L.esr_frag_gather()
            g.src_off, g.src_base, g.dst, g.n = (self._tab.data_ptr(), self.dp.arena.data_ptr(),
                                                 self.arena.data_ptr(), self._tab.numel())
            self.ops = L.OpList()
            self.ops.add(L.OP_FRAG_GATHER, 'frag_gather', g)
        gen = getattr(self.dp, 'pack_count', 0)
        if force or gen != self._gen:
            self.ops.run(stream)
            self._gen = gen


def use_train_chain(dtype_e, B, H, W, explicit_z):
    """fp16 training passes run the dense blocks as fused chains — training forward (esr_rdb_chain.mode 1) and
    backward (esr_rdb_backward) — when the image fits the chain (all tiles co-resident) and z is the fused Philox
    stream (explicit z: per-conv launches).  ESR_RDB_TRAIN_CHAIN=0 restores the per-conv training plan."""
    if dtype_e != L.ESR_F16 or explicit_z or os.environ.get('ESR_RDB_TRAIN_CHAIN', '1') == '0':
        return False
    return rdb_chain_ok(B, H, W, False, False)


def rdb_chain_ok(B, H, W, noise, explicit_z):
    """The fused dense-block chain handles eval / fused-Philox passes whose images have at most
    esr_rdb_max_tiles_per_image() 16x32 tiles (all tiles of an image are co-resident, one per CU)."""
    if os.environ.get('ESR_RDB_FUSED', '1') == '0' or (noise and explicit_z):
        return False
    tpi = ((H + 15) // 16) * ((W + 31) // 32)
    return tpi <= L.lib().esr_rdb_max_tiles_per_image()


def use_rdb_bands(dtype_e, H, W):
    """Images with more tiles than CUs: the fused trunk in row bands for fp16 (339x510, nb=23: 9.2 ms against 10.3 ms
    of per-conv launches); fp32 keeps the per-conv launches, which are 6 % faster there (73.5 / 68.9 ms,
    tools/big_image_probe.py).  ESR_RDB_BANDS=1 forces bands for both, =0 turns them off."""
    mode = os.environ.get('ESR_RDB_BANDS', 'auto')
    if mode == '0' or os.environ.get('ESR_RDB_FUSED', '1') == '0' or rdb_band_geometry(H, W) is None:
        return False
    return mode == '1' or dtype_e == L.ESR_F16


def rdb_band_geometry(H, W):
    """Row bands for an image with more 16x32 tiles than CUs (include/esrgan_hip.h: esr_rdb_chain.band_rows):
    (band_rows, band_margin, n_bands), or None when not even a one-tile-row band fits (W > 32 * CUs / 3).
    One launch runs the three dense blocks of an RRDB, so a band recomputes 15 rows of each neighbour: margin 16
    (a whole tile row), bands as tall as the CU count allows and evened out over the image."""
    margin = 16
    tiles_x = (W + 31) // 32
    t = L.lib().esr_rdb_max_tiles_per_image() // tiles_x - 2 * (margin // 16)
    if t < 1:
        return None
    n = (H + 16 * t - 1) // (16 * t)
    rows = ((H + n - 1) // n + 15) // 16 * 16
    return rows, margin, n


def _conv(dtype_e, B, H, W, src, src_ch, dst, cw, act=L.ACT_NONE, ks=None, stride=1, upsample=0):
    c = L.esr_conv()
    c.dtype = dtype_e
    c.ks = cw.ks if ks is None else ks
    c.stride = stride
    c.upsample = upsample
    if upsample == 1 and getattr(cw, 'subpix', False):    # same function, 4-phase 2x2 form
        c.ks, c.upsample = 2, 3
    c.B, c.H, c.W = B, H, W
    cpg = 16 if dtype_e == L.ESR_F16 else 8
    c.cin_groups = (src_ch + cpg - 1) // cpg
    c.cout_blocks = (cw.cout + 31) // 32
    c.in_ = src
    if dst is not None:
        c.out = dst
    c.w = cw.w_ptr
    c.bias = cw.bias_ptr
    c.act = act
    c.alpha = 1.0
    c.beta = 1.0
    c.sigma = SIGMA
    c.noise_mode = L.NOISE_OFF
    c.layer1 = L.NO_LAYER
    c.layer2 = L.NO_LAYER
    c.layer3 = L.NO_LAYER
    c.gamma = 1.0
    c.mask_act = L.ACT_LRELU
    c.debug_flags = (_STORE_FLAVOUR << 3) | (int(os.environ.get('ESR_DBG', '0')) & ~7)
    return c


class Plan:
    """A recorded forward pass for one (module, input shape, dtype, mode)."""

    def __init__(self):
        self.ops = L.OpList()
        self.bufs = []
        self.in_op = None        # index of the NCHW->G32 layout op of the input
        self.out_op = None       # index of the op producing the NCHW output
        self.out_shape = None
        self.noise_ops = []      # (op index, which) of convs carrying a noise epilogue
        self.chain_ops = []      # indices of OP_RDB_CHAIN ops (noise mode / seed set per run)
        self.streams = None      # RdbStreams feeding the chain ops
        self.z_ops = []          # layout ops that import explicit z tensors, in noise-layer order
        self.wgen = None
        self.chain_noise = False

    def run(self, x, out, stream, seed=0, zs=None):
        arr = self.ops.array()
        arr[self.in_op].u.layout.nchw = x.data_ptr()
        o = arr[self.out_op]
        if o.kind == L.OP_CONV:
            o.u.conv.nchw_out = out.data_ptr()
        else:
            o.u.layout.nchw = out.data_ptr()
        mode = L.NOISE_OFF
        if not getattr(self, 'graph_bound', False):
            for i in self.chain_ops:
                ch = arr[i].u.rdb_chain
                ch.noise_mode = L.NOISE_PHILOX if self.chain_noise else L.NOISE_OFF
                ch.seed = seed
        if self.noise_ops:
            if zs is not None:
                assert len(zs) == len(self.z_ops), (len(zs), len(self.z_ops))
                for zi, z in zip(self.z_ops, zs):
                    arr[zi].u.layout.nchw = z.data_ptr()
                mode = L.NOISE_EXPLICIT
            else:
                mode = L.NOISE_PHILOX
            for i in self.noise_ops:
                arr[i].u.conv.noise_mode = mode
                arr[i].u.conv.seed = seed
        if self.streams is not None:
            self.streams.ensure(stream)
        if mode != L.NOISE_EXPLICIT and self.z_ops:
            # explicit-z import ops are recorded first; skip them when z is not supplied
            first = max(self.z_ops) + 1
            L.check(L.lib().esr_run_ops(C.cast(C.byref(arr, first * C.sizeof(L.esr_op)), C.c_void_p),
                                        len(self.ops.ops) - first, C.c_void_p(stream)), 'esr_run_ops')
        else:
            self.ops.run(stream)


class Builder:
    """Emits the fused-conv sequence of RDB / RRDB / RRDBNet into a Plan."""

    def __init__(self, wp, B, H, W, dtype, device, noise, variant):
        self.wp = wp
        self.B, self.H, self.W = B, H, W
        self.dt_e, self.tdtype, self.cpg = _dt(dtype)
        self.dtype = dtype
        self.device = device
        self.noise = noise            # bool: emit noise epilogues (training mode)
        self.variant = variant        # 'codes' | 'test_image'
        self.plan = Plan()
        self.n_noise = 0
        self.zbufs = []

    def buf(self, C_, H=None, W=None):
        b = G32(self.B, C_, H or self.H, W or self.W, self.dtype, self.device)
        self.plan.bufs.append(b)
        return b

    def import_nchw(self, dst, C_, affine=None):
        lo = L.esr_layout()
        lo.dtype, lo.to_g32 = self.dt_e, 1
        lo.B, lo.C, lo.H, lo.W = self.B, C_, dst.H, dst.W
        lo.g32 = dst.view(0, C_)
        return self.plan.ops.add(L.OP_LAYOUT, 'layout', lo)

    def export_nchw(self, src, C_):
        lo = L.esr_layout()
        lo.dtype, lo.to_g32 = self.dt_e, 0
        lo.B, lo.C, lo.H, lo.W = self.B, C_, src.H, src.W
        lo.g32 = src.view(0, C_)
        return self.plan.ops.add(L.OP_LAYOUT, 'layout', lo)

    def alloc_z(self, n):
        """Explicit-z staging: one 64-channel G32 buffer + import op per noise layer."""
        self.zbufs = []
        for _ in range(n):
            zb = self.buf(64)
            self.plan.z_ops.append(self.import_nchw(zb, 64))
            self.zbufs.append(zb)

    def _noise(self, c, which):
        """Attach noise layer (next id) to conv c as z1 (which=1) or z2 (which=2)."""
        lid = self.n_noise
        self.n_noise += 1
        zb = self.zbufs[lid] if lid < len(self.zbufs) else None
        if which == 1:
            c.layer1 = lid
            if zb is not None:
                c.z1 = zb.view(0, 64)
        else:
            c.layer2 = lid
            if zb is not None:
                c.z2 = zb.view(0, 64)

    def rdb(self, prefix, bf, bn, rrdb_x=None, rrdb_noise=False):
        """ResidualDenseBlock_5C (block.py:260-268) over concat buffer ``bf`` (x in ch 0..63);
        result -> ``bn`` channels 0..63.  ``rrdb_x``: fuse the RRDB tail (block.py:291)."""
        e = self.wp.entries
        B, H, W, d = self.B, self.H, self.W, self.dt_e
        add = self.plan.ops.add_conv
        # x1 = lrelu(conv1(x))
        add(_conv(d, B, H, W, bf.view(0), 64, bf.view(64, 32), e[prefix + '.conv1.0'], L.ACT_LRELU))
        # x2 = lrelu(conv2([x,x1])) + conv1x1(x)            (block.py:262-263)
        c = _conv(d, B, H, W, bf.view(0), 96, bf.view(96, 32), e[prefix + '.conv2.0'], L.ACT_LRELU)
        c.w1x1 = e[prefix + '.conv1x1'].w_ptr
        c.n1x1_groups = 64 // self.cpg
        add(c)
        # x3 = lrelu(conv3([x,x1,x2]))
        add(_conv(d, B, H, W, bf.view(0), 128, bf.view(128, 32), e[prefix + '.conv3.0'], L.ACT_LRELU))
        # x4 = lrelu(conv4([x..x3])) + x2                     (block.py:265-266)
        c = _conv(d, B, H, W, bf.view(0), 160, bf.view(160, 32), e[prefix + '.conv4.0'], L.ACT_LRELU)
        c.res1, c.alpha = bf.view(96, 32), 1.0
        add(c)
        # out = noise(conv5([x..x4]) * 0.2 + x)               (block.py:267-268)
        c = _conv(d, B, H, W, bf.view(0), 192, bn.view(0, 64), e[prefix + '.conv5.0'], L.ACT_NONE)
        c.res1, c.alpha = bf.view(0, 64), 0.2
        if self.noise:
            self._noise(c, 1)
        if rrdb_x is not None:                                 # RRDB: out*0.2 + x (block.py:291)
            c.res2, c.beta = rrdb_x.view(0, 64), 0.2
            if self.noise and rrdb_noise:                      # test_image/block.py:256
                self._noise(c, 2)
        i = add(c)
        if self.noise:
            self.plan.noise_ops.append(i)

    def rrdb(self, prefix, x0, x1, x2):
        """RRDB (block.py:287-291): x0 -> x1 -> x2 -> back into x0 (in place, pixel-local)."""
        self.rdb(prefix + '.RDB1', x0, x1)
        self.rdb(prefix + '.RDB2', x1, x2)
        self.rdb(prefix + '.RDB3', x2, x0, rrdb_x=x0, rrdb_noise=(self.variant == 'test_image'))

    def rdb_chain(self, specs):
        """ONE fused launch for a chain of dense blocks.  specs: list of (prefix, x_in, x_out, res2 or None,
        rrdb_noise) over 64-channel G32 buffers; the 128-channel dense scratch is shared."""
        P = self.plan
        if P.streams is None:
            P.streams = RdbStreams(self.wp, [s[0] for s in specs])
            base = 0
        else:
            base = len(P.streams.prefixes)
            raise NotImplementedError('one chain per plan')
        dense = self.buf(128)
        blocks = (L.esr_rdb_block * len(specs))()
        ent = self.wp.entries
        for i, (prefix, xi, xo, r2, rrdb_noise) in enumerate(specs):
            b = blocks[i]
            b.w = P.streams.w_ptr(base + i)
            b.bias = P.streams.bias_ptr(base + i)
            b.x_in, b.x_out = xi.view(0, 64), xo.view(0, 64)
            if r2 is not None:
                b.res2 = r2.view(0, 64)
            b.layer1 = b.layer2 = L.NO_LAYER
            # the output must be complete in memory when it is the chain's result or a later RRDB input
            later_res2 = any(s[3] is xo for s in specs[i + 1:])
            b.flags = L.RDB_FULL_OUT if (i == len(specs) - 1 or later_res2) else 0
            if self.noise:
                b.layer1 = self.n_noise
                self.n_noise += 1
                if r2 is not None and rrdb_noise:
                    b.layer2 = self.n_noise
                    self.n_noise += 1
        blk_t = torch.frombuffer(bytearray(bytes(blocks)), dtype=torch.uint8).to(self.device)
        ws_bytes = L.lib().esr_rdb_workspace_bytes(self.B, self.H, self.W)
        ws = torch.zeros((ws_bytes + 3) // 4, dtype=torch.int32, device=self.device)
        P.bufs.extend([blk_t, ws])
        ch = L.esr_rdb_chain()
        ch.dtype, ch.B, ch.H, ch.W = self.dt_e, self.B, self.H, self.W
        ch.n_blocks, ch.noise_mode, ch.sigma = len(specs), L.NOISE_OFF, SIGMA
        ch.dense = dense.view(0, 128)
        ch.blocks, ch.workspace, ch.workspace_bytes = blk_t.data_ptr(), ws.data_ptr(), ws_bytes
        i = P.ops.add(L.OP_RDB_CHAIN, 'rdb_chain', ch)
        P.chain_ops.append(i)
        P.chain_noise = self.noise
        P.chain_ws = ws
        return i

    def rdb_chain_banded(self, nb, geom, head):
        """The trunk of an image too large for one chain launch: per RRDB (and per image of the batch) one launch
        over row bands.  RRDB i reads the 64-channel buffer xs[i % 2] and writes xs[(i + 1) % 2] — out of place: a
        band's margin rows are its neighbours' own rows, so nothing a band reads may change during the launch —
        with the two intermediate block outputs and the dense scratch in band-local buffers.  head(view): emits
        the op(s) that fill xs[0]'s image rows.  Returns the view of the result's image rows."""
        P = self.plan
        S, m, nbands = geom
        hb = S + 2 * m
        B, H, W = self.B, self.H, self.W
        tall = [G32(B, 64, nbands * S + 2 * m, W, self.dtype, self.device) for _ in range(2)]
        mid = G32(nbands, 64, hb, W, self.dtype, self.device)
        dense = G32(nbands, 128, hb, W, self.dtype, self.device)
        P.bufs.extend(tall + [mid, dense])
        prefixes = ['model.1.sub.%d.RDB%d' % (i, j + 1) for i in range(nb) for j in range(3)]
        P.streams = RdbStreams(self.wp, prefixes)
        ws_bytes = L.lib().esr_rdb_workspace_bytes(nbands, hb, W)
        ws = torch.zeros((ws_bytes + 3) // 4, dtype=torch.int32, device=self.device)
        P.bufs.append(ws)
        head(tall[0].view(0, 64, m))

        def band(buf, bi):
            v = buf.view(0, 64)
            v.ptr += bi * buf.bs
            v.batch_stride = S * buf.Wp * 32
            return v
        for i in range(nb):
            src, dst = tall[i % 2], tall[(i + 1) % 2]
            for bi in range(B):
                blocks = (L.esr_rdb_block * 3)()
                for j in range(3):
                    b = blocks[j]
                    b.w, b.bias = P.streams.w_ptr(3 * i + j), P.streams.bias_ptr(3 * i + j)
                    b.x_in = band(src, bi) if j == 0 else mid.view(0, 64)
                    b.x_out = band(dst, bi) if j == 2 else mid.view(0, 64)
                    b.layer1 = b.layer2 = L.NO_LAYER
                    b.flags = 0
                blocks[2].res2 = band(src, bi)
                blocks[2].flags = L.RDB_FULL_OUT | L.RDB_BAND_OWN
                blk_t = torch.frombuffer(bytearray(bytes(blocks)), dtype=torch.uint8).to(self.device)
                P.bufs.append(blk_t)
                ch = L.esr_rdb_chain()
                ch.dtype, ch.B, ch.H, ch.W = self.dt_e, nbands, hb, W
                ch.n_blocks, ch.noise_mode, ch.sigma = 3, L.NOISE_OFF, SIGMA
                ch.dense = dense.view(0, 128)
                ch.blocks, ch.workspace, ch.workspace_bytes = blk_t.data_ptr(), ws.data_ptr(), ws_bytes
                ch.band_rows, ch.band_margin, ch.img_H = S, m, H
                P.chain_ops.append(P.ops.add(L.OP_RDB_CHAIN, 'rdb_chain', ch))
        P.chain_noise = False
        P.chain_ws = ws
        return tall[nb % 2].view(0, 64, m)

    def n_noise_layers(self, nb):
        return (4 if self.variant == 'test_image' else 3) * nb if self.noise else 0

    def rrdbnet(self, nb, in_nc, out_nc, explicit_z):
        """RRDBNet x4 (architecture.py:47-78): fea_conv, nb x RRDB, LR_conv + trunk shortcut,
        2 x (nearest x2 + conv + lrelu), HR_conv0 + lrelu, HR_conv1."""
        e = self.wp.entries
        B, H, W, d = self.B, self.H, self.W, self.dt_e
        P = self.plan
        self.zbufs = []
        if explicit_z and self.noise:
            self.alloc_z(self.n_noise_layers(nb))
        xin = self.buf(in_nc)
        fea = self.buf(64)
        P.in_op = self.import_nchw(xin, in_nc)
        if nb and rdb_chain_ok(B, H, W, self.noise, explicit_z):
            # fused trunk: two 64-channel slots + the chain's 128-channel dense scratch.  Per RRDB:
            # RDB1 xa -> xb, RDB2 xb -> xb (in place), RDB3 xb -> xa with the RRDB tail reading xa.
            xa, xb = self.buf(64), self.buf(64)
            c = _conv(d, B, H, W, xin.view(0), in_nc, xa.view(0, 64), e['model.0'])
            c.aux_out = fea.view(0, 64)
            P.ops.add_conv(c)
            specs = []
            for i in range(nb):
                pre = 'model.1.sub.%d' % i
                specs.append((pre + '.RDB1', xa, xb, None, False))
                specs.append((pre + '.RDB2', xb, xb, None, False))
                specs.append((pre + '.RDB3', xb, xa, xa, self.variant == 'test_image'))
            self.rdb_chain(specs)
            x0, x1 = xa, xb
        elif nb and not self.noise and use_rdb_bands(self.dt_e, H, W):
            # more tiles than CUs (a DIV2K-sized LR image): the fused trunk in row bands, one launch per RRDB
            def head(dst):
                c = _conv(d, B, H, W, xin.view(0), in_nc, dst, e['model.0'])
                c.aux_out = fea.view(0, 64)
                P.ops.add_conv(c)
            res, x1 = self.rdb_chain_banded(nb, rdb_band_geometry(H, W), head), self.buf(64)
            c = _conv(d, B, H, W, res, 64, x1.view(0, 64), e['model.1.sub.%d' % nb])
            c.res1, c.alpha = fea.view(0, 64), 1.0
            P.ops.add_conv(c)
            x0 = None
        else:
            x0, x1, x2 = self.buf(192), self.buf(192), self.buf(192)
            c = _conv(d, B, H, W, xin.view(0), in_nc, x0.view(0, 64), e['model.0'])
            c.aux_out = fea.view(0, 64)          # keep fea for the trunk shortcut (block.py:84-86)
            P.ops.add_conv(c)
            for i in range(nb):
                self.rrdb('model.1.sub.%d' % i, x0, x1, x2)
        if x0 is not None:
            c = _conv(d, B, H, W, x0.view(0), 64, x1.view(0, 64), e['model.1.sub.%d' % nb])
            c.res1, c.alpha = fea.view(0, 64), 1.0
            P.ops.add_conv(c)
        u1 = self.buf(64, 2 * H, 2 * W)
        P.ops.add_conv(_conv(d, B, 2 * H, 2 * W, x1.view(0), 64, u1.view(0, 64), e['model.3'],
                             L.ACT_LRELU, upsample=1))
        u2 = self.buf(64, 4 * H, 4 * W)
        P.ops.add_conv(_conv(d, B, 4 * H, 4 * W, u1.view(0), 64, u2.view(0, 64), e['model.6'],
                             L.ACT_LRELU, upsample=1))
        u3 = self.buf(64, 4 * H, 4 * W)
        P.ops.add_conv(_conv(d, B, 4 * H, 4 * W, u2.view(0), 64, u3.view(0, 64), e['model.8'],
                             L.ACT_LRELU))
        c = _conv(d, B, 4 * H, 4 * W, u3.view(0), 64, None, e['model.10'])
        c.nchw_out_c = out_nc
        P.out_op = P.ops.add_conv(c)
        P.out_shape = (B, out_nc, 4 * H, 4 * W)
        return P


def use_graphs():
    """hipGraph replay of the training plans (ESR_GRAPH=1).  Off by default: on ROCm 7.2 a graph launch
    of ~1 000 kernel nodes costs the host about as much as the individual launches (train step 25.6 ms
    with graphs vs 25.2 ms without, profiles/r01_experiments.md), so it buys nothing yet."""
    return os.environ.get('ESR_GRAPH', '0') == '1'


def deterministic_wgrad():
    """Two-stage weight-gradient reduction (esr_wgrad.partial): bit-identical gradients run to run instead of
    fp32 atomics (ESR_WGRAD_DET=0 restores the atomics)."""
    return os.environ.get('ESR_WGRAD_DET', '1') != '0'


def attach_wgrad_arena(oplist, device, exclusive=False):
    """Give every fp16 weight-gradient op of a backward list the partial arena of the deterministic reduction
    (one arena per list: a slot only lives inside one esr_run_ops launch group, and the list's wgrad runs are
    ordered on one stream).  exclusive: every op gets a region of its own (ESR_OPF_SIDE_FREE runs are in flight
    together).  Returns the arena tensor (keep it alive with the plan) or None."""
    if not deterministic_wgrad():
        return None
    wops = [o for o in oplist.ops if o.kind == L.OP_WGRAD]      # fp16 kernels and the fp32 parity kernel alike
    if not wops:
        return None
    if exclusive:
        arr = oplist.array()
        needs = []
        for i, o in enumerate(oplist.ops):
            if o.kind == L.OP_WGRAD:
                n = L.lib().esr_wgrad_workspace_elems(C.cast(C.byref(arr[i]), C.c_void_p), 1)
                needs.append((o, (int(n) + 63) // 64 * 64))
        total = sum(n for _, n in needs)
        if total <= 0:
            return None
        arena = torch.empty(total, dtype=torch.float32, device=device)
        off = 0
        for o, n in needs:
            o.u.wgrad.partial, o.u.wgrad.partial_elems = (arena.data_ptr() + 4 * off, n) if n else (None, 0)
            off += n
        oplist._arr = None
        return arena
    need = L.lib().esr_wgrad_workspace_elems(C.cast(oplist.array(), C.c_void_p), len(oplist.ops))
    if need <= 0:
        return None
    arena = torch.empty(need, dtype=torch.float32, device=device)
    for o in wops:
        o.u.wgrad.partial, o.u.wgrad.partial_elems = arena.data_ptr(), need
    oplist._arr = None
    return arena


def attach_free_wgrad_regions(oplist, device):
    """ESR_OPF_SIDE_FREE weight-gradient ops of a list whose other wgrad ops share one arena (attach_wgrad_arena): they
    are in flight together, so each gets a partial region of its own.  Returns the tensor that holds them (or None)."""
    if not deterministic_wgrad():
        return None
    arr = oplist.array()
    needs = []
    for i, o in enumerate(oplist.ops):
        if o.kind == L.OP_WGRAD and (o.flags & L.OPF_SIDE_FREE):
            n = L.lib().esr_wgrad_workspace_elems(C.cast(C.byref(arr[i]), C.c_void_p), 1)
            needs.append((o, (int(n) + 63) // 64 * 64))
    total = sum(n for _, n in needs)
    if total <= 0:
        return None
    arena = torch.empty(total, dtype=torch.float32, device=device)
    off = 0
    for o, n in needs:
        o.u.wgrad.partial, o.u.wgrad.partial_elems = (arena.data_ptr() + 4 * off, n) if n else (None, 0)
        off += n
    oplist._arr = None
    return arena


def current_stream():
    return torch.cuda.current_stream().cuda_stream


_concurrent_cache = {}


def concurrent_streams(device, n, probe=True):
    """n torch streams on `device` that really run NEXT TO the current stream and next to each other.  HIP maps streams
    onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4) and two streams on one queue serialise; measured on
    MI355X / ROCm 7.2 (tools/stream_probe.py): the FIRST stream a process creates shares its queue with the default
    stream — the train step's "side" stream of round 3, whose D step therefore never overlapped the main stream's
    work.  Probe: hold one workgroup on a (esr_debug_hold_cus, released as soon as the answer is known) and see whether
    a tiny kernel on b completes meanwhile, both directions, against the current stream and the streams already
    chosen.  Cached per (device, current stream).  probe=False: plain new streams."""
    import time
    dev = torch.device(device)
    cur = torch.cuda.current_stream(dev)
    key = (dev.index, cur.cuda_stream)
    have = _concurrent_cache.setdefault(key, [])
    if len(have) >= n:
        return have[:n]
    if not probe or torch.cuda.is_current_stream_capturing():
        while len(have) < n:
            have.append(torch.cuda.Stream(device=dev))
        return have[:n]
    words = torch.zeros(16, dtype=torch.int32).pin_memory()
    p_release, p_started = C.c_void_p(words.data_ptr()), C.c_void_p(words.data_ptr() + 4)
    x = torch.zeros(64, device=dev)

    def runs_next_to(a, b):
        # tiny kernel on b while one workgroup is held on a
        torch.cuda.synchronize(dev)
        words.zero_()
        L.check(L.lib().esr_debug_hold_cus(1, p_release, 20, p_started, C.c_void_p(a.cuda_stream)), 'esr_debug_hold_cus')
        t0 = time.perf_counter()
        with torch.cuda.stream(b):
            x.add_(1.0)
        b.synchronize()
        dt = time.perf_counter() - t0
        words[0] = 1
        torch.cuda.synchronize(dev)
        return dt < 0.008

    keep = []          # rejected candidates stay alive until the search ends (a freed stream's slot would be re-issued)
    for _ in range(24):
        if len(have) >= n:
            break
        cand = torch.cuda.Stream(device=dev)
        if all(runs_next_to(o, cand) and runs_next_to(cand, o) for o in [cur] + have):
            have.append(cand)
        else:
            keep.append(cand)
    while len(have) < n:                         # nothing better found: serialising streams still give correct results
        have.append(keep.pop() if keep else torch.cuda.Stream(device=dev))
    return have[:n]


class StreamOrder:
    """A module's launch plans own their activation buffers, packed weights and chain workspaces, so two inference
    calls of ONE module from two streams must not overlap (torch modules are stateless in that respect; a second
    stream is how users overlap independent work).  enter() makes the calling stream wait — on the device — for the
    module's previous call when that ran on another stream; leave() marks the end of this call.  Skipped under
    graph capture (the graph's order applies)."""

    def __init__(self):
        self.last = None
        self.event = None

    @staticmethod
    def of(mod):
        so = mod.__dict__.get('_stream_order')
        if so is None:
            so = mod.__dict__['_stream_order'] = StreamOrder()
        return so

    def enter(self):
        cur = torch.cuda.current_stream()
        if torch.cuda.is_current_stream_capturing():
            return cur
        if self.last is not None and self.last != cur.cuda_stream:
            cur.wait_event(self.event)
        return cur

    def leave(self, cur):
        if torch.cuda.is_current_stream_capturing():
            return
        if self.event is None:
            self.event = torch.cuda.Event()
        self.event.record(cur)
        self.last = cur.cuda_stream


def env_int(name, default, lo, hi):
    """An integer schedule knob from the environment, validated: a misspelt or out-of-range value is an error that
    names the knob and its range, not an opaque ValueError in the middle of a plan build or a silently ignored
    setting."""
    v = os.environ.get(name)
    if v is None or v == '':
        return default
    try:
        n = int(v)
    except ValueError:
        raise ValueError('%s=%r: expected an integer in [%d, %d]' % (name, v, lo, hi)) from None
    if not lo <= n <= hi:
        raise ValueError('%s=%d: expected an integer in [%d, %d]' % (name, n, lo, hi))
    return n


def bwd_chain_split(B, H, W, nb):
    """Launches the fused backward chain of a training plan is cut into (runs of whole RRDBs; 1 = one launch).  More
    than one only when the chain's grid leaves at least half of the CUs idle (4-row tiles, at most cus / 2 of them):
    the weight gradients of a run then execute under the next run's chain.  ESR_BWD_SPLIT = n forces n (1: off)."""
    env = env_int('ESR_BWD_SPLIT', None, 1, max(1, nb))
    if env is not None:
        return env
    cus = L.lib().esr_rdb_max_tiles_per_image()
    tiles4 = B * ((H + 3) // 4) * ((W + 31) // 32)
    return min(2, nb) if 2 * tiles4 <= cus else 1      # (train step, same box: 1 run 7.09 ms, 2 runs 7.01, 4 runs 7.06)


def bwd_follow():
    """Training crops: the backward chain's weight gradients as a follower pass next to ONE chain launch (ESR_OPF_FOLLOW)
    instead of the two-launch form of bwd_chain_split.  ESR_BWD_FOLLOW=0: the round-5 form (A/B)."""
    if os.environ.get('ESR_BWD_SPLIT'):            # an explicit split asks for the multi-launch form
        return False
    return os.environ.get('ESR_BWD_FOLLOW', '1') != '0'


def use_rdb_wgrad():
    """fp16 training plans: the six weight gradients of a dense block as ONE esr_rdb_wgrad pass over its saved
    concat buffer and gradient concat (csrc/rdb_wgrad.hip) instead of six esr_conv_wgrad problems
    (ESR_RDB_WGRAD=0 restores those, for A/B runs)."""
    return os.environ.get('ESR_RDB_WGRAD', '1') != '0'


def build_block_plan(kind, wp, B, H, W, dtype, device, noise, variant, explicit_z):
    """Stand-alone ResidualDenseBlock_5C ('rdb') or RRDB ('rrdb') pass: NCHW in -> NCHW out."""
    bld = Builder(wp, B, H, W, dtype, device, noise, variant)
    P = bld.plan
    n_noise = 0
    if noise:
        n_noise = 1 if kind == 'rdb' else (4 if variant == 'test_image' else 3)
    if explicit_z and noise:
        bld.alloc_z(n_noise)
    if rdb_chain_ok(B, H, W, noise, explicit_z):
        xa, xb = bld.buf(64), bld.buf(64)
        P.in_op = bld.import_nchw(xa, 64)
        if kind == 'rdb':
            bld.rdb_chain([('rdb', xa, xb, None, False)])
            P.out_op = bld.export_nchw(xb, 64)
        else:
            bld.rdb_chain([('rrdb.RDB1', xa, xb, None, False), ('rrdb.RDB2', xb, xb, None, False),
                           ('rrdb.RDB3', xb, xa, xa, variant == 'test_image')])
            P.out_op = bld.export_nchw(xa, 64)
        P.out_shape = (B, 64, H, W)
        return P
    x0, x1 = bld.buf(192), bld.buf(192)
    P.in_op = bld.import_nchw(x0, 64)
    if kind == 'rdb':
        bld.rdb('rdb', x0, x1)
        P.out_op = bld.export_nchw(x1, 64)
    else:
        x2 = bld.buf(192)
        bld.rrdb('rrdb', x0, x1, x2)
        P.out_op = bld.export_nchw(x0, 64)
    P.out_shape = (B, 64, H, W)
    return P


def build_rrdbnet_plan(wp, nb, in_nc, out_nc, B, H, W, dtype, device, noise, variant, explicit_z):
    bld = Builder(wp, B, H, W, dtype, device, noise, variant)
    return bld.rrdbnet(nb, in_nc, out_nc, explicit_z)


# =================================================================================================
# Training path: forward that keeps every activation + the matching backward launch list
# =================================================================================================
class DgradPack:
    """Packed operands of the input-gradient convolutions: Cin<->Cout transposed, taps rotated 180
    degrees (esr_pack.transpose_flip); conv5 of an RDB additionally folds the x4->x2 identity path
    (block.py:266) into its x2 output slice, and the upconvs get the 4x4/stride-2 adjoint kernel."""

    def __init__(self, convs, dtype, device, special, gathers=()):
        # convs: list of (key, weight_param); special: key -> dict(sum=(dst,src,count)) / dict(ups=True)
        # gathers: list of (key, dst_cout, [(weight_param, src_co0, scale), ...]) — gather-form operands
        # of a dense block (include/esrgan_hip.h: esr_pack.gather): K = the pieces' forward couts, in order
        self.esr_dtype, self.tdtype, self.cpg = _dt(dtype)
        self.convs, self.special, self.gathers = convs, special, list(gathers)
        self.entries = {}
        total, offs = 0, []
        for key, w in convs:
            cout, cin, ks, _ = w.shape
            ks_out = 4 if special.get(key, {}).get('ups') else ks
            offs.append(total)
            total += L.packed_weight_bytes(cin, cout, ks_out, self.esr_dtype)
        # a gather entry is (key, dst_cout, [(weight, src_co0, scale[, fold_co0]), ...]) or, for the transposed 1x1 of
        # the backward chain, (key, 'one_t', conv1x1.weight) — 4 KB of fragments (esr_pack.one_t; fp16 only)
        self.ones = [g for g in self.gathers if g[1] == 'one_t']
        self.gathers = [g for g in self.gathers if g[1] != 'one_t']
        if self.esr_dtype != L.ESR_F16:
            self.ones = []
        goffs = []
        for key, dst_cout, pieces in self.gathers:
            k_total = sum(pc[0].shape[0] for pc in pieces)
            goffs.append(total)
            total += L.packed_weight_bytes(dst_cout, k_total, 3, self.esr_dtype)
        ooffs = []
        for key, _, w in self.ones:
            ooffs.append(total)
            total += 4096
        self.arena = torch.zeros(total, dtype=torch.uint8, device=device)
        for (key, dst_cout, pieces), off in zip(self.gathers, goffs):
            e = ConvW()
            e.key, e.cout, e.cin, e.ks = key, dst_cout, sum(pc[0].shape[0] for pc in pieces), 3
            e.w_ptr, e.bias_ptr, e.has_bias = self.arena.data_ptr() + off, None, False
            self.entries[key] = e
        for (key, _, w), off in zip(self.ones, ooffs):
            e = ConvW()
            e.key, e.cout, e.cin, e.ks = key, 64, 32, 1
            e.w_ptr, e.bias_ptr, e.has_bias = self.arena.data_ptr() + off, None, False
            self.entries[key] = e
        for (key, w), off in zip(convs, offs):
            e = ConvW()
            e.key = key
            fc, fi, ks = w.shape[:3]
            e.cout, e.cin = fi, fc                  # the dgrad conv maps fwd-Cout -> fwd-Cin
            e.ks = 4 if special.get(key, {}).get('ups') else ks
            e.w_ptr, e.bias_ptr, e.has_bias = self.arena.data_ptr() + off, None, False
            self.entries[key] = e
        self._ptrs = None
        self.ops = None

    def ensure(self, stream, force=True, record_sig=False):
        """force=False (a frozen eval-mode net, the VGG feature extractor): re-pack only when a parameter's
        storage or version changed, like WeightPack.ensure."""
        fp = (len(self.convs), len(self.gathers), len(self.ones)) + tuple(
            lst[i][-1].data_ptr() if torch.is_tensor(lst[i][-1]) else lst[i][-1][0][0].data_ptr()
            for lst in (self.convs, self.gathers, self.ones) if lst for i in sorted({0, len(lst) // 2, len(lst) - 1}))
        ptrs = self._ptrs
        self._calls = getattr(self, '_calls', 0) + 1
        if (self._calls % WeightPack.FULL_CHECK_EVERY == 0
                or fp != getattr(self, '_fp', None)):      # (sampled storages first, as WeightPack.ensure)
            ptrs = tuple(w.data_ptr() for _, w in self.convs) + tuple(
                pc[0].data_ptr() for _, _, pieces in self.gathers for pc in pieces) + tuple(w.data_ptr() for _, _, w in self.ones)
            self._fp = fp
        if ptrs != self._ptrs:
            packs = []
            for key, dst_cout, pieces in self.gathers:
                e = self.entries[key]
                nchunks, chunk0 = e.cin // self.cpg, 0
                for pc in pieces:
                    w, src_co0, scale = pc[:3]
                    assert w.shape[0] % self.cpg == 0 and src_co0 + dst_cout <= w.shape[1]
                    pk = L.esr_pack()
                    pk.src, pk.dst = w.data_ptr(), e.w_ptr
                    pk.cout, pk.cin, pk.ks = w.shape[0], w.shape[1], 3
                    pk.dtype, pk.transpose_flip, pk.gather = self.esr_dtype, 1, 1
                    pk.dst_cout, pk.dst_chunk0, pk.dst_nchunks = dst_cout, chunk0, nchunks
                    pk.src_co0, pk.src_ks, pk.scale = src_co0, w.shape[2], scale
                    pk.fold_co0 = pc[3] if len(pc) > 3 else 0
                    packs.append(pk)
                    chunk0 += w.shape[0] // self.cpg
            for key, _, w in self.ones:
                pk = L.esr_pack()
                pk.src, pk.dst = w.data_ptr(), self.entries[key].w_ptr
                pk.cout, pk.cin, pk.ks, pk.dtype, pk.one_t, pk.scale = w.shape[0], w.shape[1], 1, self.esr_dtype, 1, 1.0
                packs.append(pk)
            for key, w in self.convs:
                e = self.entries[key]
                sp = self.special.get(key, {})
                pk = L.esr_pack()
                pk.src, pk.dst = w.data_ptr(), e.w_ptr
                pk.cout, pk.cin = w.shape[0], w.shape[1]
                pk.ks = e.ks
                pk.dtype = self.esr_dtype
                pk.transpose_flip = 2 if sp.get('ts2') else 1
                if 'sum' in sp:
                    pk.sum_dst, pk.sum_src, pk.sum_count = sp['sum']
                pk.ups_dgrad = 1 if sp.get('ups') else 0
                packs.append(pk)
            ops = L.OpList()
            bp, self._pack_keep = L.batch_pack_op(packs, self.arena.device)
            ops.add(L.OP_PACK_BATCH, 'pack_batch', bp)
            self.ops, self._ptrs = ops, ptrs
            self._sig = None
        sig = None if (force and not record_sig) else (tuple(w._version for _, w in self.convs)
                                  + tuple(pc[0]._version for _, _, pieces in self.gathers for pc in pieces)
                                  + tuple(w._version for _, _, w in self.ones))
        if force or sig != getattr(self, '_sig', None):
            self.ops.run(stream)  # (training nets: weights change every optimizer step, always re-pack)
            self._sig = sig
            self.pack_count = getattr(self, 'pack_count', 0) + 1


class TapMajorGrads:
    """fp16 wgrad accumulates 3x3 weight gradients tap-major ([tap][cout][cin]: atomics of one wave
    instruction land in 2 cache lines instead of ~36); one esr_grad_unpermute launch at the end of the
    backward pass rewrites all of them into the OIHW slots of the flat gradient buffer."""

    def __init__(self, grad_flat):
        self.flat = grad_flat
        self.tm = torch.zeros_like(grad_flat)
        self.rows = []
        self.total = 0

    def slot(self, off_elems, cout, cin, ntap=9):
        self.rows.append((off_elems, off_elems, self.total, cout, cin, ntap))
        self.total += cout * cin * ntap
        return self.tm.data_ptr() + 4 * off_elems

    def op(self, lo=None, hi=None):
        """The unpermute of every slot (default) or of the slots inside flat elements [lo, hi) — the segmented
        backward rewrites each finished span before its all-reduce starts."""
        rows = [r for r in self.rows if lo is None or lo <= r[0] < hi]
        if not rows:
            return None
        arr = (L.esr_unperm_entry * len(rows))()
        begin = pairs = 0
        for i, r in enumerate(rows):
            arr[i].src_off, arr[i].dst_off, arr[i].elem_begin = r[0], r[1], begin
            arr[i].cout, arr[i].cin, arr[i].ntap, arr[i].pair_begin = r[3], r[4], r[5], pairs
            begin += r[3] * r[4] * r[5]
            pairs += r[3] * r[4]
        raw = bytes(arr)
        table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.flat.device)
        self.tables = getattr(self, 'tables', []) + [table]
        up = L.esr_unpermute()
        up.table, up.n, up.total = table.data_ptr(), len(rows), begin
        up.n_pairs = pairs if pairs < 2 ** 31 else 0
        up.src, up.dst = self.tm.data_ptr(), self.flat.data_ptr()
        return up


class TrainPlan:
    """Forward (all activations kept) + backward launch lists of RRDBNet for one input shape."""

    def __init__(self):
        self.fwd = Plan()
        self.bwd = L.OpList()
        self.bufs = []
        self.busy = False
        self.gy_op = None            # layout op importing dL/dy (NCHW fp32) in the backward list
        self.bwd_noise_ops = []      # backward conv ops that need (noise_mode, seed)
        self.grad_flat = None        # fp32 flat gradient buffer; views per parameter
        self.grad_views = None
        self.tapmajor = None
        self.gx_op = None            # op exporting dL/dx (NCHW fp32): a layout op (stand-alone blocks) or fea_conv's dgrad
        self.gx_begin = None         # whole generator: ops [gx_begin, end) produce dL/dx and run only when it is wanted
        self.segments = None         # segmented backward: [(op_end, elem_lo, elem_hi)] — after ops [.., op_end) the
                                     # gradients in flat[elem_lo:elem_hi] are final (see build_rrdbnet_train_plan)
        self.graph = False           # hipGraph replay with I/O bound to the static tensors below
        self.follow_op = None        # index of the ESR_OPF_FOLLOW weight-gradient op of the backward list (follow_spare: the CUs its
        self.follow_spare = 0        # chain leaves free; follow_wgs: the plan's default grid)
        self.follow_wgs = 0
        self.bwd_chain_ops = []      # indices of OP_RDB_CHAIN_BWD ops in the backward list (noise mode / seed per run)
        self.bwd_streams = None      # RdbBwdStreams feeding them

    def enable_graph(self, in_shape, device):
        """Bind every per-step pointer / scalar of both launch lists to fixed device buffers so the
        lists can be captured once and replayed as hipGraphs (the train step is host-launch-bound):
        input / output / upstream-gradient tensors become static staging buffers, the Philox seed is
        read from `seed_t` on the device."""
        P = self.fwd
        assert not P.z_ops, 'graph replay is for the fused-Philox path (explicit z tensors re-bind pointers)'
        f32 = dict(dtype=torch.float32, device=device)
        self.x_static = torch.empty(in_shape, **f32)
        self.out_static = torch.empty(P.out_shape, **f32)
        self.gy_static = torch.empty(P.out_shape, **f32)
        self.gx_static = torch.empty(in_shape, **f32) if self.gx_op is not None else None
        self.seed_t = torch.zeros(1, dtype=torch.int64, device=device)
        arr = P.ops.array()
        arr[P.in_op].u.layout.nchw = self.x_static.data_ptr()
        o = arr[P.out_op]
        if o.kind == L.OP_CONV:
            o.u.conv.nchw_out = self.out_static.data_ptr()
        else:
            o.u.layout.nchw = self.out_static.data_ptr()
        for i in P.noise_ops:
            arr[i].u.conv.noise_mode = L.NOISE_PHILOX
            arr[i].u.conv.seed_dev = self.seed_t.data_ptr()
        for i in P.chain_ops:
            arr[i].u.rdb_chain.noise_mode = L.NOISE_PHILOX if P.chain_noise else L.NOISE_OFF
            arr[i].u.rdb_chain.seed_dev = self.seed_t.data_ptr()
        P.graph_bound = True         # Plan.run keeps its hands off the chain ops from here on
        barr = self.bwd.array()
        barr[self.gy_op].u.layout.nchw = self.gy_static.data_ptr()
        if self.gx_op is not None:
            if barr[self.gx_op].kind == L.OP_CONV:
                barr[self.gx_op].u.conv.nchw_out = self.gx_static.data_ptr()
            else:
                barr[self.gx_op].u.layout.nchw = self.gx_static.data_ptr()
        for i in self.bwd_noise_ops:
            barr[i].u.conv.noise_mode = L.NOISE_PHILOX
            barr[i].u.conv.seed_dev = self.seed_t.data_ptr()
        for i in self.bwd_chain_ops:
            barr[i].u.rdb_chain.noise_mode = L.NOISE_PHILOX if P.chain_noise else L.NOISE_OFF
            barr[i].u.rdb_chain.seed_dev = self.seed_t.data_ptr()
        self.graph = True


def build_rrdbnet_train_plan(net, wp, dp, nb, in_nc, out_nc, B, H, W, dtype, device, noise, variant,
                             explicit_z, kind='net', segmented=False):
    """RRDBNet forward keeping every RDB concat buffer (+ pre-residual activations of conv2/conv4,
    whose signs are the LeakyReLU masks) and the backward pass:
      * input gradients = the same fused conv kernel over transposed/rotated weights, with the
        LeakyReLU-mask / noise / residual-scale backward applied in its epilogue;
      * weight/bias gradients = esr_conv_wgrad.
    kind 'net' = the whole generator; 'rrdb' / 'rdb' = a stand-alone RRDB / ResidualDenseBlock_5C
    (64-channel NCHW in and out, gradient w.r.t. the input returned): same block code, no head/tail.
    segmented ('net' only): the backward list records, per RRDB (and for the tail / the first conv), the op
    index after which that slice of the flat gradient buffer is final — data-parallel runs start its
    all-reduce there, under the rest of the backward (TrainPlan.segments, functional._train_backward)."""
    dt_e, tdtype, cpg = _dt(dtype)
    block = kind != 'net'
    nj = 1 if kind == 'rdb' else 3                 # dense blocks per RRDB
    if block:
        nb = 1

    def pkey(i, j):
        if kind == 'rdb':
            return 'rdb'
        return ('rrdb.RDB%d' % (j + 1)) if kind == 'rrdb' else 'model.1.sub.%d.RDB%d' % (i, j + 1)
    TP = TrainPlan()
    P = TP.fwd
    e = wp.entries
    d = dt_e
    per = 4 if variant == 'test_image' else 3
    n_noise = ((1 if kind == 'rdb' else per * nb) if noise else 0)

    def buf(C_, h=H, w=W):
        b = G32(B, C_, h, w, dtype, device)
        TP.bufs.append(b)
        return b

    def imp(ops, dst, C_):
        lo = L.esr_layout()
        lo.dtype, lo.to_g32 = dt_e, 1
        lo.B, lo.C, lo.H, lo.W = B, C_, dst.H, dst.W
        lo.g32 = dst.view(0, C_)
        return ops.add(L.OP_LAYOUT, 'layout', lo)

    zb = []
    if noise and explicit_z:
        for _ in range(n_noise):
            z = buf(64)
            P.z_ops.append(imp(P.ops, z, 64))
            zb.append(z)

    def set_noise(c, slot, lid):
        """slot 1/2/3 of conv c multiplies by (1 + sigma * z[lid])."""
        if not noise or lid is None:
            return
        setattr(c, 'layer%d' % slot, lid)
        if zb:
            setattr(c, 'z%d' % slot, zb[lid].view(0, 64))

    # ------------------------------------------------------------------ forward
    chain = nb > 0 and use_train_chain(dt_e, B, H, W, explicit_z)
    S = [[buf(192) for _ in range(nj)] for _ in range(nb)]
    AUX = [[buf(64) for _ in range(nj)] for _ in range(nb)] if not chain else None
    XF = buf(64)                                   # output of the last RRDB
    if block:
        P.in_op = imp(P.ops, S[0][0], 64)          # x straight into the concat buffer's first slice
    else:
        xin, fea = buf(in_nc), buf(64)
        P.in_op = imp(P.ops, xin, in_nc)
        c = _conv(d, B, H, W, xin.view(0), in_nc, (S[0][0] if nb else XF).view(0, 64), e['model.0'])
        c.aux_out = fea.view(0, 64)
        P.ops.add_conv(c)
    MASKS = None
    if chain:
        # ---- fused training forward: ONE esr_rdb_forward launch (mode 1) over all dense blocks; every block keeps
        # its concat buffer S[i][j] = [x | x1..x4], its output (the next block's x) and its LeakyReLU masks
        order = [(i, j) for i in range(nb) for j in range(nj)]
        P.streams = RdbStreams(wp, [pkey(i, j) for i, j in order])
        mbytes = int(L.lib().esr_rdb_mask_bytes(B, H, W))
        MASKS = {ij: torch.empty(mbytes, dtype=torch.uint8, device=device) for ij in order}
        TP.bufs.append(MASKS)
        cblocks = (L.esr_rdb_block * len(order))()
        for n, (i, j) in enumerate(order):
            bf = S[i][j]
            bn = S[i][j + 1] if j < nj - 1 else (S[i + 1][0] if i + 1 < nb else XF)
            b = cblocks[n]
            b.w, b.bias = P.streams.w_ptr(n), P.streams.bias_ptr(n)
            b.x_in, b.x_out, b.dense = bf.view(0, 64), bn.view(0, 64), bf.view(64, 128)
            b.mask = MASKS[(i, j)].data_ptr()
            b.layer1 = (per * i + j) if noise else L.NO_LAYER
            b.layer2 = L.NO_LAYER
            if j == 2 and kind != 'rdb':
                b.res2 = S[i][0].view(0, 64)
                if noise and variant == 'test_image':
                    b.layer2 = per * i + 3
            b.flags = L.RDB_FULL_OUT
        cblk_t = torch.frombuffer(bytearray(bytes(cblocks)), dtype=torch.uint8).to(device)
        ws_bytes = L.lib().esr_rdb_workspace_bytes(B, H, W)
        cws = torch.zeros((ws_bytes + 3) // 4, dtype=torch.int32, device=device)
        TP.bufs.extend([cblk_t, cws])
        ch = L.esr_rdb_chain()
        ch.dtype, ch.B, ch.H, ch.W, ch.mode = dt_e, B, H, W, 1
        ch.n_blocks, ch.noise_mode, ch.sigma, ch.save_dense = len(order), L.NOISE_OFF, SIGMA, 0
        ch.dense = S[0][0].view(64, 128)                    # (geometry of every view; blocks carry their own)
        ch.blocks, ch.workspace, ch.workspace_bytes = cblk_t.data_ptr(), cws.data_ptr(), ws_bytes
        P.chain_ops.append(P.ops.add(L.OP_RDB_CHAIN, 'rdb_chain', ch))
        P.chain_noise = bool(noise)
        P.chain_ws = cws
    for i in range(nb if not chain else 0):
        for j in range(nj):
            bf, ax = S[i][j], AUX[i][j]
            bn = S[i][j + 1] if j < nj - 1 else (S[i + 1][0] if i + 1 < nb else XF)
            p = pkey(i, j)
            P.ops.add_conv(_conv(d, B, H, W, bf.view(0), 64, bf.view(64, 32), e[p + '.conv1.0'], L.ACT_LRELU))
            c = _conv(d, B, H, W, bf.view(0), 96, bf.view(96, 32), e[p + '.conv2.0'], L.ACT_LRELU)
            c.w1x1, c.n1x1_groups = e[p + '.conv1x1'].w_ptr, 64 // cpg
            c.aux_out = ax.view(0, 32)
            P.ops.add_conv(c)
            P.ops.add_conv(_conv(d, B, H, W, bf.view(0), 128, bf.view(128, 32), e[p + '.conv3.0'], L.ACT_LRELU))
            c = _conv(d, B, H, W, bf.view(0), 160, bf.view(160, 32), e[p + '.conv4.0'], L.ACT_LRELU)
            c.res1, c.alpha = bf.view(96, 32), 1.0
            c.aux_out = ax.view(32, 32)
            P.ops.add_conv(c)
            c = _conv(d, B, H, W, bf.view(0), 192, bn.view(0, 64), e[p + '.conv5.0'], L.ACT_NONE)
            c.res1, c.alpha = bf.view(0, 64), 0.2
            set_noise(c, 1, per * i + j)
            if j == 2 and kind != 'rdb':
                c.res2, c.beta = S[i][0].view(0, 64), 0.2
                if variant == 'test_image':
                    set_noise(c, 2, per * i + 3)
            k = P.ops.add_conv(c)
            if noise:
                P.noise_ops.append(k)
    if block:
        lo = L.esr_layout()
        lo.dtype, lo.to_g32 = dt_e, 0
        lo.B, lo.C, lo.H, lo.W = B, 64, H, W
        lo.g32 = XF.view(0, 64)
        P.out_op = P.ops.add(L.OP_LAYOUT, 'layout', lo)
        P.out_shape = (B, 64, H, W)
    T_ = buf(64) if not block else None
    if not block:
        c = _conv(d, B, H, W, XF.view(0), 64, T_.view(0, 64), e['model.1.sub.%d' % nb])
        c.res1, c.alpha = fea.view(0, 64), 1.0
        P.ops.add_conv(c)
        U1, U2, U3 = buf(64, 2 * H, 2 * W), buf(64, 4 * H, 4 * W), buf(64, 4 * H, 4 * W)
        P.ops.add_conv(_conv(d, B, 2 * H, 2 * W, T_.view(0), 64, U1.view(0, 64), e['model.3'], L.ACT_LRELU, upsample=1))
        P.ops.add_conv(_conv(d, B, 4 * H, 4 * W, U1.view(0), 64, U2.view(0, 64), e['model.6'], L.ACT_LRELU, upsample=1))
        P.ops.add_conv(_conv(d, B, 4 * H, 4 * W, U2.view(0), 64, U3.view(0, 64), e['model.8'], L.ACT_LRELU))
        c = _conv(d, B, 4 * H, 4 * W, U3.view(0), 64, None, e['model.10'])
        c.nchw_out_c = out_nc
        P.out_op = P.ops.add_conv(c)
        P.out_shape = (B, out_nc, 4 * H, 4 * W)

    # ------------------------------------------------------------------ gradient storage
    plist = [(k, w, b_) for k, w, b_ in net._conv_list()]
    sizes = []
    for k, w, b_ in plist:
        sizes.append(w.numel())
        if b_ is not None:
            sizes.append(b_.numel())
    TP.grad_flat = torch.zeros(sum(sizes), dtype=torch.float32, device=device)
    views, off, gptr, goff = [], 0, {}, {}
    TP.tapmajor = TapMajorGrads(TP.grad_flat) if dt_e == L.ESR_F16 else None
    for k, w, b_ in plist:
        gw = TP.grad_flat[off:off + w.numel()].view_as(w)
        goff[k] = off
        off += w.numel()
        gb = None
        if b_ is not None:
            gb = TP.grad_flat[off:off + b_.numel()].view_as(b_)
            off += b_.numel()
        views.append((gw, gb))
        gptr[k] = (gw.data_ptr(), gb.data_ptr() if gb is not None else None)
    TP.grad_views = views
    segmented = bool(segmented) and not block
    gend = {}
    for k, w, b_ in plist:
        gend[k] = goff[k] + w.numel() + (b_.numel() if b_ is not None else 0)
    segs = []

    def close_segment(prefixes):
        """Everything the ops so far produce for the parameters whose key starts with one of `prefixes` is
        final: rewrite the tap-major pieces of that span, record the boundary."""
        if not segmented:
            return
        ks_ = [k for k, _, _ in plist if any(k == p_ or k.startswith(p_ + '.') for p_ in prefixes)]
        lo, hi = min(goff[k] for k in ks_), max(gend[k] for k in ks_)
        assert hi - lo == sum(gend[k] - goff[k] for k in ks_), 'segment parameters must tile one span'
        if TP.tapmajor is not None:
            up = TP.tapmajor.op(lo, hi)
            if up is not None:
                Bk.add(L.OP_UNPERMUTE, 'unpermute', up)
        segs.append((len(Bk.ops), lo, hi))

    # ------------------------------------------------------------------ backward
    Bk = TP.bwd
    de = dp.entries

    def dconv(Ho, Wo, src, src_ch, dst, key, **kw):
        c = _conv(d, B, Ho, Wo, src, src_ch, dst, de[key], L.ACT_NONE, **kw)
        c.bias = None
        return c

    def add_b(c, noisy=False):
        k = Bk.add_conv(c)
        if noisy and noise:
            TP.bwd_noise_ops.append(k)
        return k

    deferred = None

    def wgrad(key, g, gin, Hh, Ww, cout, cin, ks=3, ups=0, scale=1.0):
        wg = L.esr_wgrad()
        wg.dtype, wg.ks, wg.stride, wg.upsample = dt_e, ks, 1, ups
        wg.B, wg.H, wg.W = B, Hh, Ww
        wg.cout, wg.cin = cout, cin
        wg.g, wg.in_ = g, gin
        wg.dw, wg.dbias = gptr[key]
        if TP.tapmajor is not None and ks == 3:
            wg.dw, wg.tap_major = TP.tapmajor.slot(goff[key], cout, cin), 1
        wg.scale = scale
        if deferred is not None:
            deferred.append(wg)      # emitted as one run at the end of the block (one batched launch)
        else:
            # head / tail convs: their gradient and input buffers are written once per backward pass and their slots of
            # the gradient buffer are read only behind the list's unpermute, so these launches run next to the main
            # chain WITHOUT ordering among themselves (ESR_OPF_SIDE_FREE: up to three in flight, each with its own
            # partial region — attach_free_wgrad_regions).  Round 5: as ordered side runs the main stream waited for
            # run k - 1 before forking run k — at training crops a 70 us weight gradient per 25 us dgrad conv, i.e.
            # the tail's backward took 0.4 ms of the step's critical path instead of 0.15
            Bk.add(L.OP_WGRAD, 'wgrad', wg, flags=(_SIDE | _SIDE_FREE) if _SIDE else 0)

    if block:
        GY = buf(64)
        TP.gy_op = imp(Bk, GY, 64)
        GTt = None
    else:
        GY = buf(out_nc, 4 * H, 4 * W)
        TP.gy_op = imp(Bk, GY, out_nc)
        GA8, GA6 = buf(64, 4 * H, 4 * W), buf(64, 4 * H, 4 * W)
        GA3 = buf(64, 2 * H, 2 * W)
        GTt = buf(64)                               # dL/dT (trunk output)
    gA = [buf(64), buf(64)]                         # RRDB skip gradient A(i), ping-pong
    # The six weight gradients of a block run on the side stream, concurrently with the NEXT block's
    # dgrad chain (both are latency-bound, ~64-workgroup launches at training sizes), and are joined
    # before the block after that starts: what they read (g_t, GA, G[96:128]) rotates so that the
    # chain running next to them writes other buffers — g_t through 3, G/GA through 2.
    # Gather-form dgrad (block._rdb_gathers): per block one 224-channel gradient concat
    #   Q = [g_t (64) | g_a4 | g_a3 | g_a2 | g_a1 | g_x2 raw]   (32 each)
    # that the slice convs read as a growing prefix and each fills one slice of — the dense
    # connectivity mirrored, no read-modify-write of an accumulator.  Q rotates through 3 buffers: the
    # block's weight gradients read it on the side stream while the next block runs, and the block
    # after that is the first to overwrite it (its predecessor already writes ITS g_t into slot 0).
    # The side runs are forked once per RRDB (its three blocks' 18 weight gradients in one run: every fork costs the
    # main stream an event record + wait, ~11 us of idle chip at LR sizes) and joined at the next fork, so a block's
    # Q has to survive two groups: 2 * nj + 1 buffers.
    NQ = 2 * nj + 1 if not chain else nb * nj       # fused backward: every block keeps its Q for the weight gradients
    Qs = [buf(224) for _ in range(NQ)]
    gT = [q for q in Qs]                            # g_t of a block = channels [0,64) of its Q
    X4 = buf(32) if not chain else None             # raw g_x4 (identity path x4 = lrelu(a4) + x2)
    GF = buf(64)                                    # dL/dfea

    if block:
        # Entry of the chain: the incoming gradient through the block's own tail.
        #   rdb : y = (0.2 x5 + x) n            -> g_t = g_y n
        #   rrdb: y = ((t3 n2) 0.2 + x) [n3']   -> skip gradient A = g_y [n3'],  g_t3 = 0.2 A n2
        # = a 1x1 identity "conv" (key '__eye') whose epilogue applies the noise / scale stages.
        c = dconv(H, W, GY.view(0), 64, None, '__eye')
        if kind == 'rdb':
            c.out = gT[0].view(0, 64)
            set_noise(c, 2, 0)
        else:
            c.out = gA[0].view(0, 64)
            if variant == 'test_image':
                set_noise(c, 2, 3)
            c.out3, c.gamma = gT[0].view(0, 64), 0.2
            set_noise(c, 3, 2)
        add_b(c, noisy=True)
    else:
        # HR_conv1 (model.10): u3 -> y
        wgrad('model.10', GY.view(0, out_nc), U3.view(0, 64), 4 * H, 4 * W, out_nc, 64)
        c = dconv(4 * H, 4 * W, GY.view(0), out_nc, None, 'model.10')
        c.mask, c.out2, c.mask_cb_begin = U3.view(0, 64), GA8.view(0, 64), 0
        add_b(c)
        # HR_conv0 (model.8): u2 -> u3 (lrelu)
        wgrad('model.8', GA8.view(0, 64), U2.view(0, 64), 4 * H, 4 * W, 64, 64)
        c = dconv(4 * H, 4 * W, GA8.view(0), 64, None, 'model.8')
        c.mask, c.out2 = U2.view(0, 64), GA6.view(0, 64)
        add_b(c)
        # upconv 2 (model.6): up(u1) -> u2 ; adjoint = 4x4/s2 conv
        wgrad('model.6', GA6.view(0, 64), U1.view(0, 64), 4 * H, 4 * W, 64, 64, ups=1)
        c = dconv(2 * H, 2 * W, GA6.view(0), 64, None, 'model.6', ks=4, stride=2)
        c.mask, c.out2 = U1.view(0, 64), GA3.view(0, 64)
        add_b(c)
        # upconv 1 (model.3): up(T) -> u1
        wgrad('model.3', GA3.view(0, 64), T_.view(0, 64), 2 * H, 2 * W, 64, 64, ups=1)
        add_b(dconv(H, W, GA3.view(0), 64, GTt.view(0, 64), 'model.3', ks=4, stride=2))
        # LR_conv (model.1.sub.nb): XF -> T - fea
        lrk = 'model.1.sub.%d' % nb
        wgrad(lrk, GTt.view(0, 64), XF.view(0, 64), H, W, 64, 64)
        c = dconv(H, W, GTt.view(0), 64, gA[0].view(0, 64) if nb else GF.view(0, 64), lrk)
        if nb:
            if variant == 'test_image':
                set_noise(c, 2, per * (nb - 1) + 3)
            c.out3, c.gamma = gT[0].view(0, 64), 0.2
            set_noise(c, 3, per * (nb - 1) + 2)
        else:
            c.res1 = GTt.view(0, 64)                    # fea feeds both the trunk and the shortcut
        add_b(c, noisy=bool(nb))
        close_segment(['model.1.sub.%d' % nb, 'model.3', 'model.6', 'model.8', 'model.10'])
    ca, ct = 0, 0
    fused_wgrad = dt_e == L.ESR_F16 and use_rdb_wgrad() and nb > 0
    if fused_wgrad and not chain:
        # per-task partial sums of the deterministic two-stage reduction: one arena, reused by every RRDB's pass
        # (the passes are ordered on the side stream; a slot only lives inside one pass)
        rdbw_arena = torch.empty(int(L.lib().esr_rdb_wgrad_workspace_elems(B, H, W, nj)), dtype=torch.float32, device=device)
        TP.bufs.append(rdbw_arena)
    GX = buf(64) if block else None                 # dL/dx of a stand-alone block
    if chain:
        # ---- fused backward: ONE esr_rdb_backward launch over the dense blocks in backward order (block n reads its
        # g_t from Qs[n][0:64], leaves g_a4..g_a1 and the raw g_x2 in Qs[n][64:224], and writes the next block's g_t),
        # then the weight gradients of all blocks in esr_rdb_wgrad passes over (S, Q)
        border = [(i, j) for i in range(nb - 1, -1, -1) for j in range(nj - 1, -1, -1)]
        TP.bwd_streams = RdbBwdStreams(dp, [pkey(i, j) for i, j in border])
        bblocks = (L.esr_rdb_block * len(border))()
        wblocks = []
        for n, (i, j) in enumerate(border):
            Q, bf, p = Qs[n], S[i][j], pkey(i, j)
            b = bblocks[n]
            b.w = TP.bwd_streams.w_ptr(n)
            b.x_in, b.dense, b.aux = Q.view(0, 64), Q.view(64, 128), Q.view(192, 32)
            b.mask = MASKS[(i, j)].data_ptr()
            b.layer1 = b.layer2 = L.NO_LAYER
            b.flags = L.RDB_FULL_OUT
            if j > 0:
                # g_x = g_y of RDB j - 1 -> its g_t = g_y * n
                b.x_out = Qs[n + 1].view(0, 64)
                if noise:
                    b.layer1 = per * i + j - 1
            elif kind == 'rdb':
                b.x_out = GX.view(0, 64)
            else:
                b.res2 = gA[ca].view(0, 64)
                if block:
                    b.x_out = GX.view(0, 64)
                elif i > 0:
                    b.out_a = gA[ca ^ 1].view(0, 64)
                    if noise and variant == 'test_image':
                        b.layer2 = per * (i - 1) + 3
                    b.x_out = Qs[n + 1].view(0, 64)
                    if noise:
                        b.layer1 = per * (i - 1) + 2
                    ca ^= 1
                else:
                    b.x_out = GF.view(0, 64)
            wb = L.esr_rdb_wgrad_block()
            wb.in_, wb.q = bf.view(0, 192), Q.view(0, 224)
            for k in range(5):
                key = p + '.conv%d.0' % (k + 1)
                wb.dw[k] = TP.tapmajor.slot(goff[key], 64 if k == 4 else 32, 64 + 32 * k)
                wb.db[k] = gptr[key][1]
            wb.dw[5] = gptr[p + '.conv1x1'][0]
            wblocks.append(wb)
        bblk_t = torch.frombuffer(bytearray(bytes(bblocks)), dtype=torch.uint8).to(device)
        ws_bytes = L.lib().esr_rdb_workspace_bytes(B, H, W)
        bws = torch.zeros((ws_bytes + 3) // 4, dtype=torch.int32, device=device)
        TP.bufs.extend([bblk_t, bws])
        def chain_op(k0, k1):
            ch = L.esr_rdb_chain()
            ch.dtype, ch.B, ch.H, ch.W, ch.mode = dt_e, B, H, W, 2
            ch.n_blocks, ch.noise_mode, ch.sigma, ch.save_dense = k1 - k0, L.NOISE_OFF, SIGMA, 1
            ch.dense = Qs[k0].view(64, 128)
            ch.blocks = bblk_t.data_ptr() + k0 * C.sizeof(L.esr_rdb_block)
            ch.workspace, ch.workspace_bytes = bws.data_ptr(), ws_bytes
            TP.bwd_chain_ops.append(Bk.add(L.OP_RDB_CHAIN_BWD, 'rdb_chain', ch))

        def wgrad_op(grp, arena, flags=0, max_wg=0):
            arr_ = (L.esr_rdb_wgrad_block * len(grp))(*grp)
            wt = torch.frombuffer(bytearray(bytes(arr_)), dtype=torch.uint8).to(device)
            TP.bufs.append(wt)
            rw = L.esr_rdb_wgrad()
            rw.dtype, rw.B, rw.H, rw.W = dt_e, B, H, W
            rw.n_blocks, rw.tap_major, rw.scale5, rw.scale = len(grp), 1, 0.2, 1.0
            rw.blocks = wt.data_ptr()
            rw.partial, rw.partial_elems = arena.data_ptr(), arena.numel()
            rw.max_workgroups = max_wg
            Bk.add(L.OP_RDB_WGRAD, 'rdb_wgrad', rw, flags=flags)

        TP.bwd_chain_ws = bws
        nsplit = 1 if segmented else bwd_chain_split(B, H, W, nb)
        cus_ = L.lib().esr_rdb_max_tiles_per_image()
        spare_ = cus_ - B * ((H + 3) // 4) * ((W + 31) // 32)           # CUs a 4-row-tile launch leaves free
        if nsplit > 1 and _SIDE and bwd_follow() and spare_ >= 32:
            # Round 6: ONE chain launch and, launched with it on the side stream, the weight gradients of ALL blocks as
            # a follower pass on the CUs the chain leaves free — a block's tasks start when the chain has published the
            # block (csrc/rdb_wgrad.hip: follow_wait), its partial sums are reduced by its last task.  Behind the chain
            # only the last block's tasks are left (the two-launch form below left the second run's pass + reduction:
            # 0.4 + 0.08 ms of the step's critical path).
            warena = torch.empty(int(L.lib().esr_rdb_wgrad_workspace_elems(B, H, W, len(border))), dtype=torch.float32, device=device)
            TP.bufs.append(warena)
            chain_op(0, len(border))
            # (the follower's workgroups each hold a whole CU's LDS for the length of the chain: what else runs next to
            # the G backward — the D step on the caller's second stream — needs CUs too.  Train step, same box, two runs
            # each: round-5 form 6.52 ms; follower on 40 / 48 / 56 / 64 / 72 / 80 / 88 / 96 / 128 workgroups 7.33 / 6.86 /
            # 6.52 / 6.28 / 6.29 / 6.26 / 6.27 / 6.23* / 6.57* (* another box: 6.42 without).  ESR_BWD_FOLLOW_WGS: A/B knob)
            wgrad_op(wblocks, warena, flags=_SIDE | L.OPF_FOLLOW, max_wg=env_int('ESR_BWD_FOLLOW_WGS', min(spare_, 80), 32, max(32, spare_)))
            # (functional._train_backward: a caller may size the follower per run — the train step's logging form)
            TP.follow_op, TP.follow_spare = len(Bk.ops) - 1, spare_
            TP.follow_wgs = Bk.ops[TP.follow_op].u.rdb_wgrad.max_workgroups
        elif nsplit > 1:
            # Small grids (the reference's training crops: 16 x 32^2 LR = 128 four-row tiles on 256 CUs): the chain leaves
            # half of the chip idle and the weight gradients — 0.7 ms behind a 1.9 ms chain — sit on the step's critical
            # path.  The chain runs as `nsplit` launches over runs of whole RRDBs, and the weight gradients of a run go
            # to the SIDE stream right behind its chain launch: they execute on the idle CUs under the next run's chain
            # (block n reads its g_t from Qs[n], which the previous launch's last block wrote: launch boundaries are
            # free of semantics).  Only the last run's weight gradients are left behind the chain.
            per_run = (nb + nsplit - 1) // nsplit
            # run boundaries (RRDB indices).  ESR_BWD_SPLIT_FIRST = n: two runs, the first of n RRDBs (A/B: the LAST run's
            # weight gradients are the ones left behind the chain, the first run's must fit under the second chain)
            first = env_int('ESR_BWD_SPLIT_FIRST', 0, 0, nb)          # 0 / nb: equal runs
            bounds = list(range(0, nb, per_run)) + [nb]
            if nsplit == 2 and 0 < first < nb:
                bounds = [0, first, nb]
            # (the arena of a pass is not monotonic in its block count: fewer blocks -> fewer images per task -> more slots)
            need = max(int(L.lib().esr_rdb_wgrad_workspace_elems(B, H, W, (b1 - b0) * nj)) for b0, b1 in zip(bounds[:-1], bounds[1:]))
            warena = torch.empty(need, dtype=torch.float32, device=device)
            TP.bufs.append(warena)
            cus = L.lib().esr_rdb_max_tiles_per_image()
            spare = max(32, cus - B * ((H + 3) // 4) * ((W + 31) // 32))     # CUs the chain's grid leaves free
            for r0, r1 in zip(bounds[:-1], bounds[1:]):
                k0, k1 = r0 * nj, r1 * nj
                chain_op(k0, k1)
                # every run but the last shares the chip with the next run's chain: its pass keeps to the spare CUs
                # (a persistent grid of that many workgroups) so that the chain's workgroups find theirs free
                wgrad_op(wblocks[k0:k1], warena, flags=_SIDE, max_wg=spare if k1 < len(border) else 0)
        else:
            chain_op(0, len(border))
            # weight gradients: one pass over all blocks — or, data-parallel, one per RRDB so that each RRDB's slice of the
            # flat gradient buffer goes to its all-reduce while the next pass runs
            groups = [wblocks] if not segmented else [wblocks[k:k + nj] for k in range(0, len(wblocks), nj)]
            n_max = max(len(g_) for g_ in groups)
            warena = torch.empty(int(L.lib().esr_rdb_wgrad_workspace_elems(B, H, W, n_max)), dtype=torch.float32, device=device)
            TP.bufs.append(warena)
            for gi, grp in enumerate(groups):
                wgrad_op(grp, warena)
                if segmented:
                    close_segment(['model.1.sub.%d' % (nb - 1 - gi)])
    for i in range(nb - 1 if not chain else -1, -1, -1):
        for j in range(nj - 1, -1, -1):
            bf, ax = S[i][j], AUX[i][j]
            p = pkey(i, j)
            Q = Qs[ct]                              # Q[0:64] already holds this block's g_t
            # the block's six weight gradients read Q and the saved input, intact until the group after next
            # starts -> emitted together with the rest of the RRDB's after its last dgrad chain (one side run)
            if deferred is None:
                deferred = []
            if fused_wgrad:
                # one esr_rdb_wgrad pass per RRDB over (saved concat buffer, Q) of its blocks (rdb_wgrad.hip)
                wb = L.esr_rdb_wgrad_block()
                wb.in_, wb.q = bf.view(0, 192), Q.view(0, 224)
                for k in range(5):
                    key = p + '.conv%d.0' % (k + 1)
                    wb.dw[k] = TP.tapmajor.slot(goff[key], 64 if k == 4 else 32, 64 + 32 * k)
                    wb.db[k] = gptr[key][1]
                wb.dw[5] = gptr[p + '.conv1x1'][0]
                deferred.append(wb)
            else:
                wgrad(p + '.conv5.0', Q.view(0, 64), bf.view(0, 192), H, W, 64, 192, scale=0.2)
                wgrad(p + '.conv4.0', Q.view(64, 32), bf.view(0, 160), H, W, 32, 160)
                wgrad(p + '.conv3.0', Q.view(96, 32), bf.view(0, 128), H, W, 32, 128)
                wgrad(p + '.conv2.0', Q.view(128, 32), bf.view(0, 96), H, W, 32, 96)
                wgrad(p + '.conv1x1', Q.view(192, 32), bf.view(0, 64), H, W, 32, 64, ks=1)
                wgrad(p + '.conv1.0', Q.view(160, 32), bf.view(0, 64), H, W, 32, 64)
            # slice x4: g_x4 = conv5^T[x4](0.2 g_t)  -> raw to X4, masked (lrelu'(a4)) to Q[64:96]
            c = dconv(H, W, Q.view(0), 64, X4.view(0, 32), p + '.g4')
            c.mask, c.out2, c.mask_cb_begin = ax.view(32, 32), Q.view(64, 32), 0
            add_b(c)
            # slice x3 -> g_a3 = masked into Q[96:128]
            c = dconv(H, W, Q.view(0), 96, None, p + '.g3')
            c.mask, c.out2, c.mask_cb_begin = bf.view(128, 32), Q.view(96, 32), 0
            add_b(c)
            # slice x2 (+ g_x4: x4 = lrelu(a4) + x2) -> raw to Q[192:224] (feeds the 1x1), masked to Q[128:160]
            c = dconv(H, W, Q.view(0), 128, Q.view(192, 32), p + '.g2')
            c.res1 = X4.view(0, 32)
            c.mask, c.out2, c.mask_cb_begin = ax.view(0, 32), Q.view(128, 32), 0
            add_b(c)
            # slice x1 -> g_a1 = masked into Q[160:192]
            c = dconv(H, W, Q.view(0), 160, None, p + '.g1')
            c.mask, c.out2, c.mask_cb_begin = bf.view(64, 32), Q.view(160, 32), 0
            add_b(c)
            # slice x closes the block: g_x = sum_k conv_k^T[x](g_ak) + conv1x1^T(g_x2) + g_t
            # (d(0.2 x5 + x)/dx)  (+ RRDB skip for RDB1)
            c = dconv(H, W, Q.view(0), 224, None, p + '.g0')
            c.res1 = Q.view(0, 64)
            if j > 0:
                # g_x = g_y of RDB j (previous in forward order) -> its g_t = g_y * n
                c.out = gT[(ct + 1) % NQ].view(0, 64)
                set_noise(c, 2, per * i + j - 1)
                ct = (ct + 1) % NQ
            elif kind == 'rdb':
                c.out = GX.view(0, 64)
            else:
                c.res2, c.beta = gA[ca].view(0, 64), 1.0
                if block:
                    c.out = GX.view(0, 64)
                elif i > 0:
                    c.out = gA[ca ^ 1].view(0, 64)
                    if variant == 'test_image':
                        set_noise(c, 2, per * (i - 1) + 3)
                    c.out3, c.gamma = gT[(ct + 1) % NQ].view(0, 64), 0.2
                    set_noise(c, 3, per * (i - 1) + 2)
                    ca ^= 1
                    ct = (ct + 1) % NQ
                else:
                    c.out = GF.view(0, 64)
            add_b(c, noisy=True)
        if fused_wgrad:
            arr = (L.esr_rdb_wgrad_block * len(deferred))(*deferred)
            blk_t = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
            TP.bufs.append(blk_t)
            rw = L.esr_rdb_wgrad()
            rw.dtype, rw.B, rw.H, rw.W = dt_e, B, H, W
            rw.n_blocks, rw.tap_major, rw.scale5, rw.scale = len(deferred), 1, 0.2, 1.0
            rw.blocks = blk_t.data_ptr()
            rw.partial, rw.partial_elems = rdbw_arena.data_ptr(), rdbw_arena.numel()
            Bk.add(L.OP_RDB_WGRAD, 'rdb_wgrad', rw, flags=_SIDE)
        else:
            for wg in deferred:
                Bk.add(L.OP_WGRAD, 'wgrad', wg, flags=_SIDE)
        deferred = None
        if not block:
            close_segment(['model.1.sub.%d' % i])
    if block:
        lo = L.esr_layout()
        lo.dtype, lo.to_g32 = dt_e, 0
        lo.B, lo.C, lo.H, lo.W = B, 64, H, W
        lo.g32 = GX.view(0, 64)
        TP.gx_op = None                             # bound after the unpermute op is appended
        gx_layout = lo
    else:
        if nb:
            # trunk shortcut (fea feeds T directly as well): dL/dfea = chain result + dL/dT
            GF2 = buf(64)
            c = dconv(H, W, GF.view(0), 64, GF2.view(0, 64), '__eye')
            c.res1 = GTt.view(0, 64)
            add_b(c)
            GF = GF2
        # fea_conv (model.0): weight gradient only (the LR input image needs no gradient)
        wgrad('model.0', GF.view(0, 64), xin.view(0, in_nc), H, W, 64, in_nc)
        close_segment(['model.0'])
    if TP.tapmajor is not None and not segmented:
        up = TP.tapmajor.op()
        if up is not None:
            Bk.add(L.OP_UNPERMUTE, 'unpermute', up)
    if segmented:
        assert sorted((lo, hi) for _, lo, hi in segs)[0][0] == 0 and sum(hi - lo for _, lo, hi in segs) == TP.grad_flat.numel()
        TP.segments = segs
    if block:
        TP.gx_op = Bk.add(L.OP_LAYOUT, 'layout', gx_layout)
    else:
        # dL/dx of the whole generator (autograd through architecture.py:76-78 when the LR input requires a gradient):
        # fea_conv's input gradient, written straight into the caller's NCHW tensor.  Recorded at the END of the list
        # and only run on request (TrainPlan.gx_begin: functional._train_backward stops there otherwise).
        TP.gx_begin = len(Bk.ops)
        c = dconv(H, W, GF.view(0), 64, None, 'model.0')
        c.nchw_out_c = in_nc
        TP.gx_op = add_b(c)
    TP.wgrad_arena = attach_wgrad_arena(Bk, device)
    TP.wgrad_free_arena = attach_free_wgrad_regions(Bk, device)
    return TP
