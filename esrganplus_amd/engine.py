"""Host-side launch planner for the HIP hot path.

Turns one pass of a drop-in module (``RRDBNet.forward`` — reference
codes/models/modules/architecture.py:76-78 — or a single ``ResidualDenseBlock_5C`` / ``RRDB``,
block.py:260-268 / 287-291) into a recorded list of fused-conv launches over G32 buffers
(include/esrgan_hip.h) that ONE C call replays on torch's current HIP stream.

torch is used here for device memory (buffers are torch tensors) and the stream handle only.
"""
import ctypes as C

import torch

from . import _lib as L

SIGMA = 0.1   # GaussianNoise sigma (block.py:111)


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

    def view(self, c0=0, nch=None):
        """esr_g32 view starting at channel c0 (must be group aligned)."""
        assert c0 % self.cpg == 0, (c0, self.cpg)
        g0 = c0 // self.cpg
        ng = self.ng - g0 if nch is None else (nch + self.cpg - 1) // self.cpg
        v = L.esr_g32()
        v.ptr = self.t.data_ptr() + g0 * self.gs
        v.batch_stride = self.bs
        v.group_stride = self.gs
        v.wp = self.Wp
        v.ngroups = ng
        return v

    def groups(self, nch):
        return (nch + self.cpg - 1) // self.cpg


class ConvW:
    """Packed weights of one conv (a slice of a WeightPack arena)."""
    __slots__ = ('key', 'cout', 'cin', 'ks', 'w_ptr', 'bias_ptr', 'has_bias')


class WeightPack:
    """MFMA-fragment-ordered copies of a module's conv weights (fp32 OIHW nn.Parameters stay the
    master copy — networks.py:30-44 pokes ``m.weight.data``, Adam updates them in place).
    ``ensure()`` re-packs (one launch per tensor, recorded once) whenever a parameter's storage
    or version changed, or unconditionally when ``force``."""

    def __init__(self, convs, dtype, device):
        # convs: list of (key, weight_param, bias_param_or_None)
        self.esr_dtype, self.tdtype, self.cpg = _dt(dtype)
        self.device = device
        self.convs = convs
        self.entries = {}
        total = 0
        offs = []
        for key, w, b in convs:
            cout, cin, ks, _ = w.shape
            offs.append(total)
            total += L.packed_weight_bytes(cout, cin, ks, self.esr_dtype)
        self.arena = torch.zeros(total, dtype=torch.uint8, device=device)
        nb_pad = sum(((w.shape[0] + 31) // 32) * 32 for _, w, b in convs if b is not None and w.shape[0] % 32)
        self.bias_arena = torch.zeros(max(nb_pad, 1), dtype=torch.float32, device=device)
        self._bias_copies = []
        bo = 0
        for (key, w, b), off in zip(convs, offs):
            e = ConvW()
            e.key, (e.cout, e.cin, e.ks) = key, w.shape[:3]
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
        ops = L.OpList()
        for key, w, b in self.convs:
            e = self.entries[key]
            pk = L.esr_pack()
            pk.src = w.data_ptr()
            pk.dst = e.w_ptr
            pk.cout, pk.cin, pk.ks = e.cout, e.cin, e.ks
            pk.dtype = self.esr_dtype
            pk.transpose_flip = 0
            ops.add(L.OP_PACK, 'pack', pk)
            if b is not None and e.cout % 32 == 0:
                e.bias_ptr = b.data_ptr()
        self.ops = ops
        self.generation += 1

    def ensure(self, stream, force=False):
        ptrs = tuple(p.data_ptr() for _, w, b in self.convs for p in (w, b) if p is not None)
        if ptrs != self._ptrs:
            self._rebuild()
            self._ptrs = ptrs
            self._sig = None
        sig = tuple(p._version for _, w, b in self.convs for p in (w, b) if p is not None)
        if force or sig != self._sig:
            self.ops.run(stream)
            with torch.no_grad():
                for src, dst in self._bias_copies:
                    dst.copy_(src)
            self._sig = sig


def _conv(dtype_e, B, H, W, src, src_ch, dst, cw, act=L.ACT_NONE, ks=None, stride=1, upsample=0):
    c = L.esr_conv()
    c.dtype = dtype_e
    c.ks = cw.ks if ks is None else ks
    c.stride = stride
    c.upsample = upsample
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
        self.z_ops = []          # layout ops that import explicit z tensors, in noise-layer order
        self.wgen = None

    def run(self, x, out, stream, seed=0, zs=None):
        arr = self.ops.array()
        arr[self.in_op].u.layout.nchw = x.data_ptr()
        o = arr[self.out_op]
        if o.kind == L.OP_CONV:
            o.u.conv.nchw_out = out.data_ptr()
        else:
            o.u.layout.nchw = out.data_ptr()
        mode = L.NOISE_OFF
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
        x0, x1, x2 = self.buf(192), self.buf(192), self.buf(192)
        P.in_op = self.import_nchw(xin, in_nc)
        c = _conv(d, B, H, W, xin.view(0), in_nc, x0.view(0, 64), e['model.0'])
        c.aux_out = fea.view(0, 64)          # keep fea for the trunk shortcut (block.py:84-86)
        P.ops.add_conv(c)
        for i in range(nb):
            self.rrdb('model.1.sub.%d' % i, x0, x1, x2)
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


def current_stream():
    return torch.cuda.current_stream().cuda_stream


def build_block_plan(kind, wp, B, H, W, dtype, device, noise, variant, explicit_z):
    """Stand-alone ResidualDenseBlock_5C ('rdb') or RRDB ('rrdb') pass: NCHW in -> NCHW out."""
    bld = Builder(wp, B, H, W, dtype, device, noise, variant)
    P = bld.plan
    n_noise = 0
    if noise:
        n_noise = 1 if kind == 'rdb' else (4 if variant == 'test_image' else 3)
    if explicit_z and noise:
        bld.alloc_z(n_noise)
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
