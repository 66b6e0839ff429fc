"""ctypes binding of libesrgan_hip.so (C ABI declared in include/esrgan_hip.h).

The product path has NO fallback: if the shared library is missing or a kernel launch fails this
module raises — it never routes through torch ops or the oracle.
"""
import ctypes as C
import os
import threading

# torch bundles its own libamdhip64.so.7; it MUST be the HIP runtime already resident when
# libesrgan_hip.so is dlopen'ed (same soname -> shared), otherwise our launches would go through a
# second runtime that does not see torch's device/stream state.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('ESR_LIB_PATH', os.path.join(_HERE, 'libesrgan_hip.so'))   # override: A/B builds

ESR_F16, ESR_F32 = 0, 1
ACT_NONE, ACT_LRELU, ACT_RELU = 0, 1, 2
NOISE_OFF, NOISE_PHILOX, NOISE_EXPLICIT = 0, 1, 2
OP_CONV, OP_PACK, OP_LAYOUT, OP_NOISE_FILL, OP_WGRAD, OP_BN, OP_POOL, OP_LINEAR, OP_UNPERMUTE, OP_PACK_BATCH = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10
OP_RDB_CHAIN, OP_FRAG_GATHER, OP_RDB_WGRAD, OP_RDB_CHAIN_BWD = 11, 12, 13, 14
BN_STATS, BN_FINALIZE, BN_APPLY, BN_BWD_REDUCE, BN_BWD_FINAL, BN_BWD_APPLY, BN_RESTAT, BN_FIN_APPLY = 0, 1, 2, 3, 4, 5, 6, 7
NO_LAYER = 0xFFFFFFFF
POOL_FWD, POOL_BWD, POOL_SHUFFLE, POOL_UNSHUFFLE = 0, 1, 2, 3      # esr_pool.mode


class esr_g32(C.Structure):
    _fields_ = [('ptr', C.c_void_p), ('batch_stride', C.c_int64), ('group_stride', C.c_int64),
                ('wp', C.c_int32), ('ngroups', C.c_int32)]


class esr_conv(C.Structure):
    _fields_ = [('dtype', C.c_int32), ('ks', C.c_int32), ('stride', C.c_int32),
                ('upsample', C.c_int32), ('B', C.c_int32), ('H', C.c_int32), ('W', C.c_int32),
                ('cin_groups', C.c_int32), ('cout_blocks', C.c_int32),
                ('in_', esr_g32), ('out', esr_g32),
                ('w', C.c_void_p), ('bias', C.c_void_p), ('w1x1', C.c_void_p),
                ('n1x1_groups', C.c_int32), ('act', C.c_int32),
                ('aux_out', esr_g32),
                ('alpha', C.c_float), ('res1', esr_g32),
                ('beta', C.c_float), ('res2', esr_g32),
                ('noise_mode', C.c_int32), ('sigma', C.c_float), ('seed', C.c_uint64),
                ('layer1', C.c_uint32), ('layer2', C.c_uint32),
                ('z1', esr_g32), ('z2', esr_g32), ('mask', esr_g32), ('out2', esr_g32),
                ('nchw_out_c', C.c_int32), ('nchw_out', C.c_void_p),
                ('debug_flags', C.c_int32), ('mask_cb_begin', C.c_int32), ('gamma', C.c_float),
                ('layer3', C.c_uint32), ('z3', esr_g32), ('out3', esr_g32),
                ('mask_act', C.c_int32), ('_pad2', C.c_int32), ('seed_dev', C.c_void_p),
                ('ksplit', C.c_int32), ('stat_groups', C.c_int32), ('stat_C', C.c_int32), ('_pad3', C.c_int32),
                ('split_ws', C.c_void_p), ('stat_sums', C.c_void_p)]


class esr_pack(C.Structure):
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('cout', C.c_int32), ('cin', C.c_int32),
                ('ks', C.c_int32), ('dtype', C.c_int32), ('transpose_flip', C.c_int32),
                ('sum_dst', C.c_int32), ('sum_src', C.c_int32), ('sum_count', C.c_int32),
                ('ups_dgrad', C.c_int32),
                ('gather', C.c_int32), ('dst_cout', C.c_int32), ('dst_chunk0', C.c_int32),
                ('dst_nchunks', C.c_int32), ('src_co0', C.c_int32), ('src_ks', C.c_int32),
                ('scale', C.c_float), ('ups_fwd', C.c_int32), ('fold_co0', C.c_int32), ('one_t', C.c_int32)]


class esr_wgrad(C.Structure):
    _fields_ = [('dtype', C.c_int32), ('ks', C.c_int32), ('stride', C.c_int32),
                ('upsample', C.c_int32), ('B', C.c_int32), ('H', C.c_int32), ('W', C.c_int32),
                ('cout', C.c_int32), ('cin', C.c_int32), ('g', esr_g32), ('in_', esr_g32),
                ('dw', C.c_void_p), ('dbias', C.c_void_p), ('scale', C.c_float), ('tap_major', C.c_int32),
                ('partial', C.c_void_p), ('partial_elems', C.c_int64)]


class esr_layout(C.Structure):
    _fields_ = [('dtype', C.c_int32), ('to_g32', C.c_int32), ('B', C.c_int32), ('C', C.c_int32),
                ('H', C.c_int32), ('W', C.c_int32), ('nchw', C.c_void_p), ('g32', esr_g32),
                ('use_affine', C.c_int32), ('mean_c', C.c_float * 4), ('inv_std_c', C.c_float * 4),
                ('accumulate', C.c_int32), ('_pad', C.c_int32)]


class esr_noise_fill(C.Structure):
    _fields_ = [('dst', C.c_void_p), ('B', C.c_int32), ('C', C.c_int32), ('H', C.c_int32),
                ('W', C.c_int32), ('seed', C.c_uint64), ('layer', C.c_uint32)]


class esr_bn(C.Structure):
    _fields_ = [('dtype', C.c_int32), ('mode', C.c_int32), ('B', C.c_int32), ('C', C.c_int32),
                ('H', C.c_int32), ('W', C.c_int32), ('training', C.c_int32), ('act', C.c_int32),
                ('momentum', C.c_float), ('eps', C.c_float),
                ('x', esr_g32), ('y', esr_g32), ('g', esr_g32), ('gx', esr_g32),
                ('sums', C.c_void_p), ('mean', C.c_void_p), ('invstd', C.c_void_p),
                ('gamma', C.c_void_p), ('beta', C.c_void_p), ('running_mean', C.c_void_p),
                ('running_var', C.c_void_p), ('dgamma', C.c_void_p), ('dbeta', C.c_void_p),
                ('groups', C.c_int32), ('_pad', C.c_int32), ('num_batches_tracked', C.c_void_p)]


class esr_pool(C.Structure):
    _fields_ = [('dtype', C.c_int32), ('mode', C.c_int32), ('B', C.c_int32), ('C', C.c_int32),
                ('H', C.c_int32), ('W', C.c_int32), ('relu_mask', C.c_int32), ('_pad', C.c_int32),
                ('x', esr_g32), ('y', esr_g32), ('g', esr_g32), ('gx', esr_g32)]


class esr_linear(C.Structure):
    _fields_ = [('mode', C.c_int32), ('B', C.c_int32), ('I', C.c_int32), ('O', C.c_int32),
                ('act', C.c_int32), ('in_act', C.c_int32),
                ('x', C.c_void_p), ('w', C.c_void_p), ('b', C.c_void_p), ('y', C.c_void_p),
                ('g', C.c_void_p), ('ysaved', C.c_void_p), ('gx', C.c_void_p), ('dw', C.c_void_p),
                ('db', C.c_void_p)]


class esr_unperm_entry(C.Structure):
    _fields_ = [('src_off', C.c_int64), ('dst_off', C.c_int64), ('elem_begin', C.c_int64),
                ('cout', C.c_int32), ('cin', C.c_int32), ('ntap', C.c_int32), ('pair_begin', C.c_int32)]


class esr_unpermute(C.Structure):
    _fields_ = [('table', C.c_void_p), ('n', C.c_int32), ('n_pairs', C.c_int32), ('total', C.c_int64),
                ('src', C.c_void_p), ('dst', C.c_void_p)]


ADAM_BLOCK_ELEMS = 4096


class esr_adam_entry(C.Structure):     # rows of esr_adam.entries (optim.py builds them as an int64[n, 3] array)
    _fields_ = [('p', C.c_void_p), ('goff', C.c_int64), ('n', C.c_int64)]


class esr_adam_block(C.Structure):     # rows of esr_adam.blocks (int32[n, 2])
    _fields_ = [('entry', C.c_int32), ('first', C.c_int32)]


class esr_adam(C.Structure):
    _fields_ = [('entries', C.c_void_p), ('blocks', C.c_void_p), ('nblocks', C.c_int32), ('_pad', C.c_int32),
                ('grad', C.c_void_p), ('exp_avg', C.c_void_p), ('exp_avg_sq', C.c_void_p),
                ('lr', C.c_float), ('beta1', C.c_float), ('beta2', C.c_float), ('eps', C.c_float),
                ('bc1', C.c_float), ('bc2', C.c_float), ('grad_scale', C.c_float), ('weight_decay', C.c_float),
                ('amp_state', C.c_void_p), ('amp_slot', C.c_int32), ('_pad2', C.c_int32), ('step_count', C.c_void_p),
                ('beta1_d', C.c_double), ('beta2_d', C.c_double)]


AMP_CHECK, AMP_UPDATE, AMP_COUNT = 0, 1, 2


class esr_amp(C.Structure):
    _fields_ = [('mode', C.c_int32), ('interval', C.c_int32), ('state', C.c_void_p), ('grad', C.c_void_p),
                ('n', C.c_int64), ('growth', C.c_float), ('backoff', C.c_float),
                ('slot', C.c_int32), ('_pad', C.c_int32), ('step_count', C.c_void_p)]


class esr_resample(C.Structure):
    _fields_ = [('in_', C.c_void_p), ('out', C.c_void_p), ('planes', C.c_int32), ('in_h', C.c_int32),
                ('in_w', C.c_int32), ('out_len', C.c_int32), ('axis', C.c_int32), ('taps', C.c_int32),
                ('w', C.c_void_p), ('idx', C.c_void_p)]


class esr_pack_batch(C.Structure):
    _fields_ = [('table', C.c_void_p), ('piece_begin', C.c_void_p), ('n', C.c_int32), ('_pad', C.c_int32),
                ('total_pieces', C.c_int64)]


class esr_rdb_block(C.Structure):
    _fields_ = [('w', C.c_void_p), ('bias', C.c_void_p), ('x_in', esr_g32), ('x_out', esr_g32),
                ('res2', esr_g32), ('layer1', C.c_uint32), ('layer2', C.c_uint32), ('flags', C.c_uint32),
                ('_pad', C.c_uint32), ('dense', esr_g32), ('mask', C.c_void_p), ('aux', esr_g32), ('out_a', esr_g32)]


RDB_FULL_OUT = 1
RDB_BAND_OWN = 2


class esr_rdb_chain(C.Structure):
    _fields_ = [('dtype', C.c_int32), ('B', C.c_int32), ('H', C.c_int32), ('W', C.c_int32),
                ('n_blocks', C.c_int32), ('noise_mode', C.c_int32), ('sigma', C.c_float), ('save_dense', C.c_int32),
                ('seed', C.c_uint64), ('seed_dev', C.c_void_p), ('dense', esr_g32), ('blocks', C.c_void_p),
                ('workspace', C.c_void_p), ('workspace_bytes', C.c_size_t), ('trace', C.c_void_p),
                ('mode', C.c_int32), ('_pad2', C.c_int32),
                ('band_rows', C.c_int32), ('band_margin', C.c_int32), ('img_H', C.c_int32), ('_pad3', C.c_int32)]


class esr_l1_loss(C.Structure):
    _fields_ = [('a', C.c_void_p), ('b', C.c_void_p), ('grad_a', C.c_void_p), ('loss', C.c_void_p),
                ('scratch', C.c_void_p), ('n', C.c_int64), ('weight', C.c_float), ('grad_scale', C.c_float),
                ('grad_scale_dev', C.c_void_p)]


class esr_ragan_loss(C.Structure):
    _fields_ = [('x', C.c_void_p), ('y', C.c_void_p), ('grad_x', C.c_void_p), ('grad_y', C.c_void_p),
                ('loss', C.c_void_p), ('mean_x', C.c_void_p), ('mean_y', C.c_void_p),
                ('bce_x', C.c_void_p), ('bce_y', C.c_void_p), ('n', C.c_int32),
                ('tx', C.c_float), ('ty', C.c_float), ('weight', C.c_float),
                ('mode', C.c_int32), ('grad_scale', C.c_float), ('sums', C.c_void_p), ('ext', C.c_void_p),
                ('grad_scale_dev', C.c_void_p)]


class esr_img_metrics(C.Structure):
    _fields_ = [('sr', C.c_void_p), ('hr', C.c_void_p), ('C', C.c_int32), ('H', C.c_int32), ('W', C.c_int32),
                ('crop', C.c_int32), ('y_only', C.c_int32), ('_pad', C.c_int32), ('lo', C.c_float), ('hi', C.c_float),
                ('img_sr', C.c_void_p), ('img_hr', C.c_void_p), ('y_sr', C.c_void_p), ('y_hr', C.c_void_p),
                ('out', C.c_void_p), ('win', C.c_double * 11)]


class esr_frag_gather(C.Structure):
    _fields_ = [('src_off', C.c_void_p), ('src_base', C.c_void_p), ('dst', C.c_void_p), ('n', C.c_int64),
                ('piece_bytes', C.c_int32), ('_pad', C.c_int32)]


class esr_rdb_wgrad_block(C.Structure):
    _fields_ = [('in_', esr_g32), ('q', esr_g32), ('dw', C.c_void_p * 6), ('db', C.c_void_p * 6)]


class esr_rdb_wgrad(C.Structure):
    _fields_ = [('dtype', C.c_int32), ('B', C.c_int32), ('H', C.c_int32), ('W', C.c_int32),
                ('n_blocks', C.c_int32), ('tap_major', C.c_int32), ('scale5', C.c_float), ('scale', C.c_float),
                ('blocks', C.c_void_p), ('partial', C.c_void_p), ('partial_elems', C.c_int64),
                ('max_workgroups', C.c_int32), ('_pad', C.c_int32)]


class _op_union(C.Union):
    _fields_ = [('conv', esr_conv), ('pack', esr_pack), ('layout', esr_layout),
                ('noise_fill', esr_noise_fill), ('wgrad', esr_wgrad), ('bn', esr_bn),
                ('pool', esr_pool), ('linear', esr_linear), ('unpermute', esr_unpermute),
                ('pack_batch', esr_pack_batch), ('rdb_chain', esr_rdb_chain), ('frag_gather', esr_frag_gather),
                ('rdb_wgrad', esr_rdb_wgrad)]


class esr_op(C.Structure):
    _fields_ = [('kind', C.c_int32), ('flags', C.c_int32), ('u', _op_union)]


# every symbol include/esrgan_hip.h declares (tests check the .so exports all of them)
EXPORTS = ['esr_packed_weight_bytes', 'esr_g32_dims', 'esr_conv_forward', 'esr_pack_conv_weights',
           'esr_convert_layout', 'esr_fill_noise', 'esr_conv_wgrad', 'esr_conv_wgrad_multi', 'esr_batchnorm', 'esr_maxpool2',
           'esr_linear_op', 'esr_grad_unpermute', 'esr_adam_step', 'esr_amp_step', 'esr_resample_axis', 'esr_pack_conv_weights_batch', 'esr_pack_pieces',
           'esr_run_ops', 'esr_run_ops_timed', 'esr_graph_create', 'esr_graph_launch', 'esr_graph_destroy', 'esr_last_error',
           'esr_abi_version', 'esr_sizeof_op', 'esr_rdb_forward', 'esr_rdb_workspace_bytes', 'esr_rdb_weight_stream_bytes',
           'esr_rdb_max_tiles_per_image', 'esr_gather_fragments', 'esr_image_metrics', 'esr_wgrad_workspace_elems',
           'esr_l1_loss_forward', 'esr_ragan_loss_forward', 'esr_rdb_wgrad_run', 'esr_rdb_wgrad_workspace_elems', 'esr_rdb_backward',
           'esr_rdb_mask_bytes', 'esr_rdb_check_abort', 'esr_debug_hold_cus', 'esr_debug_device_alias', 'esr_debug_chain_order_waits',
           'esr_debug_mfma_probe', 'esr_debug_rdb_wgrad_follow']

_lib = None
_lock = threading.Lock()


class HipExtensionError(RuntimeError):
    pass


def lib():
    """Load libesrgan_hip.so once; raise loudly if it is missing or does not match this binding."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise HipExtensionError(
                'esrganplus_amd: %s not found — build it with `python -c "import __graft_entry__ as g; '
                'g.build()"` (hipcc --offload-arch=gfx950). There is no CPU/eager fallback.' % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name in EXPORTS:
            if not hasattr(L, name):
                raise HipExtensionError('libesrgan_hip.so lacks symbol ' + name)
        L.esr_last_error.restype = C.c_char_p
        L.esr_sizeof_op.restype = C.c_size_t
        L.esr_pack_pieces.restype = C.c_int64
        L.esr_pack_pieces.argtypes = [C.POINTER(esr_pack)]
        L.esr_packed_weight_bytes.restype = C.c_size_t
        L.esr_packed_weight_bytes.argtypes = [C.c_int32] * 4
        L.esr_g32_dims.restype = None
        L.esr_g32_dims.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.esr_run_ops.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.esr_run_ops_timed.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.esr_graph_create.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
        L.esr_graph_launch.argtypes = [C.c_void_p, C.c_void_p]
        L.esr_graph_destroy.argtypes = [C.c_void_p]
        L.esr_conv_wgrad_multi.argtypes = [C.POINTER(esr_wgrad), C.c_int32, C.c_void_p]
        L.esr_wgrad_workspace_elems.restype = C.c_int64
        L.esr_wgrad_workspace_elems.argtypes = [C.c_void_p, C.c_int32]
        L.esr_rdb_workspace_bytes.restype = C.c_size_t
        L.esr_rdb_workspace_bytes.argtypes = [C.c_int32] * 3
        L.esr_rdb_weight_stream_bytes.restype = C.c_size_t
        L.esr_rdb_weight_stream_bytes.argtypes = [C.c_int32]
        L.esr_rdb_mask_bytes.restype = C.c_size_t
        L.esr_rdb_mask_bytes.argtypes = [C.c_int32] * 3
        L.esr_debug_chain_order_waits.restype = C.c_uint64
        L.esr_debug_device_alias.argtypes = [C.c_int32]
        L.esr_rdb_wgrad_workspace_elems.restype = C.c_int64
        L.esr_rdb_wgrad_workspace_elems.argtypes = [C.c_int32] * 4
        for name, st in (('esr_conv_forward', esr_conv), ('esr_pack_conv_weights', esr_pack),
                         ('esr_convert_layout', esr_layout), ('esr_fill_noise', esr_noise_fill),
                         ('esr_conv_wgrad', esr_wgrad), ('esr_batchnorm', esr_bn),
                         ('esr_maxpool2', esr_pool), ('esr_linear_op', esr_linear),
                         ('esr_grad_unpermute', esr_unpermute), ('esr_adam_step', esr_adam), ('esr_amp_step', esr_amp), ('esr_resample_axis', esr_resample),
                         ('esr_pack_conv_weights_batch', esr_pack_batch), ('esr_rdb_forward', esr_rdb_chain),
                         ('esr_gather_fragments', esr_frag_gather), ('esr_image_metrics', esr_img_metrics),
                         ('esr_l1_loss_forward', esr_l1_loss), ('esr_ragan_loss_forward', esr_ragan_loss),
                         ('esr_rdb_wgrad_run', esr_rdb_wgrad), ('esr_rdb_backward', esr_rdb_chain)):
            getattr(L, name).argtypes = [C.POINTER(st), C.c_void_p]
        if L.esr_sizeof_op() != C.sizeof(esr_op):
            raise HipExtensionError('ABI mismatch: sizeof(esr_op) C=%d ctypes=%d'
                                    % (L.esr_sizeof_op(), C.sizeof(esr_op)))
        _lib = L
    return _lib


def check(rc, what=''):
    if rc != 0:
        raise HipExtensionError('%s failed (%d): %s' % (what or 'libesrgan_hip', rc,
                                                        lib().esr_last_error().decode()))


def g32_dims(h, w):
    hp, wp = C.c_int32(), C.c_int32()
    lib().esr_g32_dims(h, w, C.byref(hp), C.byref(wp))
    return hp.value, wp.value


def packed_weight_bytes(cout, cin, ks, dtype):
    return lib().esr_packed_weight_bytes(cout, cin, ks, dtype)


OPF_SIDE = 1   # esr_op.flags: run of wgrad ops on the library's side stream (include/esrgan_hip.h)
OPF_FOLLOW = 4   # on an OP_RDB_WGRAD op right behind its OP_RDB_CHAIN_BWD op: launched with the chain, one block behind it
OPF_SIDE_FREE = 2   # with OPF_SIDE: independent run (own partial region, inputs never overwritten): no waits between runs


class OpList:
    """A recorded launch sequence: a contiguous array of esr_op replayed by ONE C call."""

    def __init__(self):
        self.ops = []
        self._arr = None
        self.keep = []   # tensors referenced by raw pointers in the ops

    def add_conv(self, conv):
        o = esr_op()
        o.kind = OP_CONV
        o.u.conv = conv
        self.ops.append(o)
        self._arr = None
        self.drop_graph()
        return len(self.ops) - 1

    def add(self, kind, field, st, flags=0):
        o = esr_op()
        o.kind = kind
        o.flags = flags
        setattr(o.u, field, st)
        self.ops.append(o)
        self._arr = None
        return len(self.ops) - 1

    def array(self):
        if self._arr is None:
            self._arr = (esr_op * len(self.ops))(*self.ops)
        return self._arr

    def run_timed(self, stream, n=None):
        """Measurement only: per-op elapsed ms of the first n ops (default: all; hipEvents on `stream`; synchronises)."""
        arr = self.array()
        n = len(self.ops) if n is None else n
        ms = (C.c_float * n)()
        check(lib().esr_run_ops_timed(C.cast(arr, C.c_void_p), n, C.c_void_p(stream),
                                      C.cast(ms, C.c_void_p)), 'esr_run_ops_timed')
        return list(ms)

    def run(self, stream):
        if not self.ops:
            return
        arr = self.array()
        check(lib().esr_run_ops(C.cast(arr, C.c_void_p), len(self.ops), C.c_void_p(stream)),
              'esr_run_ops')

    def run_range(self, stream, i0, i1):
        """Ops [i0, i1) of the list (segmented backward: the gradient exchange starts between segments)."""
        if i1 <= i0:
            return
        arr = self.array()
        check(lib().esr_run_ops(C.c_void_p(C.addressof(arr) + i0 * C.sizeof(esr_op)), i1 - i0, C.c_void_p(stream)),
              'esr_run_ops')

    def graph_launch(self, stream):
        """Replay the list as a captured hipGraph (one host call).  The graph is captured on first use
        from the CURRENT contents of the op array: every pointer / scalar is baked in, so callers must
        have bound their I/O to fixed buffers before the first launch."""
        if not self.ops:
            return
        g = self.__dict__.get('_graph')
        if g is None:
            g = C.c_void_p()
            check(lib().esr_graph_create(C.cast(self.array(), C.c_void_p), len(self.ops), C.byref(g)),
                  'esr_graph_create')
            self._graph = g
        check(lib().esr_graph_launch(g, C.c_void_p(stream)), 'esr_graph_launch')

    def drop_graph(self):
        g = self.__dict__.pop('_graph', None)
        if g is not None and _lib is not None:
            _lib.esr_graph_destroy(g)

    def __del__(self):
        try:
            self.drop_graph()
        except Exception:
            pass


def batch_pack_op(packs, device):
    """One OP_PACK_BATCH replacing a list of esr_pack ops; returns (esr_pack_batch, keepalive)."""
    n = len(packs)
    arr = (esr_pack * n)(*packs)
    begins, tot = [], 0
    for pk in packs:
        begins.append(tot)
        tot += lib().esr_pack_pieces(C.byref(pk))
    begins.append(tot)
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
    pb = torch.tensor(begins, dtype=torch.int64, device=device)
    b = esr_pack_batch()
    b.table, b.piece_begin, b.n, b.total_pieces = table.data_ptr(), pb.data_ptr(), n, tot
    return b, (table, pb)
