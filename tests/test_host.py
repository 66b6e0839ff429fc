"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol, ctypes
struct mirrors match the C structs, drop-in modules keep the reference's state-dict layout, and
the product path refuses to run without a GPU (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

from esrganplus_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def built():
    import __graft_entry__ as ge
    ge.build()
    from esrganplus_amd import _lib
    return _lib


def test_library_exports_every_declared_symbol(built):
    hdr = open(os.path.join(ROOT, 'include', 'esrgan_hip.h')).read()
    declared = set(re.findall(r'\b(esr_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(built.EXPORTS), declared ^ set(built.EXPORTS)
    L = ctypes.CDLL(built.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name


def test_struct_mirror_sizes(built):
    L = built.lib()
    assert L.esr_sizeof_op() == ctypes.sizeof(built.esr_op)
    assert L.esr_abi_version() == 6
    assert built.packed_weight_bytes(32, 64, 3, built.ESR_F16) == 1 * 4 * 9 * 1024
    assert built.packed_weight_bytes(64, 192, 3, built.ESR_F32) == 2 * 24 * 9 * 1024
    assert built.g32_dims(128, 128) == (134, 130)
    assert built.g32_dims(57, 86)[1] >= 86 + 2


def test_invalid_arguments_return_error_codes(built):
    L = built.lib()
    c = built.esr_conv()
    assert L.esr_conv_forward(ctypes.byref(c), None) == -1
    assert b'invalid' in L.esr_last_error()
    with pytest.raises(built.HipExtensionError):
        built.check(-1, 'x')


def test_dropin_state_dict_layout():
    from esrganplus_amd import architecture as arch
    for nb in (1, 3):
        sd = synth.rrdbnet_state_dict(nb, 0)
        for cls in (arch.RRDBNet, arch.RRDB_Net):
            net = cls(3, 3, 64, nb)
            assert list(net.state_dict().keys()) == list(sd.keys())
            for k, v in net.state_dict().items():
                assert v.shape == sd[k].shape, k
            net.load_state_dict(sd, strict=True)
    assert sum(p.numel() for p in arch.RRDBNet(3, 3, 64, 23).parameters()) == 16839299


def test_init_weights_by_classname_and_optimizer():
    """networks.py:30-44 pattern: apply(fn) matching 'Conv' in the class name, then Adam."""
    from esrganplus_amd import architecture as arch
    net = arch.RRDBNet(3, 3, 64, 1)
    seen = []

    def init(m):
        if m.__class__.__name__.find('Conv') != -1:
            torch.nn.init.kaiming_normal_(m.weight.data, a=0, mode='fan_in')
            m.weight.data *= 0.1
            if m.bias is not None:
                m.bias.data.zero_()
            seen.append(m)
    net.apply(init)
    assert len(seen) == 1 + 3 * 6 + 1 + 4
    assert net._force_repack
    opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=1e-4)
    assert len(opt.param_groups[0]['params']) == 45
    assert str(net).startswith('RRDBNet(')
    wrapped = torch.nn.DataParallel(net)
    assert wrapped.module is net


def test_unsupported_configs_raise_like_reference():
    from esrganplus_amd import architecture as arch, block
    with pytest.raises(NotImplementedError):
        arch.RRDBNet(3, 3, 64, 1, upsample_mode='bogus')
    with pytest.raises(NotImplementedError):
        block.act('swish')
    with pytest.raises(NotImplementedError):
        block.norm('layer', 8)


def test_no_cpu_fallback():
    from esrganplus_amd import architecture as arch, _lib
    net = arch.RRDBNet(3, 3, 64, 1).eval()
    with torch.no_grad(), pytest.raises(_lib.HipExtensionError):
        net(torch.rand(1, 3, 8, 8))


def test_metrics_match_reference_goldens(golden):
    """esrganplus_amd.metrics (tensor2img / PSNR harness code) against the reference's util.py."""
    import numpy as np
    from esrganplus_amd import metrics
    g = golden('psnr')
    for i in range(3):
        a, b = torch.from_numpy(g['a%d' % i]), torch.from_numpy(g['b%d' % i])
        assert np.array_equal(metrics.tensor2img(a), g['img_a%d' % i])
        assert abs(metrics.validation_psnr(a, b, 4) - float(g['psnr%d' % i])) < 1e-9


def test_adam_tables_cover_every_element_once():
    from esrganplus_amd import optim, _lib as L
    sizes = [1, 31, L.ADAM_BLOCK_ELEMS, L.ADAM_BLOCK_ELEMS + 1, 3 * L.ADAM_BLOCK_ELEMS + 17, 64]
    goff, blocks, total = optim.adam_tables(sizes)
    assert total == sum(sizes) and goff == [sum(sizes[:i]) for i in range(len(sizes))]
    seen = [0] * len(sizes)
    for e, first in blocks:
        assert first % L.ADAM_BLOCK_ELEMS == 0 and first < sizes[e]
        seen[e] += min(L.ADAM_BLOCK_ELEMS, sizes[e] - first)
    assert seen == sizes


def test_net_interp_and_network_roundtrip(tmp_path):
    import torch
    from esrganplus_amd import architecture as arch, checkpoint as ck, synth
    a, b = synth.rrdbnet_state_dict(nb=1, seed=1), synth.rrdbnet_state_dict(nb=1, seed=2)
    mid = ck.interpolate(a, b, 0.8)
    k = 'model.1.sub.0.RDB2.conv3.0.weight'
    assert torch.allclose(mid[k], 0.2 * a[k] + 0.8 * b[k])
    net = arch.RRDBNet(3, 3, 64, 1)
    net.load_state_dict(mid, strict=True)
    ck.save_network(torch.nn.DataParallel(net), str(tmp_path / 'n.pth'))
    net2 = arch.RRDBNet(3, 3, 64, 1)
    ck.load_network(str(tmp_path / 'n.pth'), net2)
    assert all(torch.equal(v, net2.state_dict()[kk]) for kk, v in net.state_dict().items())


def test_data_parallel_replica_does_not_inherit_caches():
    from esrganplus_amd import architecture as arch
    net = arch.RRDBNet(3, 3, 64, 1)
    net._convs()                                   # populate the cache on the original
    net._plans['x'] = object()
    rep = net._replicate_for_data_parallel()
    assert rep.__dict__['_conv_cache'] is None and rep._plans == {} and rep._wp == {}
    assert net.__dict__['_conv_cache'] is not None and 'x' in net._plans


def test_data_parallel_replica_takes_the_per_tensor_gradient_route():
    """nn.DataParallel's replicas hold NON-LEAF copies of the weights (Broadcast outputs); assigning `.grad`
    on them (the flat route) would drop G's gradients.  The flat route is only taken on leaf parameters, and a store that
    was marked stale is zeroed before a per-tensor backward adds into it."""
    import torch
    from esrganplus_amd import architecture as arch, functional as F_
    net = arch.RRDBNet(3, 3, 64, 1)
    assert net.flat_param_grads and F_._flat_grad_route(net, net._convs()[1])
    net.__dict__['_grad_proxy'] = torch.zeros(1, requires_grad=True)
    rep = net._replicate_for_data_parallel()
    assert '_grad_proxy' not in rep.__dict__ and '_grad_proxy' in net.__dict__
    copies = [p * 1.0 for p in net._convs()[1]]                     # what replicate() swaps in: requires_grad, not leaf
    assert all(c.requires_grad and not c.is_leaf for c in copies)
    assert not F_._flat_grad_route(net, copies)
    frozen = net._convs()[1]
    frozen[3].requires_grad_(False)
    assert not F_._flat_grad_route(net, frozen)
    frozen[3].requires_grad_(True)
    gs = net._grad_store(torch.device('cpu'))
    gs['flat'].fill_(2.0)
    for p_, v in zip(gs['params'], gs['views']):
        p_.grad = v
    assert net.mark_grads_stale() and gs['stale']
    net._flush_stale_grads()
    assert not gs['stale'] and float(gs['flat'].abs().sum()) == 0.0 and float(frozen[0].grad.abs().sum()) == 0.0


def test_header_is_plain_c_and_structs_match_ctypes(tmp_path):
    """include/esrgan_hip.h must be consumable from C (the boundary is a C ABI) and EVERY struct it declares must
    have a ctypes mirror in _lib.py with the size AND the field offsets the C compiler gives it.  The struct list
    is read from the header, so a struct added there without a mirror fails here."""
    import ctypes as C
    import re
    import subprocess
    from esrganplus_amd import _lib as L
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'esrgan_hip.h')
    subprocess.check_call(['gcc', '-std=c99', '-fsyntax-only', '-x', 'c', hdr])
    names = re.findall(r'^typedef struct (esr_\w+) \{', open(hdr).read(), flags=re.M)
    assert len(names) >= 26 and {'esr_rdb_block', 'esr_rdb_chain', 'esr_rdb_wgrad', 'esr_rdb_wgrad_block', 'esr_l1_loss',
                                 'esr_ragan_loss', 'esr_img_metrics', 'esr_amp', 'esr_frag_gather',
                                 'esr_unperm_entry', 'esr_op'} <= set(names), names
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, 'structs of the header without a ctypes mirror: %s' % missing
    prints = []
    for n in names:
        prints.append('printf("%s . %%zu\\n", sizeof(%s));' % (n, n))
        for f in getattr(L, n)._fields_:
            cname = f[0][:-1] if f[0].endswith('_') and not f[0].startswith('_') else f[0]   # in_ -> in
            prints.append('printf("%s %s %%zu\\n", offsetof(%s, %s));' % (n, f[0], n, cname))
    src = tmp_path / 'sz.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "%s"\nint main(void){%s return 0;}\n'
                   % (hdr, '\n'.join(prints)))
    exe = tmp_path / 'sz'
    subprocess.check_call(['gcc', '-std=c99', str(src), '-o', str(exe)])
    seen = 0
    for line in subprocess.check_output([str(exe)]).decode().splitlines():
        n, f, v = line.split()
        st = getattr(L, n)
        got = C.sizeof(st) if f == '.' else getattr(st, f).offset
        assert int(v) == got, (n, f, int(v), got)
        seen += 1
    assert seen == len(prints)
    # the members of the op union are exactly the header's
    union_c = re.search(r'typedef struct esr_op \{.*?union \{(.*?)\} u;', open(hdr).read(), flags=re.S).group(1)
    assert re.findall(r'(\w+);', union_c) == [f[0] for f in L._op_union._fields_]


def test_committed_bench_line_follows_the_contract():
    """profiles/r01_bench.json is the line `python bench.py` printed on the MI355X: every key the
    measurement contract names must be there, with the headline workload and a roofline/cpu_baseline."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.loads(open(os.path.join(root, 'profiles', 'r01_bench.json')).read().strip().splitlines()[-1])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
              'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['unit'] == 'HR-Mpix/s' and d['higher_is_better'] is True and d['scaling'] == 'weak'
    assert d['vs_baseline'] is None and d['dtype'] == 'f16' and d['data'] == 'synthetic'
    assert 'workload' in d['config'] and 'configs[1]' in d['config']['workload']
    r = d['roofline']
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s')
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3 and (r['traffic'] is None or r['traffic'] > 0)
    c = d['cpu_baseline']
    assert c['kind'] in ('reference', 'port') and c['cores'] >= 1 and c['value'] > 0 and c['sample']
    # consistency of the headline itself: value = HR megapixels of one step / time of one step
    hr_mpix = d['config']['global_batch'] * 512 * 512 / 1e6
    assert abs(d['value'] - hr_mpix / (d['ms_per_step'] * 1e-3)) / d['value'] < 1e-3


def test_srresnet_state_dict_keys_match_the_reference_layout():
    """SRResNet / pixelshuffle_block build on CPU with the reference's key names and shapes (synth.srresnet_keys
    was accepted by the imported reference with strict=True when the fixtures were generated)."""
    from esrganplus_amd import architecture as arch, synth
    for mode in ('pixelshuffle', 'upconv'):
        net = arch.SRResNet(3, 3, 64, 2, upsample_mode=mode)
        sd = synth.srresnet_state_dict(nb=2, seed=1, upsample_mode=mode)
        assert list(net.state_dict().keys()) == list(sd.keys())
        assert all(tuple(net.state_dict()[k].shape) == tuple(v.shape) for k, v in sd.items())
        net.load_state_dict(sd, strict=True)


def test_bare_name_architecture_and_block_shims():
    """test_image/test.py:7-21 imports `architecture` by bare name (and test_image/architecture.py:4 `block`):
    with dropin/ on sys.path instead of the reference's test_image/, the script's statements run unchanged."""
    import subprocess
    import sys
    code = r'''
import sys, os
sys.path.insert(0, os.path.join(%r, 'dropin'))
import torch
import architecture as arch
import block as B
model = arch.RRDB_Net(3, 3, 64, 23, gc=32, upscale=4, norm_type=None, act_type='leakyrelu', \
                        mode='CNA', res_scale=1, upsample_mode='upconv')
from esrganplus_amd import synth
model.load_state_dict(synth.rrdbnet_state_dict(23, 0), strict=False)
model.eval()
for k, v in model.named_parameters():
    v.requires_grad = False
assert len(model.state_dict()) == 771 and arch.RRDB_Net.__module__ == 'esrganplus_amd.architecture'
assert B.ResidualDenseBlock_5C.__module__ == 'esrganplus_amd.block' and callable(B.upconv_blcok)
print('ok')
''' % ROOT
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith('ok'), out.stderr[-2000:]


def test_discriminator_sn_state_dict_keys_and_power_iteration():
    """Discriminator_VGG_128_SN keeps the reference's state-dict entries (spectral_norm.py:55-75: weight_orig, bias,
    weight, weight_u per layer) and its one-step power iteration / sigma are the textbook ones (CPU tensors)."""
    import torch
    from esrganplus_amd import architecture as arch, synth
    net = arch.Discriminator_VGG_128_SN()
    sd = synth.discriminator_sn_state_dict(seed=2)
    net.load_state_dict(sd, strict=True)
    assert sorted(net.state_dict().keys()) == sorted(sd.keys()) and len(sd) == 48
    m = net.conv3
    w, u0 = sd['conv3.weight_orig'], sd['conv3.weight_u']
    wm = w.reshape(w.shape[0], -1).double()
    v = wm.t() @ u0.double()
    v = v / v.norm()
    u = wm @ v
    u = u / u.norm()
    sigma = u @ (wm @ v)
    w_eff = m.normalised()
    assert (m.weight_u.double() - u).abs().max().item() < 1e-6
    assert (w_eff.detach().double() - w.double() / sigma).abs().max().item() < 1e-6
    assert torch.equal(m.weight, w_eff.detach())
    w_eff.sum().backward()                                 # the gradient runs through sigma to weight_orig
    assert m.weight_orig.grad is not None and torch.isfinite(m.weight_orig.grad).all()


def test_flat_gradient_store_follows_autograd_accumulation_rules():
    """block._PlannedModule._deliver_flat_grads — how the fused backward of RRDBNet hands ~770 parameter gradients over
    without one autograd output per tensor — must behave like AccumulateGrad: None -> set, present -> add, and a
    store marked stale (train.ESRGANPlusStep's stand-in for zero_grad) -> overwrite; foreign gradients are honoured."""
    from esrganplus_amd import architecture as arch
    from esrganplus_amd.optim import FusedAdam
    net = arch.RRDBNet(3, 3, 64, 1)
    params = list(net.parameters())
    n = sum(p.numel() for p in params)
    assert [id(p) for p in net._convs()[1]] == [id(p) for p in params]     # store order == optimizer order
    a = torch.arange(n, dtype=torch.float32) / n
    assert not net.mark_grads_stale()                   # nothing delivered yet
    net._deliver_flat_grads(a)
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    assert torch.equal(flat, a) and all(p.grad.shape == p.shape for p in params)
    views = [p.grad for p in params]
    net._deliver_flat_grads(a)                          # second backward without zero_grad: accumulated
    assert all(p.grad is v for p, v in zip(params, views))
    assert torch.equal(torch.cat([p.grad.reshape(-1) for p in params]), 2 * a)
    assert net.mark_grads_stale()
    net._deliver_flat_grads(3 * a)                      # stale: overwritten
    assert torch.equal(torch.cat([p.grad.reshape(-1) for p in params]), 3 * a)
    net.zero_grad(set_to_none=True)
    net._deliver_flat_grads(a)                          # None everywhere: set to the store's views again
    assert all(p.grad is v for p, v in zip(params, views))
    assert torch.equal(torch.cat([p.grad.reshape(-1) for p in params]), a)
    # the views tile one storage in optimizer order: FusedAdam's zero-copy test accepts them
    opt = FusedAdam(params)
    st = dict(goff=[0] * len(params), total=n)
    off = 0
    for i, p in enumerate(params):
        st['goff'][i] = off
        off += p.numel()
    got = opt._flat_grad(st, params)
    assert got.data_ptr() == views[0].data_ptr() and got.numel() == n
    assert opt._flat_grad(st, params) is got            # identity fast path on the next step
    # a foreign gradient on one tensor: tensor-by-tensor accumulation
    params[3].grad = torch.ones_like(params[3])
    net._deliver_flat_grads(a)
    assert torch.equal(params[3].grad, torch.ones_like(params[3]) + a.split([p.numel() for p in params])[3].view_as(params[3]))
    assert not net.mark_grads_stale()


def test_adopted_plan_gradients_alias_without_a_copy_and_keep_the_accumulation_rules():
    """`_deliver_flat_grads(adopt=True)` (the hand-written training loops): an OVERWRITING delivery — gradients None or
    marked stale — makes `.grad` views of the plan's own buffer instead of copying 67 MB into the module's store; an
    accumulating one (a second backward without zero_grad) still adds, through the copying route; another plan's buffer
    is adopted in turn; FusedAdam's zero-copy test accepts the aliased views."""
    from esrganplus_amd import architecture as arch
    from esrganplus_amd.optim import FusedAdam
    net = arch.RRDBNet(3, 3, 64, 1)
    params = list(net.parameters())
    n = sum(p.numel() for p in params)
    a, b = torch.arange(n, dtype=torch.float32) / n, torch.ones(n)
    net._deliver_flat_grads(a, adopt=True)              # None everywhere: adopted
    assert params[0].grad.data_ptr() == a.data_ptr() and torch.equal(torch.cat([p.grad.reshape(-1) for p in params]), a)
    views = [p.grad for p in params]
    assert net.mark_grads_stale()
    a.mul_(2.0)                                         # "the next backward wrote the plan's buffer"
    net._deliver_flat_grads(a, adopt=True)              # stale + same buffer: nothing to do
    assert all(p.grad is v for p, v in zip(params, views)) and not net._gstore['stale']
    net._deliver_flat_grads(b, adopt=True)              # NOT stale: accumulate -> copying route, a + b, a untouched
    got = torch.cat([p.grad.reshape(-1) for p in params])
    assert torch.equal(got, a + b) and params[0].grad.data_ptr() not in (a.data_ptr(), b.data_ptr())
    net.zero_grad(set_to_none=True)
    net._deliver_flat_grads(b, adopt=True)              # another plan's buffer
    assert params[0].grad.data_ptr() == b.data_ptr()
    assert net.mark_grads_stale()
    net._deliver_flat_grads(a, adopt=True)              # stale, other buffer: re-pointed
    assert params[0].grad.data_ptr() == a.data_ptr() and all(p.grad.shape == p.shape for p in params)
    opt = FusedAdam(params)
    st, off = dict(goff=[0] * len(params), total=n), 0
    for i, p in enumerate(params):
        st['goff'][i] = off
        off += p.numel()
    assert opt._flat_grad(st, params).data_ptr() == a.data_ptr()
    assert net.mark_grads_stale()
    net._flush_stale_grads()                            # a per-tensor backward is about to ADD: the stale values go
    assert float(a.abs().sum()) == 0.0 and not net._gstore['stale']
    assert params[0].grad.data_ptr() == net._gstore['flat'].data_ptr()     # ... and `.grad` no longer aliases the plan


def test_adopted_gradients_survive_the_plan_buffer_being_rewritten():
    """adopted `.grad` views alias the plan's buffer, which the plan's NEXT backward zeroes and rewrites
    before it delivers.  The three sequences that gave 2 * g (expected 4 / 3 / 3, got 6):
    (a) a second backward on the same plan without zero_grad -> g1 + g2;
    (b) mark_grads_stale() and then a delivery with adopt=False (the train step falling back to autograd) -> g2;
    (c) zero_grad(set_to_none=False) after an adopted step -> g2; and (d) the per-tensor route (AccumulateGrad)."""
    from esrganplus_amd import architecture as arch
    net = arch.RRDBNet(3, 3, 64, 1)
    params = list(net.parameters())
    n = sum(p.numel() for p in params)

    def cat():
        return torch.cat([p.grad.reshape(-1) for p in params])

    def backward(buf, g, adopt):                        # what functional.rrdbnet_train_backward / _RRDBNetFn.backward do
        net._release_adopted(buf)
        buf.zero_()
        buf.add_(g)
        net._deliver_flat_grads(buf, adopt=adopt)
    buf = torch.zeros(n)
    backward(buf, 1.0, True)                            # (a)
    assert params[0].grad.data_ptr() == buf.data_ptr()
    backward(buf, 3.0, True)
    assert torch.equal(cat(), torch.full((n,), 4.0)) and params[0].grad.data_ptr() != buf.data_ptr()
    net.zero_grad(set_to_none=True)
    backward(buf, 1.0, True)                            # (b)
    assert net.mark_grads_stale()
    backward(buf, 3.0, False)
    assert torch.equal(cat(), torch.full((n,), 3.0)) and not net._gstore['stale']
    assert params[0].grad.data_ptr() != buf.data_ptr()
    net.zero_grad(set_to_none=True)
    backward(buf, 1.0, True)                            # (c)
    net.zero_grad(set_to_none=False)
    backward(buf, 3.0, True)
    assert torch.equal(cat(), torch.full((n,), 3.0))
    net.zero_grad(set_to_none=True)
    backward(buf, 1.0, True)                            # (d) per-tensor route: forward flushes, backward rewrites, AccumulateGrad adds
    assert net.mark_grads_stale()
    net._flush_stale_grads()
    buf.zero_(), buf.add_(3.0)
    for p_, g in zip(params, buf.clone().split([p.numel() for p in params])):
        p_.grad.add_(g.view_as(p_))
    assert torch.equal(cat(), torch.full((n,), 3.0))
    net.zero_grad(set_to_none=True)
    backward(buf, 1.0, True)                            # (d'), not stale: the old gradients count
    net._flush_stale_grads()
    buf.zero_(), buf.add_(3.0)
    for p_, g in zip(params, buf.clone().split([p.numel() for p in params])):
        p_.grad.add_(g.view_as(p_))
    assert torch.equal(cat(), torch.full((n,), 4.0))
    # without the release hook the aliasing is refused loudly instead of returning 2 * g
    net.zero_grad(set_to_none=True)
    backward(buf, 1.0, True)
    buf.zero_(), buf.add_(3.0)
    with pytest.raises(RuntimeError, match='rewritten'):
        net._deliver_flat_grads(buf, adopt=True)


def test_committed_traffic_numbers_belong_to_the_current_kernels():
    """bench.py quotes `roofline.traffic` from profiles/roofline_traffic.json (separate rocprofv3 --pmc passes): every
    row must have been measured on the kernel sources in the tree (tools/traffic_hashes.py: sha256 over the files that
    define the kernel) — editing a kernel without re-measuring its traffic fails here instead of shipping a stale
    number."""
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import traffic_hashes as TH
    tj = json.load(open(os.path.join(ROOT, 'profiles', 'roofline_traffic.json')))
    rows = [k for k in tj if not k.startswith('_')]
    assert set(rows) <= set(TH.SOURCES), set(rows) - set(TH.SOURCES)
    for k in rows:
        assert tj[k]['read_bytes'] > 0 and tj[k]['write_bytes'] > 0
        assert TH.fresh(tj, k), '%s: traffic was measured on other sources (re-run the PMC pass, then tools/traffic_hashes.py --update)' % k


def test_integer_schedule_knobs_are_validated(monkeypatch):
    """ESR_BWD_SPLIT / ESR_BWD_SPLIT_FIRST were parsed with a bare int(): a misspelt value raised an opaque
    ValueError in the middle of a plan build, an out-of-range one was silently ignored."""
    from esrganplus_amd import engine as E
    assert E.env_int('ESR_NO_SUCH_KNOB', 7, 0, 9) == 7
    monkeypatch.setenv('ESR_BWD_SPLIT_FIRST', 'ten')
    with pytest.raises(ValueError, match='ESR_BWD_SPLIT_FIRST'):
        E.env_int('ESR_BWD_SPLIT_FIRST', 0, 0, 23)
    monkeypatch.setenv('ESR_BWD_SPLIT_FIRST', '24')
    with pytest.raises(ValueError, match=r'\[0, 23\]'):
        E.env_int('ESR_BWD_SPLIT_FIRST', 0, 0, 23)
    monkeypatch.setenv('ESR_BWD_SPLIT_FIRST', '14')
    assert E.env_int('ESR_BWD_SPLIT_FIRST', 0, 0, 23) == 14
    monkeypatch.setenv('ESR_BWD_SPLIT', '0')
    with pytest.raises(ValueError, match='ESR_BWD_SPLIT'):
        E.env_int('ESR_BWD_SPLIT', None, 1, 23)
