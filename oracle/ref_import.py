"""ORACLE support (test infrastructure) — import the *real* reference modules in the build
container, CPU-only, without modifying ``/root/reference``.

Only usable where ``/root/reference`` exists (never on the GPU box).  Three shims, as probed in
SURVEY.md §8c:
  1. ``GaussianNoise.__init__`` hard-codes ``.to(torch.device('cuda'))`` (block.py:115,
     test_image/block.py:148) -> during construction ``torch.Tensor.to`` maps cuda -> cpu.
  2. ``architecture.py:4`` imports torchvision (absent) -> a stub module is registered.
  3. ``test_image/*.py`` use top-level ``import block as B`` -> loaded by path under those names.
"""
import contextlib
import importlib
import importlib.util
import os
import sys
import types

import torch

REF = os.environ.get('ESRGAN_REFERENCE', '/root/reference')
sys.dont_write_bytecode = True


def available():
    return os.path.isdir(os.path.join(REF, 'codes', 'models', 'modules'))


@contextlib.contextmanager
def cuda_to_cpu():
    orig = torch.Tensor.to

    def to(self, *a, **k):
        a = tuple(torch.device('cpu') if (isinstance(x, torch.device) and x.type == 'cuda') or
                  (isinstance(x, str) and x.startswith('cuda')) else x for x in a)
        return orig(self, *a, **k)
    torch.Tensor.to = to
    try:
        yield
    finally:
        torch.Tensor.to = orig


def _stub_torchvision():
    if 'torchvision' in sys.modules:
        return
    tv = types.ModuleType('torchvision')
    tvm = types.ModuleType('torchvision.models')
    tvu = types.ModuleType('torchvision.utils')
    tvu.make_grid = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError('stub'))
    tv.models, tv.utils = tvm, tvu
    sys.modules['torchvision'] = tv
    sys.modules['torchvision.models'] = tvm
    sys.modules['torchvision.utils'] = tvu


def codes_arch():
    """-> (architecture, block) modules of codes/models/modules."""
    _stub_torchvision()
    p = os.path.join(REF, 'codes')
    if p not in sys.path:
        sys.path.insert(0, p)
    arch = importlib.import_module('models.modules.architecture')
    block = importlib.import_module('models.modules.block')
    return arch, block


def test_image_arch():
    """-> (architecture, block) of the standalone inference copy test_image/."""
    def load(name, alias):
        spec = importlib.util.spec_from_file_location(alias, os.path.join(REF, 'test_image', name))
        m = importlib.util.module_from_spec(spec)
        sys.modules[alias] = m
        spec.loader.exec_module(m)
        return m
    saved = {k: sys.modules.get(k) for k in ('block', 'architecture')}
    blk = load('block.py', 'block')
    arch = load('architecture.py', 'architecture')
    for k, v in saved.items():   # do not leave the aliases behind
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v
    return arch, blk


def build_rrdbnet(nb=23, variant='codes'):
    with cuda_to_cpu():
        if variant == 'codes':
            arch, _ = codes_arch()
            return arch.RRDBNet(3, 3, 64, nb, gc=32, upscale=4, norm_type=None,
                                act_type='leakyrelu', mode='CNA', upsample_mode='upconv')
        arch, _ = test_image_arch()
        return arch.RRDB_Net(3, 3, 64, nb, gc=32, upscale=4, norm_type=None,
                             act_type='leakyrelu', mode='CNA', res_scale=1,
                             upsample_mode='upconv')


def build_srresnet(nb=16, upsample_mode='pixelshuffle'):
    """networks.py:88-91 with train_SRResNet.json:39-43 (norm_type null, mode CNA)."""
    with cuda_to_cpu():
        arch, _ = codes_arch()
        return arch.SRResNet(in_nc=3, out_nc=3, nf=64, nb=nb, upscale=4, norm_type=None, act_type='relu',
                             mode='CNA', upsample_mode=upsample_mode)


def build_discriminator(size=128):
    arch, _ = codes_arch()
    cls = getattr(arch, 'Discriminator_VGG_%d' % size)
    return cls(in_nc=3, base_nf=64, norm_type='batch', mode='CNA', act_type='leakyrelu')


def build_discriminator_sn():
    arch, _ = codes_arch()
    return arch.Discriminator_VGG_128_SN()


def data_util():
    """codes/data/util.py (imresize / augment).  It imports lmdb and cv2 at module level (absent here,
    unused by the functions we pin) -> empty stub modules; loaded by file path under a private name."""
    for m in ('lmdb', 'cv2'):
        sys.modules.setdefault(m, types.ModuleType(m))
    spec = importlib.util.spec_from_file_location('ref_data_util', os.path.join(REF, 'codes', 'data', 'util.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod

