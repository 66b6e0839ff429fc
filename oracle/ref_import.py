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
# Nothing may ever be written under /root/reference: no bytecode at all, and should some later import switch bytecode
# back on, it goes to a scratch prefix instead of the source tree's __pycache__.
sys.dont_write_bytecode = True
sys.pycache_prefix = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'esr_ref_pycache')


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


def cv2_shim():
    """The two cv2 entry points codes/utils/util.py:117-137 (``ssim``) calls, registered on the ``cv2`` module object the
    reference's ``import cv2`` binds (cv2 is absent from this image).  NOT a restatement of the reference: these are
    OpenCV's documented semantics for exactly the call forms the reference uses —
      * ``getGaussianKernel(ksize, sigma)``: (ksize, 1) float64 column ``exp(-(i-(ksize-1)/2)^2 / (2 sigma^2))``
        normalised to sum 1 (OpenCV's formula whenever sigma > 0 and ksize > 7: no fixed small-kernel table);
      * ``filter2D(src, -1, kernel)``: correlation (no kernel flip), anchor at the kernel centre, output depth = source
        depth, border ``BORDER_REFLECT_101`` (``gfedcb|abcdefgh|gfedcba`` = scipy's ``mode='mirror'``), every channel
        of an H x W x C image filtered with the same 2-D kernel.
    With it the reference's own ``ssim`` / ``calculate_ssim`` lines run (oracle/gen_golden.py: gen_ssim)."""
    import numpy as np
    from scipy.ndimage import correlate
    cv2 = sys.modules.setdefault('cv2', types.ModuleType('cv2'))

    def getGaussianKernel(ksize, sigma, ktype=None):
        assert sigma > 0 and ksize % 2 == 1 and ksize > 7, 'shim covers the reference\'s call form only'
        i = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2.0
        k = np.exp(-(i * i) / (2.0 * sigma * sigma))
        return (k / k.sum()).reshape(ksize, 1)

    def filter2D(src, ddepth, kernel):
        assert ddepth == -1 and src.dtype == np.float64 and kernel.shape[0] % 2 == 1 and kernel.shape[1] % 2 == 1
        if src.ndim == 2:
            return correlate(src, kernel, mode='mirror')
        return np.stack([correlate(src[..., c], kernel, mode='mirror') for c in range(src.shape[2])], axis=-1)

    cv2.getGaussianKernel, cv2.filter2D = getGaussianKernel, filter2D
    return cv2


def utils_util():
    """codes/utils/util.py of the reference (tensor2img, calculate_psnr, ssim, calculate_ssim), cv2 as ``cv2_shim``."""
    cv2_shim()
    _stub_torchvision()
    p = os.path.join(REF, 'codes')
    if p not in sys.path:
        sys.path.insert(0, p)
    return importlib.import_module('utils.util')


def data_util():
    """codes/data/util.py (imresize / augment).  It imports lmdb and cv2 at module level (absent here,
    unused by the functions we pin) -> empty stub modules; loaded by file path under a private name."""
    for m in ('lmdb', 'cv2'):
        sys.modules.setdefault(m, types.ModuleType(m))
    spec = importlib.util.spec_from_file_location('ref_data_util', os.path.join(REF, 'codes', 'data', 'util.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod

