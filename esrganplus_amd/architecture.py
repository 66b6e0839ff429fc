"""Drop-in for the reference's ``architecture`` module on the ESRGAN+ hot path:
``RRDBNet`` (codes/models/modules/architecture.py:47-78) and ``RRDB_Net``
(test_image/architecture.py:7-38).  Same constructor signatures, module tree and state-dict keys
(SURVEY.md Appendix B); ``forward`` replays a fused HIP launch plan instead of walking the
``nn.Sequential``.  Generators outside the path (SRResNet, pixelshuffle) raise NotImplementedError.
"""
import math

import torch.nn as nn

from . import block as B
from .functional import run_rrdbnet


class _RRDBNetBase(B._PlannedModule):
    def _build(self, in_nc, out_nc, nf, nb, upscale, norm_type, act_type, mode, upsample_mode,
               extra_noise):
        if upsample_mode == 'pixelshuffle':
            B.pixelshuffle_block()
        if upsample_mode != 'upconv':
            raise NotImplementedError('upsample mode [{:s}] is not found'.format(upsample_mode))
        if (nf, upscale, norm_type, act_type.lower(), mode) != (64, 4, None, 'leakyrelu', 'CNA'):
            raise NotImplementedError('HIP RRDBNet supports the ESRGAN+ configuration nf=64, x4, '
                                      'no norm, leakyrelu, CNA (train_ESRGANplus.json:36-45)')
        self.in_nc, self.out_nc, self.nb = in_nc, out_nc, nb
        n_up = int(math.log(upscale, 2))
        fea_conv = B.conv_block(in_nc, nf, kernel_size=3, norm_type=None, act_type=None)
        # NB: the reference ignores its ``gc`` argument and always builds gc=32
        # (architecture.py:56) — so do we.
        blocks = [B.RRDB(nf, kernel_size=3, gc=32, stride=1, bias=True, pad_type='zero',
                         norm_type=norm_type, act_type=act_type, mode='CNA',
                         extra_noise=extra_noise) for _ in range(nb)]
        lr_conv = B.conv_block(nf, nf, kernel_size=3, norm_type=norm_type, act_type=None, mode=mode)
        ups = [B.upconv_blcok(nf, nf, act_type=act_type) for _ in range(n_up)]
        hr0 = B.conv_block(nf, nf, kernel_size=3, norm_type=None, act_type=act_type)
        hr1 = B.conv_block(nf, out_nc, kernel_size=3, norm_type=None, act_type=None)
        self.model = B.sequential(fea_conv, B.ShortcutBlock(B.sequential(*blocks, lr_conv)),
                                  *ups, hr0, hr1)
        self._init_planned()
        self.max_cached_plans = 4
        self.variant = 'test_image' if extra_noise else 'codes'

    def _conv_list(self):
        m, nb = self.model, self.nb
        out = [('model.0', m[0].weight, m[0].bias)]
        for i in range(nb):
            rr = m[1].sub[i]
            for j in (1, 2, 3):
                out += B._rdb_convs('model.1.sub.%d.RDB%d' % (i, j), getattr(rr, 'RDB%d' % j))
        lr = m[1].sub[nb]
        out.append(('model.1.sub.%d' % nb, lr.weight, lr.bias))
        for idx in (3, 6, 8, 10):
            out.append(('model.%d' % idx, m[idx].weight, m[idx].bias))
        return out

    def _dgrad_special(self):
        sp = {'model.3': {'ups': True}, 'model.6': {'ups': True}}
        for i in range(self.nb):
            for j in (1, 2, 3):
                # x4 = lrelu(a4) + x2 (block.py:266): dL/dx2 also receives conv5's x4 slice
                sp['model.1.sub.%d.RDB%d.conv5.0' % (i, j)] = {'sum': (96, 160, 32)}
        return sp

    def forward(self, x, z=None):
        """x: NCHW float32 in [0,1] on the MI355X -> [B, out_nc, 4H, 4W] float32.
        ``z`` (training mode only): explicit N(0,1) tensors, one [B,64,H,W] per noise layer in
        execution order, for bit-parity tests; default = fused Philox stream."""
        return run_rrdbnet(self, x, z)


class RRDBNet(_RRDBNetBase):
    """codes/models/modules/architecture.py:47-78."""

    def __init__(self, in_nc, out_nc, nf, nb, gc=32, upscale=4, norm_type=None,
                 act_type='leakyrelu', mode='CNA', upsample_mode='upconv'):
        super().__init__()
        self._build(in_nc, out_nc, nf, nb, upscale, norm_type, act_type, mode, upsample_mode, False)


class RRDB_Net(_RRDBNetBase):
    """test_image/architecture.py:7-38 (inference copy: extra noise layer per RRDB in train
    mode, test_image/block.py:250,256; unused ``res_scale``)."""

    def __init__(self, in_nc, out_nc, nf, nb, gc=32, upscale=4, norm_type=None,
                 act_type='leakyrelu', mode='CNA', res_scale=1, upsample_mode='upconv'):
        super().__init__()
        self._build(in_nc, out_nc, nf, nb, upscale, norm_type, act_type, mode, upsample_mode, True)


class SRResNet(nn.Module):
    def __init__(self, *a, **k):
        raise NotImplementedError('SRResNet is outside the ESRGAN+ hot path (SURVEY.md §2.1 row 2)')
