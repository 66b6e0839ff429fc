"""Data path on the GPU: what ``LRHRDataset.__getitem__`` does per sample on a CPU
worker (codes/data/LRHR_dataset.py:80-112, codes/data/util.py:94-106,213-343) done per BATCH on the
device, so eight GPUs are not fed by eight Python processes looping over image rows.

* ``imresize`` — MATLAB-compatible bicubic resize with antialiasing (util.py:276-343 /
  ``imresize_np`` 345-412): output size ceil(in*scale), cubic kernel of width 4 (4/scale when
  shrinking), weights normalised per output sample, symmetric border, H pass then W pass, float32.
  The tiny weight / index tables are built on the host exactly as the reference builds them (same
  float32 torch ops); the two gather passes are HIP launches (``esr_resample_axis``).
* ``paired_random_crop`` / ``augment`` — the crop + flip/rot logic with the reference's use of
  Python's ``random`` (same call order), applied to NCHW device tensors."""
import ctypes as C
import math
import random

import torch

from . import _lib as L
from . import engine as E


def _cubic(x):
    """util.py:213-218."""
    ax = torch.abs(x)
    ax2, ax3 = ax ** 2, ax ** 3
    return ((1.5 * ax3 - 2.5 * ax2 + 1) * (ax <= 1).type_as(ax)
            + (-0.5 * ax3 + 2.5 * ax2 - 4 * ax + 2) * ((ax > 1) * (ax <= 2)).type_as(ax))


def resample_tables(in_length, scale, antialiasing=True):
    """Weights and SOURCE indices of one axis (util.py:221-273 with the symmetric padding of
    276-343 folded into the index table).  -> (weights [out, P] float32, src [out, P] int32, out_len)."""
    out_length = math.ceil(in_length * scale)
    kernel_width = 4.0
    shrink = scale < 1 and antialiasing
    if shrink:
        kernel_width = kernel_width / scale
    x = torch.linspace(1, out_length, out_length)
    u = x / scale + 0.5 * (1 - 1 / scale)
    left = torch.floor(u - kernel_width / 2)
    P = math.ceil(kernel_width) + 2
    indices = left.view(out_length, 1).expand(out_length, P) + torch.linspace(0, P - 1, P).view(1, P).expand(
        out_length, P)
    dist = u.view(out_length, 1).expand(out_length, P) - indices
    weights = scale * _cubic(dist * scale) if shrink else _cubic(dist)
    weights = weights / torch.sum(weights, 1).view(out_length, 1).expand(out_length, P)
    zero_cols = torch.sum((weights == 0), 0)
    if not math.isclose(zero_cols[0], 0, rel_tol=1e-6):
        indices, weights = indices.narrow(1, 1, P - 2), weights.narrow(1, 1, P - 2)
    if not math.isclose(zero_cols[-1], 0, rel_tol=1e-6):
        indices, weights = indices.narrow(1, 0, P - 2), weights.narrow(1, 0, P - 2)
    sym_s = int(-indices.min() + 1)
    # 1-based input coordinate c (possibly < 1 or > in_length) -> mirrored 0-based source index
    c = indices.long()
    src = torch.where(c < 1, -c, torch.where(c > in_length, 2 * in_length - c + 1 - 1, c - 1))
    assert sym_s >= 0 and int(src.min()) >= 0 and int(src.max()) < in_length
    return weights.contiguous().float(), src.to(torch.int32).contiguous(), out_length


def _axis_pass(x, axis, w, idx, out_len, stream):
    n, c, h, wd = x.shape
    out = torch.empty((n, c, out_len, wd) if axis == 0 else (n, c, h, out_len), dtype=torch.float32, device=x.device)
    a = L.esr_resample()
    a.in_, a.out = x.data_ptr(), out.data_ptr()
    a.planes, a.in_h, a.in_w, a.out_len, a.axis, a.taps = n * c, h, wd, out_len, axis, w.shape[1]
    a.w, a.idx = w.data_ptr(), idx.data_ptr()
    L.check(L.lib().esr_resample_axis(C.byref(a), C.c_void_p(stream)), 'esr_resample_axis')
    return out


_TABLES = {}


def imresize(img, scale, antialiasing=True):
    """img: [C,H,W] or [N,C,H,W] float32 on the MI355X, range [0,1], not rounded (util.py:276-343)."""
    E.require_cuda(img, 'image')
    squeeze = img.dim() == 3
    x = (img.unsqueeze(0) if squeeze else img).contiguous().float()
    n, c, h, wd = x.shape
    dev, st = x.device, E.current_stream()
    tabs = []
    for length in (h, wd):
        key = (length, float(scale), bool(antialiasing), str(dev))
        t = _TABLES.get(key)
        if t is None:
            w, idx, ol = resample_tables(length, scale, antialiasing)
            t = (w.to(dev), idx.to(dev), ol)
            if len(_TABLES) >= 64:            # (length, scale) pairs of a dataset are few; stay bounded anyway
                _TABLES.clear()
            _TABLES[key] = t
        tabs.append(t)
    y = _axis_pass(x, 0, tabs[0][0], tabs[0][1], tabs[0][2], st)
    y = _axis_pass(y, 1, tabs[1][0], tabs[1][1], tabs[1][2], st)
    return y[0] if squeeze else y


def paired_random_crop(lr, hr, lr_size, scale):
    """LRHR_dataset.py:96-103 on NCHW batches.  The reference's ``__getitem__`` draws one window PER SAMPLE
    (two ``random.randint`` calls each, in sample order); a batch does the same — B independent windows, the
    same consumption of Python's ``random`` stream as B dataset items.  A 3-D tensor is one sample."""
    single = lr.dim() == 3
    if single:
        lr, hr = lr[None], hr[None]
    h, w = lr.shape[-2:]
    hs = lr_size * scale
    outl, outh = [], []
    for b in range(lr.shape[0]):
        rnd_h = random.randint(0, max(0, h - lr_size))
        rnd_w = random.randint(0, max(0, w - lr_size))
        rh, rw = int(rnd_h * scale), int(rnd_w * scale)
        outl.append(lr[b, :, rnd_h:rnd_h + lr_size, rnd_w:rnd_w + lr_size])
        outh.append(hr[b, :, rh:rh + hs, rw:rw + hs])
    if single:
        return outl[0], outh[0]
    return torch.stack(outl), torch.stack(outh)


def augment(img_list, hflip=True, rot=True):
    """util.py:94-106: horizontal flip, vertical flip, transpose.  For NCHW batches the three coin flips are
    drawn per SAMPLE (as B dataset items would), applied to the same sample of every tensor in ``img_list``;
    [C,H,W] tensors are one sample.  Batches need square images when ``rot`` is on (a transposed sample must
    stack with an un-transposed one), which the reference's fixed-size crops guarantee."""
    if img_list[0].dim() == 3:
        return [t[0] for t in augment([t[None] for t in img_list], hflip, rot)]
    B = img_list[0].shape[0]
    flags = []
    for _ in range(B):
        hf = hflip and random.random() < 0.5
        vf = rot and random.random() < 0.5
        r9 = rot and random.random() < 0.5
        flags.append((hf, vf, r9))

    def _aug(t, f):
        if f[0]:
            t = t.flip(-1)
        if f[1]:
            t = t.flip(-2)
        if f[2]:
            t = t.transpose(-1, -2)
        return t
    return [torch.stack([_aug(t[b], flags[b]) for b in range(B)]) for t in img_list]
