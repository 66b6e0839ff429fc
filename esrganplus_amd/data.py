"""Data path on the GPU: what ``LRHRDataset.__getitem__`` does per sample on a CPU
worker (codes/data/LRHR_dataset.py:80-112, codes/data/util.py:94-106,213-343) done per BATCH on the
device, so eight GPUs are not fed by eight Python processes looping over image rows.

* ``imresize`` — MATLAB-compatible bicubic resize with antialiasing (util.py:276-343 /
  ``imresize_np`` 345-412): output size ceil(in*scale), cubic kernel of width 4 (4/scale when
  shrinking), weights normalised per output sample, symmetric border, H pass then W pass, float32.
  The tiny weight / index tables are derived on the host in float64 from the definition of the resize
  (``resample_tables``) and pinned to the reference's outputs by fixtures (<= 2e-6); the two gather passes are
  HIP launches (``esr_resample_axis``).
* ``paired_random_crop`` / ``augment`` — the crop + flip/rot logic on NCHW device tensors, one window / one
  set of coin flips per sample.  ``crop_and_augment`` draws from Python's ``random`` in the reference's
  per-item order (randint, randint, then the three coin flips, sample by sample: LRHR_dataset.py:96-110);
  the two separate helpers draw all windows first, then all flips."""
import ctypes as C
import math
import random

import torch

from . import _lib as L
from . import engine as E


def _keys_kernel(t):
    """Keys' piecewise-cubic interpolation kernel with a = -1/2 (what MATLAB's bicubic `imresize` and the
    reference's `cubic`, util.py:213-218, evaluate): support [-2, 2], C1, reproduces quadratics.
    Horner form on |t|, float64."""
    import numpy as np
    a = np.abs(np.asarray(t, dtype=np.float64))
    near = (1.5 * a - 2.5) * a * a + 1.0                     # |t| <= 1
    far = ((-0.5 * a + 2.5) * a - 4.0) * a + 2.0             # 1 < |t| <= 2
    return np.where(a <= 1.0, near, np.where(a <= 2.0, far, 0.0))


def resample_tables(in_length, scale, antialiasing=True):
    """Weights and SOURCE indices of one axis of the MATLAB-style resize (the function util.py:221-343
    computes; derived here from its definition, in float64, not from the reference's float32 tensor code):

      * output sample o (0-based) sits at input coordinate  c(o) = (o + 0.5) / scale - 0.5   (pixel centres
        of the two grids coincide at the image borders);
      * the footprint is Keys' kernel stretched by s = 1/scale when shrinking with antialiasing (a low-pass of
        width 4/scale), unstretched otherwise:  w(o, i) = k((c(o) - i) / s) / s, then normalised per o;
      * taps run over the integer positions within half a footprint of c(o); positions outside the image are
        mirrored about the border pixel EDGES (period 2 n: -1 -> 0, n -> n - 1);
      * columns that are zero for every output sample are dropped, so the table is [out, P] with P the
        widest real support.
    -> (weights [out, P] float32, src [out, P] int32, out_len)"""
    import numpy as np
    n = int(in_length)
    out_length = math.ceil(n * scale)
    stretch = 1.0 / scale if (scale < 1 and antialiasing) else 1.0
    half = 2.0 * stretch                                          # half width of the footprint
    o = np.arange(out_length, dtype=np.float64)
    centre = (o + 0.5) / scale - 0.5
    first = np.floor(centre - half).astype(np.int64) + 1          # first integer position that can fall inside
    ntap = int(math.ceil(2.0 * half)) + 1
    pos = first[:, None] + np.arange(ntap, dtype=np.int64)[None, :]
    w = _keys_kernel((centre[:, None] - pos) / stretch) / stretch
    w /= w.sum(axis=1, keepdims=True)
    live = np.nonzero(np.any(w != 0.0, axis=0))[0]                # trim all-zero columns at either end
    pos, w = pos[:, live[0]:live[-1] + 1], w[:, live[0]:live[-1] + 1]
    m = np.mod(pos, 2 * n)                                        # mirror: period 2n
    src = np.where(m < n, m, 2 * n - 1 - m)
    assert src.min() >= 0 and src.max() < n
    return (torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32)),
            torch.from_numpy(np.ascontiguousarray(src, dtype=np.int32)), out_length)


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
    (two ``random.randint`` calls each, in sample order); a batch does the same — B independent windows.  The
    stream of Python's ``random`` is consumed as B consecutive crops would consume it; use ``crop_and_augment``
    for the reference's interleaved crop / flip order.  A 3-D tensor is one sample."""
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
    drawn per SAMPLE, applied to the same sample of every tensor in ``img_list`` (``crop_and_augment`` keeps the
    reference's crop-then-flip draw order per item);
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


def crop_and_augment(lr, hr, lr_size, scale, hflip=True, rot=True):
    """One training batch exactly as B consecutive ``LRHRDataset.__getitem__`` calls would cut it
    (LRHR_dataset.py:96-110): per sample, in order, the crop window (``randint`` for the row, ``randint`` for the
    column) and then the flip / transpose coins (util.py:94-106; a disabled option draws nothing) — so a seeded
    run picks the reference's windows and flips.  lr / hr: NCHW batches (or one CHW sample)."""
    single = lr.dim() == 3
    if single:
        lr, hr = lr[None], hr[None]
    outl, outh = [], []
    for b in range(lr.shape[0]):
        l, h = paired_random_crop(lr[b], hr[b], lr_size, scale)
        l, h = augment([l, h], hflip, rot)
        outl.append(l)
        outh.append(h)
    if single:
        return outl[0], outh[0]
    return torch.stack(outl), torch.stack(outh)
