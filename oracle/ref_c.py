"""ORACLE (test infrastructure) — numpy/ctypes driver for the plain-C restatement
``oracle/conv_ref.c``.  Independent of torch: used to cross-check ``oracle/ref_torch.py``
(i.e. that the reference's ``nn.Conv2d`` & co. mean what we think they mean).  Never imported by
the product.  Structure of the network follows block.py:260-291 and architecture.py:47-78."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None
_f = ctypes.POINTER(ctypes.c_float)


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE])


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(_HERE, 'libconv_ref.so')
        if not os.path.exists(so):
            build()
        _lib = ctypes.CDLL(so)
    return _lib


def _p(a):
    return a.ctypes.data_as(_f) if a is not None else None


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def conv2d(x, w, b=None, stride=1):
    x, w = _c(x), _c(w)
    b = _c(b) if b is not None else None
    n, c, h, wd = x.shape
    k, _, ks, _ = w.shape
    pad = (ks - 1) // 2
    ho, wo = (h + 2 * pad - ks) // stride + 1, (wd + 2 * pad - ks) // stride + 1
    y = np.empty((n, k, ho, wo), np.float32)
    lib().ref_conv2d(_p(x), _p(w), _p(b), _p(y), n, c, h, wd, k, ks, stride, pad)
    return y


def lrelu(x):
    x = _c(x).copy()
    lib().ref_leaky_relu(_p(x), ctypes.c_size_t(x.size), ctypes.c_float(0.2))
    return x


def upsample2(x):
    x = _c(x)
    n, c, h, w = x.shape
    y = np.empty((n, c, 2 * h, 2 * w), np.float32)
    lib().ref_upsample2(_p(x), _p(y), n * c, h, w)
    return y


def axpb(a, alpha, b):
    a, b = _c(a), _c(b)
    y = np.empty_like(a)
    lib().ref_axpb(_p(a), ctypes.c_float(alpha), _p(b), _p(y), ctypes.c_size_t(a.size))
    return y


def noise(x, z, sigma=0.1):
    if z is None:
        return x
    x = _c(x).copy()
    lib().ref_noise(_p(x), _p(_c(z)), ctypes.c_float(sigma), ctypes.c_size_t(x.size))
    return x


def batchnorm(x, gamma, beta, rmean, rvar, training, momentum=0.1, eps=1e-5):
    x = _c(x).copy()
    n, c, h, w = x.shape
    lib().ref_batchnorm(_p(x), n, c, h * w, _p(_c(gamma)), _p(_c(beta)), _p(rmean), _p(rvar),
                        int(training), ctypes.c_float(momentum), ctypes.c_float(eps))
    return x


def maxpool2(x):
    x = _c(x)
    n, c, h, w = x.shape
    y = np.empty((n, c, h // 2, w // 2), np.float32)
    lib().ref_maxpool2(_p(x), _p(y), n * c, h, w)
    return y


def linear(x, w, b):
    x, w, b = _c(x), _c(w), _c(b)
    y = np.empty((x.shape[0], w.shape[0]), np.float32)
    lib().ref_linear(_p(x), _p(w), _p(b), _p(y), x.shape[0], x.shape[1], w.shape[0])
    return y


def rdb_forward(x, sd, p, z=None):
    """block.py:260-268."""
    g = lambda k: sd[k].numpy() if hasattr(sd[k], 'numpy') else sd[k]   # noqa: E731
    cat = lambda *a: np.concatenate(a, 1)                                # noqa: E731
    x1 = lrelu(conv2d(x, g(p + '.conv1.0.weight'), g(p + '.conv1.0.bias')))
    x2 = lrelu(conv2d(cat(x, x1), g(p + '.conv2.0.weight'), g(p + '.conv2.0.bias')))
    x2 = x2 + conv2d(x, g(p + '.conv1x1.weight'))
    x3 = lrelu(conv2d(cat(x, x1, x2), g(p + '.conv3.0.weight'), g(p + '.conv3.0.bias')))
    x4 = lrelu(conv2d(cat(x, x1, x2, x3), g(p + '.conv4.0.weight'), g(p + '.conv4.0.bias')))
    x4 = x4 + x2
    x5 = conv2d(cat(x, x1, x2, x3, x4), g(p + '.conv5.0.weight'), g(p + '.conv5.0.bias'))
    return noise(axpb(x5, 0.2, x), z)


def rrdbnet_forward(x, sd, nb, z=None, variant='codes'):
    """architecture.py:47-78; block.py:287-291; test_image/block.py:250-256."""
    g = lambda k: sd[k].numpy() if hasattr(sd[k], 'numpy') else sd[k]   # noqa: E731
    per = 3 if variant == 'codes' else 4
    fea = conv2d(x, g('model.0.weight'), g('model.0.bias'))
    t = fea
    for i in range(nb):
        zi = [None] * 4 if z is None else list(z[per * i: per * i + per]) + [None]
        o = t
        for j in (1, 2, 3):
            o = rdb_forward(o, sd, 'model.1.sub.%d.RDB%d' % (i, j), zi[j - 1])
        t = noise(axpb(o, 0.2, t), zi[3] if per == 4 else None)
    t = conv2d(t, g('model.1.sub.%d.weight' % nb), g('model.1.sub.%d.bias' % nb))
    t = fea + t
    for k in ('model.3', 'model.6'):
        t = lrelu(conv2d(upsample2(t), g(k + '.weight'), g(k + '.bias')))
    t = lrelu(conv2d(t, g('model.8.weight'), g('model.8.bias')))
    return conv2d(t, g('model.10.weight'), g('model.10.bias'))
