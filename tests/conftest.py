import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + '.npz')))
        return cache[name]
    return load


def checks(t):
    a = t.detach().cpu().numpy().astype(np.float64)
    return np.array([a.sum(), np.abs(a).sum(), np.sqrt((a * a).sum())])


def fp16_psnr_gate(y16, y32, seed=0):
    """The fp16 parity gate of BASELINE.md / north_star ("PSNR within 0.01 dB"), made binding: both outputs are
    mapped affinely from the fp32 output's range onto [0, 1] (the synthetic-weight net leaves ~40 % of its output
    outside [0, 1], which tensor2img's clamp would hide), and the HR target is the fp32 output plus Gaussian noise
    of sigma = 10^(-30/20), i.e. a realistic ~30 dB operating point — there |dPSNR| <= 0.01 dB bounds the fp16 rms
    error at ~1.5e-3 of the range (against an unrelated random target, PSNR ~ 8 dB, it bounded nothing).
    y16, y32: [3, H, W] CPU tensors.  Returns (|PSNR(y16, hr) - PSNR(y32, hr)|, PSNR(y16, y32)) with the reference's
    PSNR (codes/utils/util.py:71-114 via oracle.ref_torch.psnr_sr: uint8 images, 4-pixel crop)."""
    import torch
    from esrganplus_amd import synth
    from oracle import ref_torch as RT
    y16, y32 = y16.float(), y32.float()
    lo, hi = float(y32.min()), float(y32.max())
    n16, n32 = (y16 - lo) / (hi - lo), (y32 - lo) / (hi - lo)
    hr = (n32 + 10 ** (-30 / 20) * synth.normal_like(seed, 'gate.noise', tuple(y32.shape))).clamp(0, 1)
    p32 = RT.psnr_sr(n32, hr)
    assert 28.0 <= p32 <= 32.5, p32                      # the operating point really is ~30 dB
    return abs(RT.psnr_sr(n16, hr) - p32), RT.psnr_sr(n16, n32)
