"""Data path: MATLAB-style bicubic imresize + augment against fixtures captured from
the imported reference (oracle/gen_golden.py: gen_imresize; codes/data/util.py:94-106,213-412)."""
import random

import numpy as np
import pytest
import torch

from esrganplus_amd import synth

CASES = 7


@pytest.fixture(scope='module')
def g():
    return dict(np.load('tests/golden/imresize.npz'))


def _case(g, i):
    h, w = (int(v) for v in g['shape%d' % i])
    return synth.image_batch(60 + i, 1, 3, h, w, name='imresize.x')[0], float(g['scale%d' % i]), g['y%d' % i]


def _cpu_resample(x, scale, aa=True):
    """test-side evaluation of the product's host tables (the product applies them with a HIP kernel)."""
    from esrganplus_amd import data as D
    wh, ih, _ = D.resample_tables(x.shape[1], scale, aa)
    ww, iw, _ = D.resample_tables(x.shape[2], scale, aa)
    y = (x[:, ih.long(), :] * wh[None, :, :, None]).sum(2)
    return (y[:, :, iw.long()] * ww[None, None, :, :]).sum(3)


@pytest.mark.parametrize('i', range(CASES))
def test_resample_tables_reproduce_reference_imresize(g, i):
    x, sc, want = _case(g, i)
    got = _cpu_resample(x, sc).numpy()
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 2e-6


def test_resample_without_antialiasing(g):
    x = synth.image_batch(70, 1, 3, 6, 8, name='imresize.x')[0]
    assert np.abs(_cpu_resample(x, 0.5, False).numpy() - g['y_noaa']).max() <= 2e-6


def test_augment_matches_reference_draw_order(g):
    from esrganplus_amd import data as D
    a = torch.arange(2 * 3 * 4, dtype=torch.float32).reshape(3, 4, 2).permute(2, 0, 1)      # HWC -> CHW
    for seed in range(12):
        random.seed(seed)
        o = D.augment([a.clone()], True, True)[0]
        assert np.array_equal(o.permute(1, 2, 0).contiguous().numpy().reshape(-1), g['aug_out'][seed]), seed


def test_paired_random_crop_windows():
    from esrganplus_amd import data as D
    lr = torch.arange(40 * 50, dtype=torch.float32).reshape(1, 1, 40, 50)
    hr = torch.arange(160 * 200, dtype=torch.float32).reshape(1, 1, 160, 200)
    random.seed(3)
    rh, rw = random.randint(0, 40 - 32), random.randint(0, 50 - 32)
    random.seed(3)
    l, h = D.paired_random_crop(lr, hr, 32, 4)
    assert l.shape[-2:] == (32, 32) and h.shape[-2:] == (128, 128)
    assert float(l[0, 0, 0, 0]) == rh * 50 + rw and float(h[0, 0, 0, 0]) == 4 * rh * 200 + 4 * rw


@pytest.mark.gpu
@pytest.mark.parametrize('i', range(CASES))
def test_gpu_imresize_matches_reference(g, i):
    from esrganplus_amd import data as D
    x, sc, want = _case(g, i)
    dev = torch.device('cuda:0')
    got = D.imresize(x.to(dev), sc).cpu().numpy()
    assert got.shape == want.shape and np.abs(got - want).max() <= 2e-6
    # batched NCHW goes through the same launches
    xb = torch.stack([x, x.flip(-1)]).to(dev)
    yb = D.imresize(xb, sc).cpu().numpy()
    assert np.abs(yb[0] - want).max() <= 2e-6


@pytest.mark.gpu
def test_gpu_imresize_rejects_cpu_tensors():
    from esrganplus_amd import data as D
    with pytest.raises(Exception):
        D.imresize(torch.zeros(3, 8, 8), 0.5)


def test_crop_and_augment_draw_per_sample():
    """A batch consumes Python's `random` stream exactly like B dataset items (LRHR_dataset.__getitem__ draws
    per sample): windows / flips differ between the samples of a batch and equal the per-item results."""
    from esrganplus_amd import data as D
    B = 6
    lr = torch.arange(B * 40 * 48, dtype=torch.float32).reshape(B, 1, 40, 48)
    hr = torch.arange(B * 160 * 192, dtype=torch.float32).reshape(B, 1, 160, 192)
    random.seed(11)
    lb, hb = D.paired_random_crop(lr, hr, 32, 4)
    random.seed(11)
    per = [D.paired_random_crop(lr[b], hr[b], 32, 4) for b in range(B)]
    assert all(torch.equal(lb[b], per[b][0]) and torch.equal(hb[b], per[b][1]) for b in range(B))
    offs = {(float(lb[b, 0, 0, 0]) - b * 40 * 48) for b in range(B)}
    assert len(offs) > 1                      # not one shared window
    random.seed(5)
    ab = D.augment([lb, hb], True, True)
    random.seed(5)
    pa = [D.augment([lb[b], hb[b]], True, True) for b in range(B)]
    assert all(torch.equal(ab[0][b], pa[b][0]) and torch.equal(ab[1][b], pa[b][1]) for b in range(B))


def test_crop_and_augment_interleaves_draws_like_dataset_items():
    """crop_and_augment consumes `random` in the reference's per-item order (LRHR_dataset.py:96-110: randint,
    randint, then util.augment's three coins, sample after sample) — restated here with plain slicing."""
    from esrganplus_amd import data as D
    B, S = 5, 4
    lr = torch.arange(B * 2 * 20 * 24, dtype=torch.float32).reshape(B, 2, 20, 24)
    hr = torch.arange(B * 2 * 80 * 96, dtype=torch.float32).reshape(B, 2, 80, 96)
    random.seed(21)
    gl, gh = D.crop_and_augment(lr, hr, 16, S)
    random.seed(21)
    for b in range(B):
        rh, rw = random.randint(0, 20 - 16), random.randint(0, 24 - 16)
        hf, vf, r9 = random.random() < 0.5, random.random() < 0.5, random.random() < 0.5
        l = lr[b, :, rh:rh + 16, rw:rw + 16]
        h = hr[b, :, S * rh:S * rh + 64, S * rw:S * rw + 64]
        for flag, fn in ((hf, lambda t: t.flip(-1)), (vf, lambda t: t.flip(-2)), (r9, lambda t: t.transpose(-1, -2))):
            if flag:
                l, h = fn(l), fn(h)
        assert torch.equal(gl[b], l) and torch.equal(gh[b], h), b
