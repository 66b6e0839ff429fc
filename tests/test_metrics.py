"""Evaluation metrics: Y-channel conversion against the imported reference's outputs
(tests/golden/metrics.npz); SSIM against tests/golden/ssim.npz = the outputs of the reference's own ``ssim`` /
``calculate_ssim`` (codes/utils/util.py:117-158) run over a two-function cv2 shim (oracle/ref_import.py: cv2_shim;
cv2 itself is not installed), plus a scipy restatement of `cv2.filter2D(...)[5:-5, 5:-5]` as a second opinion."""
import numpy as np
import pytest

from esrganplus_amd import metrics as M


def test_ycbcr_matches_reference():
    g = dict(np.load('tests/golden/metrics.npz'))
    for name, fn in (('bgr', M.bgr2ycbcr), ('rgb', M.rgb2ycbcr)):
        for only_y in (True, False):
            assert np.array_equal(fn(g['u8'], only_y), g['%s_u8_%d' % (name, only_y)])
            assert np.abs(fn(g['fl'], only_y) - g['%s_fl_%d' % (name, only_y)]).max() <= 1e-6
    before = g['fl'].copy()
    M.bgr2ycbcr(g['fl'])
    assert np.array_equal(before, g['fl'])            # no in-place scaling of the caller's array


def _ssim_scipy(a, b):
    from scipy.ndimage import correlate
    w = M.gaussian_window()

    def f(x):
        return correlate(x, w, mode='mirror')[5:-5, 5:-5]
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    mu1, mu2 = f(a), f(b)
    s1, s2, s12 = f(a * a) - mu1 ** 2, f(b * b) - mu2 ** 2, f(a * b) - mu1 * mu2
    return (((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 ** 2 + mu2 ** 2 + c1) * (s1 + s2 + c2))).mean()


def test_ssim_matches_filter2d_restatement():
    rng = np.random.RandomState(3)
    a = rng.rand(40, 33) * 255
    b = np.clip(a + rng.randn(40, 33) * 12, 0, 255)
    assert abs(M.calculate_ssim(a, b) - _ssim_scipy(a, b)) < 1e-12
    assert abs(M.calculate_ssim(a, a) - 1.0) < 1e-12
    a3, b3 = np.stack([a, a * 0.5, 255 - a], -1), np.stack([b, b * 0.5, 255 - b], -1)
    want = np.mean([_ssim_scipy(a3[..., c], b3[..., c]) for c in range(3)])
    assert abs(M.calculate_ssim(a3, b3) - want) < 1e-12
    with pytest.raises(ValueError):
        M.calculate_ssim(a, a[:-1])


def test_gaussian_window_is_cv2_formula():
    w = M.gaussian_window()
    assert w.shape == (11, 11) and abs(w.sum() - 1.0) < 1e-15
    k = w.sum(0)
    assert abs(k[5] / k[4] - np.exp(1.0 / (2 * 1.5 ** 2))) < 1e-12


def test_host_psnr_y_flow_matches_reference_test_script():
    """The host restatement of codes/test.py:69-90 (tensor2img, /255, bgr2ycbcr on the float image, crop, x255,
    PSNR) against the golden produced by the imported reference (oracle/gen_golden.py: gen_metrics_y)."""
    from esrganplus_amd import synth
    g = dict(np.load('tests/golden/metrics_y.npz'))
    for i in range(3):
        h, w = (int(v) for v in g['shape%d' % i])
        crop = int(g['crop%d' % i])
        hr = synth.image_batch(80 + i, 1, 3, h, w, name='mety.hr')[0]
        sr = hr + 0.06 * synth.normal_like(80 + i, 'mety.n', (3, h, w))
        a, b = M.tensor2img(sr) / 255., M.tensor2img(hr) / 255.
        ya, yb = M.bgr2ycbcr(a, only_y=True), M.bgr2ycbcr(b, only_y=True)
        if i == 0:
            assert np.abs(ya * 255 - g['y_sr0']).max() < 1e-9
        got = M.calculate_psnr(ya[crop:-crop, crop:-crop] * 255, yb[crop:-crop, crop:-crop] * 255)
        assert abs(got - float(g['psnr_y%d' % i])) < 1e-9


def test_host_ssim_matches_reference_lines():
    """tests/golden/ssim.npz: produced by the imported codes/utils/util.py (gen_golden.py: gen_ssim) — the validation
    flow of codes/test.py:69-90 on 3 channels and on the unrounded Y plane, grey images, raw float arrays."""
    from esrganplus_amd import synth
    g = dict(np.load('tests/golden/ssim.npz'))
    for i in range(4):
        h, w = (int(v) for v in g['shape%d' % i])
        crop = int(g['crop%d' % i])
        hr = synth.image_batch(90 + i, 1, 3, h, w, name='ssim.hr')[0]
        sr = hr + 0.06 * synth.normal_like(90 + i, 'ssim.n', (3, h, w))
        a, b = M.tensor2img(sr) / 255., M.tensor2img(hr) / 255.
        sl = slice(crop, -crop) if crop else slice(None)
        assert abs(M.calculate_ssim(a[sl, sl, :] * 255, b[sl, sl, :] * 255) - float(g['ssim%d' % i])) < 1e-12
        ya, yb = M.bgr2ycbcr(a, only_y=True), M.bgr2ycbcr(b, only_y=True)
        assert abs(M.calculate_ssim(ya[sl, sl] * 255, yb[sl, sl] * 255) - float(g['ssim_y%d' % i])) < 1e-12
    g_hr = synth.image_batch(95, 1, 1, 24, 30, name='ssim.g')[0]
    g_sr = g_hr + 0.05 * synth.normal_like(95, 'ssim.gn', (1, 24, 30))
    ia, ib = M.tensor2img(g_sr), M.tensor2img(g_hr)
    assert abs(M.calculate_ssim(ia[2:-2, 2:-2].astype(np.float64), ib[2:-2, 2:-2].astype(np.float64)) - float(g['grey'])) < 1e-12
    assert abs(M.calculate_ssim(ia[..., None], ib[..., None]) - float(g['grey_hw1'])) < 1e-12
    a, b = g['raw_a'], g['raw_b']
    a3, b3 = np.stack([a, a * 0.5, 255 - a], -1), np.stack([b, b * 0.5, 255 - b], -1)
    assert abs(M.calculate_ssim(a, b) - float(g['raw_ssim'])) < 1e-12
    assert abs(M.calculate_ssim(a3, b3) - float(g['raw_ssim3'])) < 1e-12
    assert abs(M.calculate_ssim(a, a) - float(g['raw_same'])) < 1e-12
    assert M.calculate_ssim(a3[..., :2], b3[..., :2]) is None        # the reference falls off the end here (util.py:151-156)
    with pytest.raises(ValueError):
        M.calculate_ssim(a[None, None], b[None, None])
