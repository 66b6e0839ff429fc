"""Device-side validation metrics (csrc/metrics.hip) against the reference's own outputs
(tests/golden/psnr.npz: tensor2img images and PSNR values produced by the imported codes/utils/util.py) and
against the host restatement in esrganplus_amd.metrics (itself pinned by tests/test_host.py / test_metrics.py)."""
import numpy as np
import pytest
import torch

from esrganplus_amd import metrics as M
from esrganplus_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def test_device_tensor2img_and_psnr_match_reference_goldens(dev):
    g = dict(np.load('tests/golden/psnr.npz'))
    n = sum(1 for k in g if k.startswith('img_a'))
    assert n > 0
    for i in range(n):
        a, b = torch.from_numpy(g['a%d' % i]).to(dev), torch.from_numpy(g['b%d' % i]).to(dev)
        assert np.array_equal(M.device_tensor2img(a).cpu().numpy(), g['img_a%d' % i])     # bit-exact uint8 BGR
        psnr, _ = M.device_psnr_ssim(a, b, crop=4)
        assert abs(psnr - float(g['psnr%d' % i])) < 1e-9


@pytest.mark.parametrize('shape,crop', [((3, 40, 52), 4), ((3, 33, 47), 0), ((1, 24, 30), 2), ((3, 128, 96), 4)])
@pytest.mark.parametrize('y_only', [False, True])
def test_device_psnr_ssim_match_host_metrics(dev, shape, crop, y_only):
    if y_only and shape[0] != 3:
        pytest.skip('Y channel needs 3 channels')
    hr = synth.image_batch(21, 1, *shape, name='met.hr')[0]
    # out-of-range values exercise the clamp; .5/255 steps exercise round-half-to-even
    sr = (hr + 0.08 * synth.normal_like(22, 'met.n', shape)).mul(255).round().div(255) + 0.5 / 255
    sr[0, 0, :3] = torch.tensor([-0.2, 1.3, 0.5])
    a, b = M.tensor2img(sr), M.tensor2img(hr)
    assert np.array_equal(M.device_tensor2img(sr.to(dev)).cpu().numpy(), a)
    if a.ndim == 2:
        a, b = a[..., None], b[..., None]
    ca = a[crop:a.shape[0] - crop, crop:a.shape[1] - crop]
    cb = b[crop:b.shape[0] - crop, crop:b.shape[1] - crop]
    if y_only:      # codes/test.py:81-86: the conversion runs on the float images (/255.), luma stays unrounded
        ca, cb = M.bgr2ycbcr(ca / 255., only_y=True) * 255, M.bgr2ycbcr(cb / 255., only_y=True) * 255
    want_psnr = M.calculate_psnr(ca, cb)
    want_ssim = M.calculate_ssim(np.squeeze(ca) if ca.shape[-1] == 1 else ca, np.squeeze(cb) if cb.shape[-1] == 1 else cb)
    psnr, ssim = M.device_psnr_ssim(sr.to(dev), hr.to(dev), crop=crop, y_only=y_only)
    assert abs(psnr - want_psnr) < 1e-9
    assert abs(ssim - want_ssim) < 1e-9


def test_identical_images_give_inf_psnr_and_unit_ssim(dev):
    x = synth.image_batch(23, 1, 3, 32, 32, name='met.same')[0].to(dev)
    psnr, ssim = M.device_psnr_ssim(x, x.clone(), crop=4)
    assert psnr == float('inf') and abs(ssim - 1.0) < 1e-12


def test_device_psnr_y_matches_reference_test_script(dev):
    """PSNR_Y under the golden captured from the reference's own flow (codes/test.py:69-90 with the imported
    utils/util.py + data/util.py; oracle/gen_golden.py: gen_metrics_y): the Y conversion runs on the FLOAT images,
    so the luma is not rounded to uint8."""
    g = dict(np.load('tests/golden/metrics_y.npz'))
    for i in range(3):
        h, w = (int(v) for v in g['shape%d' % i])
        crop = int(g['crop%d' % i])
        hr = synth.image_batch(80 + i, 1, 3, h, w, name='mety.hr')[0]
        sr = hr + 0.06 * synth.normal_like(80 + i, 'mety.n', (3, h, w))
        psnr, _ = M.device_psnr_ssim(sr.to(dev), hr.to(dev), crop=crop)
        psnr_y, ssim_y = M.device_psnr_ssim(sr.to(dev), hr.to(dev), crop=crop, y_only=True)
        assert abs(psnr - float(g['psnr%d' % i])) < 1e-9
        assert abs(psnr_y - float(g['psnr_y%d' % i])) < 1e-7, (psnr_y, float(g['psnr_y%d' % i]))
        assert 0.0 < ssim_y <= 1.0


def test_device_ssim_matches_reference_lines(dev):
    """SSIM / SSIM_Y on the device against the outputs of the reference's own ``calculate_ssim`` (codes/utils/util.py:
    117-158 over oracle/ref_import.cv2_shim; tests/golden/ssim.npz) in the validation flow of codes/test.py:69-90."""
    g = dict(np.load('tests/golden/ssim.npz'))
    for i in range(4):
        h, w = (int(v) for v in g['shape%d' % i])
        crop = int(g['crop%d' % i])
        hr = synth.image_batch(90 + i, 1, 3, h, w, name='ssim.hr')[0]
        sr = hr + 0.06 * synth.normal_like(90 + i, 'ssim.n', (3, h, w))
        _, ssim = M.device_psnr_ssim(sr.to(dev), hr.to(dev), crop=crop)
        _, ssim_y = M.device_psnr_ssim(sr.to(dev), hr.to(dev), crop=crop, y_only=True)
        assert abs(ssim - float(g['ssim%d' % i])) < 1e-9, (i, ssim, float(g['ssim%d' % i]))
        assert abs(ssim_y - float(g['ssim_y%d' % i])) < 1e-9, (i, ssim_y, float(g['ssim_y%d' % i]))
    g_hr = synth.image_batch(95, 1, 1, 24, 30, name='ssim.g')[0]
    g_sr = g_hr + 0.05 * synth.normal_like(95, 'ssim.gn', (1, 24, 30))
    _, ssim = M.device_psnr_ssim(g_sr.to(dev), g_hr.to(dev), crop=2)
    assert abs(ssim - float(g['grey'])) < 1e-9
