"""test_image/test.py:26-40 end to end: tools/sr_infer.py (the reference script's statements with PIL for cv2 and the
drop-in RRDB_Net) over ALL five bundled LR images — 128x128, 72x72, 64x64, 70x70 and the non-square 57x86 — against the
uint8 images the imported reference writes for the same synthetic nb = 23 weights (tests/golden/sr_infer.npz,
oracle/gen_golden.py: gen_sr_infer)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('prec', ['fp32', 'fp16'])
def test_sr_infer_script_over_the_five_bundled_images(tmp_path, golden, prec):
    from PIL import Image
    g = golden('sr_infer')
    names = [str(n) for n in g['names']]
    assert names == ['baby', 'bird', 'butterfly', 'head', 'woman']
    in_dir, out_dir = tmp_path / 'LR', tmp_path / 'results'
    in_dir.mkdir()
    for n in names:
        Image.fromarray(g['lr_' + n]).save(str(in_dir / (n + '.png')))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'sr_infer.py'), 'synthetic', str(in_dir), str(out_dir), prec],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    for n in names:
        got = np.array(Image.open(str(out_dir / (n + '_rlt.png'))).convert('RGB')).astype(np.int32)
        ref = g['sr_' + n].astype(np.int32)
        assert got.shape == ref.shape == (4 * g['lr_' + n].shape[0], 4 * g['lr_' + n].shape[1], 3)
        d = np.abs(got - ref)
        mse = float((d.astype(np.float64) ** 2).mean())
        psnr = 99.0 if mse == 0 else 20 * np.log10(255.0 / np.sqrt(mse))
        print('%-10s %s  %s: max|diff| %d LSB, differing pixels %.4f %%, PSNR %.1f dB'
              % (n, got.shape[:2], prec, d.max(), 100 * np.mean(d > 0), psnr))
        if prec == 'fp32':
            # +-1 LSB where a value sits on a rounding boundary (fp32 summation order), nothing else
            assert d.max() <= 1 and np.mean(d > 0) <= 2e-3, (n, d.max(), np.mean(d > 0))
        else:
            # fp16 storage: rms error ~1e-3 of the range -> a fraction of an LSB; >= 50 dB between the two uint8 images
            assert d.max() <= 4 and psnr >= 50.0, (n, d.max(), psnr)
