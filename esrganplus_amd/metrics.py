"""Image conversion + PSNR exactly as the reference's validation does it
(codes/utils/util.py:71-95 ``tensor2img``, 107-114 ``calculate_psnr``; codes/train.py:131-148: crop
``scale`` pixels, compare uint8 images).  numpy only — host-side harness code, not on the hot path."""
import math

import numpy as np


def tensor2img(t, min_max=(0, 1)):
    """3-D (C,H,W) RGB tensor -> uint8 HWC **BGR** image, clamped and rounded like the reference."""
    a = t.detach().squeeze().float().cpu().clamp(*min_max).numpy()
    a = (a - min_max[0]) / (min_max[1] - min_max[0])
    if a.ndim == 3:
        a = np.transpose(a[[2, 1, 0], :, :], (1, 2, 0))
    return (a * 255.0).round().astype(np.uint8)


def calculate_psnr(img1, img2):
    mse = np.mean((img1.astype(np.float64) - img2.astype(np.float64)) ** 2)
    if mse == 0:
        return float('inf')
    return 20 * math.log10(255.0 / math.sqrt(mse))


def validation_psnr(sr, hr, scale=4):
    a, b = tensor2img(sr) / 255., tensor2img(hr) / 255.
    a, b = a[scale:-scale, scale:-scale, :], b[scale:-scale, scale:-scale, :]
    return calculate_psnr(a * 255, b * 255)
