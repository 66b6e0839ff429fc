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


# ---- SSIM and Y-channel conversion (codes/utils/util.py:117-158, codes/data/util.py:123-168) -----------
_Y_RGB = np.array([65.481, 128.553, 24.966])
_YCC_RGB = np.array([[65.481, -37.797, 112.0], [128.553, -74.203, -93.786], [24.966, 112.0, -18.214]])


def _ycbcr(img, coeff_y, coeff_full, only_y):
    """MATLAB-style conversion for uint8 [0,255] (rounded) or float [0,1] HWC images.  Unlike the
    reference (data/util.py:132-134) the input array is not scaled in place."""
    is_u8 = img.dtype == np.uint8
    x = img.astype(np.float64) if is_u8 else img.astype(np.float64) * 255.0
    if only_y:
        r = np.dot(x, coeff_y) / 255.0 + 16.0
    else:
        r = np.matmul(x, coeff_full) / 255.0 + np.array([16, 128, 128])
    r = r.round() if is_u8 else r / 255.0
    return r.astype(img.dtype)


def rgb2ycbcr(img, only_y=True):
    return _ycbcr(img, _Y_RGB, _YCC_RGB, only_y)


def bgr2ycbcr(img, only_y=True):
    return _ycbcr(img, _Y_RGB[::-1], _YCC_RGB[::-1], only_y)


def gaussian_window(ksize=11, sigma=1.5):
    """cv2.getGaussianKernel(11, 1.5) outer itself: exp(-(i-c)^2 / (2 sigma^2)), normalised."""
    i = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2.0
    k = np.exp(-(i * i) / (2.0 * sigma * sigma))
    k /= k.sum()
    return np.outer(k, k)


def _ssim_plane(a, b):
    """util.py:117-137: 11x11 Gaussian-window SSIM over the valid region of two [0,255] planes.
    The filtering runs as a conv2d on whatever device the planes live on (float64)."""
    import torch
    import torch.nn.functional as F
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64, device=a.device)
    w = torch.as_tensor(gaussian_window(), dtype=torch.float64, device=a.device)[None, None]
    stack = torch.stack([a, b, a * a, b * b, a * b])[:, None]           # 5 x 1 x H x W
    f = F.conv2d(stack, w)[:, 0]                                          # 'valid' == filter2D(...)[5:-5, 5:-5]
    mu1, mu2 = f[0], f[1]
    s1, s2, s12 = f[2] - mu1 * mu1, f[3] - mu2 * mu2, f[4] - mu1 * mu2
    m = ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s1 + s2 + c2))
    return float(m.mean())


def calculate_ssim(img1, img2):
    """util.py:140-158, including its quirk: for 3-channel inputs the reference averages three calls
    on the FULL (H, W, 3) arrays — cv2.filter2D filters each channel, so that equals the mean over
    channels of the per-channel SSIM maps."""
    if img1.shape != img2.shape:
        raise ValueError('Input images must have the same dimensions.')
    if img1.ndim == 2:
        return _ssim_plane(img1, img2)
    if img1.ndim == 3:
        if img1.shape[2] == 3:
            return float(np.mean([_ssim_plane(img1[..., c], img2[..., c]) for c in range(3)]))
        if img1.shape[2] == 1:
            return _ssim_plane(np.squeeze(img1), np.squeeze(img2))
    raise ValueError('Wrong input image dimensions.')
