"""Image conversion + PSNR exactly as the reference's validation does it
(codes/utils/util.py:71-95 ``tensor2img``, 107-114 ``calculate_psnr``; codes/train.py:131-148: crop
``scale`` pixels, compare uint8 images).  The numpy functions are the host restatement (pinned by
tests/golden/psnr.npz, metrics.npz); ``device_tensor2img`` / ``device_psnr_ssim`` run the same arithmetic in
csrc/metrics.hip on the tensors' GPU, so that a validation pass over many images reads back four doubles per
image pair instead of two images."""
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
        return None             # util.py:151-156 falls off the end for 2- or 4-channel images: None, as the reference
    raise ValueError('Wrong input image dimensions.')


# ---- the same on the device (csrc/metrics.hip, esr_image_metrics) -------------------------------------
def _metrics_call(sr, hr, crop, y_only, min_max):
    import torch
    from . import _lib as L
    from . import engine as E
    E.require_cuda(sr, 'sr')
    a = sr.detach().squeeze().float().contiguous()
    if a.dim() == 2:
        a = a[None]
    C_, H, W = a.shape
    dev = a.device
    p = L.esr_img_metrics()
    p.sr, p.C, p.H, p.W, p.crop, p.y_only = a.data_ptr(), C_, H, W, int(crop), 1 if y_only else 0
    p.lo, p.hi = float(min_max[0]), float(min_max[1])
    keep = [a]
    img_sr = torch.empty((H, W, C_) if C_ == 3 else (H, W), dtype=torch.uint8, device=dev)
    p.img_sr = img_sr.data_ptr()
    img_hr = out = None
    if hr is not None:
        b = hr.detach().squeeze().float().contiguous()
        if b.dim() == 2:
            b = b[None]
        if b.shape != a.shape or b.device != dev:
            raise ValueError('Input images must have the same dimensions.')
        img_hr = torch.empty_like(img_sr)
        out = torch.empty(4, dtype=torch.float64, device=dev)
        p.hr, p.img_hr, p.out = b.data_ptr(), img_hr.data_ptr(), out.data_ptr()
        keep.append(b)
    if y_only:
        ys = torch.empty((2, H, W), dtype=torch.float64, device=dev)     # unrounded luma (codes/test.py:81-86)
        p.y_sr, p.y_hr = ys[0].data_ptr(), ys[1].data_ptr()
        keep.append(ys)
    k1 = gaussian_window()[5] / gaussian_window()[5].sum()       # the normalised 1-D kernel (outer(k, k) == window)
    for i in range(11):
        p.win[i] = float(k1[i])
    L.check(L.lib().esr_image_metrics(p, E.current_stream()), 'esr_image_metrics')
    return img_sr, img_hr, out, (C_, H, W), keep


def device_tensor2img(t, min_max=(0, 1)):
    """tensor2img on the tensor's GPU: uint8 HWC BGR (or HW) tensor, bit-identical with ``tensor2img``."""
    return _metrics_call(t, None, 0, False, min_max)[0]


def device_psnr_ssim(sr, hr, crop=4, y_only=False, min_max=(0, 1)):
    """(PSNR, SSIM) of two (C,H,W) tensors exactly as the validation loops compute them (tensor2img both, crop
    `crop` pixels per side), evaluated on the device; one small read-back.  y_only: PSNR_Y / SSIM_Y of
    codes/test.py:81-90 — ``bgr2ycbcr(img / 255., only_y=True)`` on the FLOAT images, i.e. unrounded luma."""
    _, _, out, (C_, H, W), keep = _metrics_call(sr, hr, crop, y_only, min_max)
    o = out.cpu().numpy()                       # synchronises with the metric kernels
    planes = 1 if (y_only or C_ == 1) else C_
    ch, cw = H - 2 * crop, W - 2 * crop
    mse = o[0] / (ch * cw * planes)
    psnr = float('inf') if mse == 0 else 20 * math.log10(255.0 / math.sqrt(mse))
    ssim = float(o[2] / ((ch - 10) * (cw - 10) * planes)) if (ch > 10 and cw > 10) else float('nan')
    return psnr, ssim
