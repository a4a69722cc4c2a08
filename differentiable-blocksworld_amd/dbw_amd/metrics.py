"""Evaluation metrics of the reference's `quantitative_eval` (src/model/dbw.py:464-493): PSNR from the MSE
(loss.py:28-29) and SSIM with an 11-tap Gaussian window, sigma 1.5, C1 = 0.01^2, C2 = 0.03^2 (loss.py:124-156; the
evaluation uses it WITHOUT padding).  Host-side torch code run once per evaluation -- not part of the hot path.

The reference convolves with the 11x11 outer product of the 1-D Gaussian; the window is separable, so this file filters rows
and columns with the 1-D kernel (22 instead of 121 taps per pixel and statistic).  tests/golden/ssim.npz pins it against the
reference's own SSIMLoss."""
import math

import torch
import torch.nn.functional as F


class AverageMeter:
    """Running average weighted by the batch size (utils/metrics.py:17-35)."""

    def __init__(self):
        self.val = self.sum = self.avg = 0.0
        self.count = 0

    def update(self, val, N=1):
        if isinstance(val, torch.Tensor):
            if val.numel() != 1:
                raise ValueError('AverageMeter takes scalars')
            val = val.item()
        self.val = val
        self.sum += val * N
        self.count += N
        self.avg = self.sum / self.count if self.count else 0.0


def mse2psnr(mse):
    """-10 log10(mse) for images in [0, 1]."""
    return -10.0 * torch.log(mse) / math.log(10.0)


def gaussian_window(size=11, sigma=1.5, device=None, dtype=torch.float32):
    x = torch.arange(size, dtype=torch.float64) - size // 2
    g = torch.exp(-x * x / (2.0 * sigma * sigma))
    return (g / g.sum()).to(dtype=dtype, device=device)


def _blur(x, g, pad):
    """Depthwise separable Gaussian filter of (N,C,H,W)."""
    C, k = x.shape[1], g.numel()
    x = F.conv2d(x, g.view(1, 1, 1, k).expand(C, 1, 1, k), padding=(0, pad), groups=C)
    return F.conv2d(x, g.view(1, 1, k, 1).expand(C, 1, k, 1), padding=(pad, 0), groups=C)


def ssim_map(img1, img2, window_size=11, sigma=1.5, padding=False):
    """Per-pixel SSIM of two (N,C,H,W) images in [0,1]; padding=False keeps only windows that lie inside the image."""
    g = gaussian_window(window_size, sigma, img1.device, img1.dtype)
    pad = window_size // 2 if padding else 0
    mu1, mu2 = _blur(img1, g, pad), _blur(img2, g, pad)
    s11 = _blur(img1 * img1, g, pad) - mu1 * mu1
    s22 = _blur(img2 * img2, g, pad) - mu2 * mu2
    s12 = _blur(img1 * img2, g, pad) - mu1 * mu2
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    return ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s11 + s22 + c2))


def ssim(img1, img2, window_size=11, padding=False):
    """Mean SSIM per image, (N,)."""
    return ssim_map(img1, img2, window_size, padding=padding).flatten(1).mean(1)
