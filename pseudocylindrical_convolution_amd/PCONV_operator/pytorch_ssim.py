"""Gaussian-window SSIM (reference: PCONV_operator/pytorch_ssim.py:8-72):
11-tap sigma-1.5 window, per-channel depthwise filtering, C1 = 0.01^2, C2 = 0.03^2."""
import math

import torch
import torch.nn.functional as F


def gaussian(window_size, sigma):
    g = torch.Tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    return g / g.sum()


def create_window(window_size, channel):
    g1 = gaussian(window_size, 1.5).unsqueeze(1)
    g2 = g1.mm(g1.t()).float().unsqueeze(0).unsqueeze(0)
    return g2.expand(channel, 1, window_size, window_size).contiguous()


def _ssim(img1, img2, window, window_size, channel, size_average=True):
    pad = window_size // 2
    blur = lambda t: F.conv2d(t, window, padding=pad, groups=channel)
    mu1, mu2 = blur(img1), blur(img2)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = blur(img1 * img1) - mu1_sq
    sigma2_sq = blur(img2 * img2) - mu2_sq
    sigma12 = blur(img1 * img2) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    return ssim_map.mean() if size_average else ssim_map.mean(1).mean(1).mean(1)


class SSIM(torch.nn.Module):

    def __init__(self, window_size=11, channel=1, size_average=True):
        super(SSIM, self).__init__()
        self.window_size, self.size_average, self.channel = window_size, size_average, channel
        self.window = create_window(window_size, channel)

    def forward(self, img1, img2):
        channel = img1.size(1)
        if channel != self.channel or self.window.device != img1.device or self.window.dtype != img1.dtype:
            self.window = create_window(self.window_size, channel).to(device=img1.device, dtype=img1.dtype)
            self.channel = channel
        return _ssim(img1, img2, self.window, self.window_size, channel, self.size_average)


def ssim(img1, img2, window_size=11, size_average=True):
    channel = img1.size(1)
    window = create_window(window_size, channel).to(device=img1.device, dtype=img1.dtype)
    return _ssim(img1, img2, window, window_size, channel, size_average)
