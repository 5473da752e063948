"""Loss / schedule helpers of the train-step harness — the repo's own counterparts of
/root/reference/utils/loss_utils.py:17-66 (l1, SSIM map), utils/image_utils.py:17-19 (psnr),
utils/general_utils.py:18,29-62 (inverse_sigmoid, exponential LR schedule) and utils/sh_utils.py:114 (RGB2SH).
Pinned by tests/golden/ref_helpers.npz (tests/test_golden_helpers.py).  Plain PyTorch: these sit around the
rasterizer boundary, not on the native hot path."""
import math

import numpy as np
import torch
import torch.nn.functional as F

SH_C0 = 0.28209479177387814


def l1_loss(pred, gt, reduce=True):
    d = (pred - gt).abs()
    return d.mean() if reduce else d


_WINDOWS = {}


def _gauss_window(size, sigma, channels, like):
    key = (size, sigma, channels, like.device, like.dtype)
    w = _WINDOWS.get(key)
    if w is None:
        g = torch.tensor([math.exp(-((x - size // 2) ** 2) / float(2 * sigma ** 2)) for x in range(size)])
        g = (g / g.sum()).unsqueeze(1)
        w2 = (g @ g.t()).float()[None, None].expand(channels, 1, size, size).contiguous()
        w = w2.to(device=like.device, dtype=like.dtype)
        _WINDOWS[key] = w
    return w


def ssim(img1, img2, window_size=11, size_average=True):
    """Structural similarity with an 11x11 Gaussian window (sigma 1.5), zero padding, per channel (depthwise)."""
    ch = img1.size(-3)
    w = _gauss_window(window_size, 1.5, ch, img1)
    pad = window_size // 2
    conv = lambda x: F.conv2d(x, w, padding=pad, groups=ch)
    mu1, mu2 = conv(img1), conv(img2)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = conv(img1 * img1) - mu1_sq
    s2 = conv(img2 * img2) - mu2_sq
    s12 = conv(img1 * img2) - mu12
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu12 + c1) * (2 * s12 + c2)) / ((mu1_sq + mu2_sq + c1) * (s1 + s2 + c2))
    return m.mean() if size_average else m


def psnr(img1, img2):
    mse = ((img1 - img2) ** 2).reshape(img1.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def rgb_to_sh(rgb):
    return (rgb - 0.5) / SH_C0


def expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """Log-linear interpolation lr_init -> lr_final over max_steps, optionally eased in over lr_delay_steps."""
    def f(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        if lr_delay_steps > 0:
            delay = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
        else:
            delay = 1.0
        t = np.clip(step / max_steps, 0, 1)
        return delay * np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t)
    return f
