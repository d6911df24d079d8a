"""CPU restatement (TEST INFRASTRUCTURE, never on the product path) of the image metrics of
/root/reference/utils/evaluate_images.ipynb: cell 4 (PSNR) and cell 5 (SSIM: 11x11 gaussian window sigma 1.5, five grouped
conv2d without padding, C1 = (0.01 L)^2, C2 = (0.03 L)^2).  Pinned by tests/golden/f4_ssim.npz, which holds what the
notebook's own classes returned (tests/golden/gen_golden_ssim.py executes the notebook cells)."""
import math

import torch
import torch.nn.functional as F


def psnr(pred, gt):
    return 10 * torch.log10(1 / torch.mean((pred - gt) ** 2))


def window(channel, w_size=11, sigma=1.5):
    g = torch.tensor([math.exp(-(x - w_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(w_size)], dtype=torch.float32)
    g = (g / g.sum()).unsqueeze(1)
    w2 = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)
    return w2.expand(channel, 1, w_size, w_size).contiguous()


def ssim(pred, gt, w_size=11, size_average=True):
    max_val = 255 if torch.max(pred) > 128 else 1
    min_val = -1 if torch.min(pred) < -0.5 else 0
    L = max_val - min_val
    c = pred.shape[1]
    w = window(c, w_size)
    conv = lambda t: F.conv2d(t, w, padding=0, groups=c)      # noqa: E731
    mu1, mu2 = conv(pred), conv(gt)
    m11, m22, m12 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s11, s22, s12 = conv(pred * pred) - m11, conv(gt * gt) - m22, conv(pred * gt) - m12
    C1, C2 = (0.01 * L) ** 2, (0.03 * L) ** 2
    v1, v2 = 2.0 * s12 + C2, s11 + s22 + C2
    smap = ((2 * m12 + C1) * v1) / ((m11 + m22 + C1) * v2)
    return smap.mean() if size_average else smap.mean(1).mean(1).mean(1)
