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


def lpips_random_weights(seed=0):
    """A deterministic stand-in for the state dict of lpips.LPIPS(net='vgg') (same keys and shapes; He-scaled random convolutions, non-negative
    1x1 weights like the trained ones): the pretrained weights are not available offline, so the architecture is what can be checked."""
    g = torch.Generator().manual_seed(seed)
    idx = ((0, 2), (5, 7), (10, 12, 14), (17, 19, 21), (24, 26, 28))
    ch = (64, 128, 256, 512, 512)
    sd, cin = {}, 3
    for s, (ii, co) in enumerate(zip(idx, ch)):
        for i in ii:
            sd[f"net.slice{s + 1}.{i}.weight"] = torch.randn(co, cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * cin))
            sd[f"net.slice{s + 1}.{i}.bias"] = torch.randn(co, generator=g) * 0.05
            cin = co
        sd[f"lin{s}.model.1.weight"] = torch.rand(1, co, 1, 1, generator=g) / co
    sd["scaling_layer.shift"] = torch.tensor([-.030, -.088, -.188]).view(1, 3, 1, 1)
    sd["scaling_layer.scale"] = torch.tensor([.458, .448, .450]).view(1, 3, 1, 1)
    return sd


def lpips(pred, gt, sd, normalized=True):
    """Restatement of lpips.LPIPS(net='vgg') (v0.1, lpips=True, spatial=False) + the notebook's wrapper (cell 6: [0,1] -> [-1,1], torch.mean) from the
    package's published architecture: scaling layer, VGG16 feature slices relu1_2 / 2_2 / 3_3 / 4_3 / 5_3 (a 2x2 max-pool opens slices 2-5), unit-normalised
    channels (eps 1e-10 added to the norm), squared difference, non-negative 1x1 layer, spatial mean, sum over the five layers.  PARITY UNPINNED: neither the
    package nor its weights exist in this image."""
    if normalized:
        pred, gt = pred * 2.0 - 1.0, gt * 2.0 - 1.0
    idx = ((0, 2), (5, 7), (10, 12, 14), (17, 19, 21), (24, 26, 28))

    def feats(x):
        h = (x - sd["scaling_layer.shift"].view(1, 3, 1, 1)) / sd["scaling_layer.scale"].view(1, 3, 1, 1)
        out = []
        for s, ii in enumerate(idx):
            if s > 0:
                h = F.max_pool2d(h, 2, 2)
            for i in ii:
                h = F.relu(F.conv2d(h, sd[f"net.slice{s + 1}.{i}.weight"], sd[f"net.slice{s + 1}.{i}.bias"], padding=1))
            out.append(h)
        return out
    total = 0.
    for k, (f0, f1) in enumerate(zip(feats(pred), feats(gt))):
        n0 = f0 / (torch.sqrt(torch.sum(f0 ** 2, dim=1, keepdim=True)) + 1e-10)
        n1 = f1 / (torch.sqrt(torch.sum(f1 ** 2, dim=1, keepdim=True)) + 1e-10)
        d = F.conv2d((n0 - n1) ** 2, sd[f"lin{k}.model.1.weight"])
        total = total + d.mean([2, 3], keepdim=True)
    return torch.mean(total)
