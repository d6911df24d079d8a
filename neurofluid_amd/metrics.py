"""Image metrics of the evaluation callers (/root/reference/utils/evaluate_images.ipynb cells 3-5, 7): MSE, PSNR and SSIM on
the device.  LPIPS (cell 6) needs the pretrained VGG weights of the `lpips` package and is not provided."""
import ctypes
import math

import torch

from . import _lib
from ._lib import check, ptr


def mse(pred, gt):
    return torch.mean((pred - gt) ** 2)


def psnr(pred, gt):
    """10 log10(1 / mse) (cell 4)."""
    return 10 * torch.log10(1 / torch.mean((pred - gt) ** 2))


def gaussian_window(w_size=11, sigma=1.5):
    """The 1-D factor of the notebook's window (cell 5, SSIM.gaussian): float32 values of exp(-(x - w//2)^2 / (2 sigma^2)),
    normalised in float32."""
    g = torch.tensor([math.exp(-(x - w_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(w_size)], dtype=torch.float32)
    return g / g.sum()


def dynamic_range(pred):
    """L of cell 5: 255 if max(pred) > 128 else 1, minus (-1 if min(pred) < -0.5 else 0)."""
    max_val = 255 if float(pred.max()) > 128 else 1
    min_val = -1 if float(pred.min()) < -0.5 else 0
    return max_val - min_val


def ssim(pred, gt, w_size=11, size_average=True):
    """SSIM of (B, C, H, W) images on the device (nf_image_ssim): the notebook's SSIM.__call__ without `full`."""
    if w_size != 11:
        raise NotImplementedError("the kernel is built for the notebook's 11x11 window")
    if pred.dim() != 4 or pred.shape != gt.shape:
        raise ValueError("pred and gt must be (B, C, H, W) tensors of one shape")
    if not pred.is_cuda:
        raise RuntimeError("neurofluid_amd.metrics.ssim runs on the GPU (HIP kernel); got a CPU tensor")
    lib = _lib.load()
    B, C, H, W = pred.shape
    p, g = pred.detach().contiguous().float(), gt.detach().contiguous().float()
    L = dynamic_range(p)
    win = (ctypes.c_float * 11)(*gaussian_window().tolist())
    ws = torch.empty(max(1, lib.nf_image_ssim_workspace_floats(B, C, H, W)), dtype=torch.float32, device=p.device)
    out = torch.empty(B, dtype=torch.float32, device=p.device)
    check(lib.nf_image_ssim(ptr(p), ptr(g), B, C, H, W, win, float(L), ptr(ws), ptr(out), _lib.stream()), "nf_image_ssim")
    return out.mean() if size_average else out
