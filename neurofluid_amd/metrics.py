"""Image metrics of the evaluation callers (/root/reference/utils/evaluate_images.ipynb cells 3-5, 7): MSE, PSNR and SSIM on
the device, and LPIPS (cell 6) — the network is built here (class LPIPS below); its pretrained weights are third-party data the caller passes."""
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


# ------------------------------------------------------------------------------------------------
# LPIPS (cell 6 of the notebook: lpips.LPIPS(net='vgg'), inputs scaled [0,1] -> [-1,1], torch.mean of the per-image distances)
# ------------------------------------------------------------------------------------------------
_VGG_SLICES = ((0, 2), (5, 7), (10, 12, 14), (17, 19, 21), (24, 26, 28))      # conv indices of torchvision's vgg16.features per LPIPS slice
_VGG_CHANNELS = (64, 128, 256, 512, 512)


class LPIPS:
    """Learned Perceptual Image Patch Similarity, the `lpips` package's `LPIPS(net='vgg')` (version 0.1, linear layers, spatial average) as the
    notebook calls it (`utils/evaluate_images.ipynb` cell 6), on the device: the 13 VGG16 convolutions run as im2col (channels-last slices) +
    `nf_gemm_f32` — no vendor convolution library.

    The network's WEIGHTS (VGG16 pretrained on ImageNet + the five learned 1x1 layers) are third-party data that cannot be fetched in this
    image; the caller passes them: `weights` = the state dict of `lpips.LPIPS(net='vgg')` (or a path to it saved with torch.save), i.e. on any
    machine with the package:  python -c "import lpips, torch; torch.save(lpips.LPIPS(net='vgg').state_dict(), 'lpips_vgg.pt')"
    Keys used: net.slice{1..5}.{idx}.weight / .bias (idx = torchvision's vgg16.features numbering), lin{0..4}.model.1.weight,
    scaling_layer.shift / .scale (defaults are the package's constants).  tests/test_gpu_metrics.py checks the arithmetic against the
    oracle's restatement with random weights; with the real weights the numbers are the package's."""

    SHIFT = (-.030, -.088, -.188)
    SCALE = (.458, .448, .450)

    def __init__(self, weights, device=None):
        if isinstance(weights, str):
            weights = torch.load(weights, map_location="cpu")
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        self.convs = []
        for s, idxs in enumerate(_VGG_SLICES):
            layer = []
            for i in idxs:
                w = weights[f"net.slice{s + 1}.{i}.weight"].float()            # (Cout, Cin, 3, 3)
                b = weights[f"net.slice{s + 1}.{i}.bias"].float()
                # im2col column order below is (dy, dx, cin): weight matrix (9 * Cin, Cout)
                wm = w.permute(2, 3, 1, 0).reshape(9 * w.shape[1], w.shape[0]).contiguous().to(dev)
                layer.append((wm, b.to(dev)))
            self.convs.append(layer)
        self.lins = [weights[f"lin{k}.model.1.weight"].float().reshape(-1).to(dev) for k in range(5)]
        self.shift = weights.get("scaling_layer.shift", torch.tensor(self.SHIFT)).float().reshape(3).to(dev)
        self.scale = weights.get("scaling_layer.scale", torch.tensor(self.SCALE)).float().reshape(3).to(dev)

    @staticmethod
    def _conv3x3_relu(x, wm, b):
        """x (N, H, W, C) channels-last -> relu(conv3x3, padding 1) (N, H, W, Cout): nine shifted slices side by side, one GEMM per image."""
        from . import ops
        N, H, W, C = x.shape
        xp = torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1))
        out = torch.empty(N, H, W, wm.shape[1], dtype=torch.float32, device=x.device)
        for n in range(N):          # (per image: the column matrix of a 400 x 400 image at 64 channels is 368 MB)
            cols = torch.cat([xp[n, dy:dy + H, dx:dx + W, :] for dy in range(3) for dx in range(3)], dim=-1).reshape(H * W, 9 * C)
            ops.gemm(cols, wm, out=out[n].view(H * W, -1))
        return torch.relu_(out.add_(b))

    def features(self, x):
        """x (N, 3, H, W) in [-1, 1] -> the five activation maps (channels-last)."""
        h = ((x.permute(0, 2, 3, 1) - self.shift) / self.scale).contiguous()
        feats = []
        for s, layer in enumerate(self.convs):
            if s > 0:               # MaxPool2d(2, 2) opens slices 2..5 (floor: a trailing odd row / column is dropped)
                N, H, W, C = h.shape
                h = h[:, :H // 2 * 2, :W // 2 * 2].reshape(N, H // 2, 2, W // 2, 2, C).amax(dim=(2, 4))
            for wm, b in layer:
                h = self._conv3x3_relu(h, wm, b)
            feats.append(h)
        return feats

    def __call__(self, y_pred, y_true, normalized=True):
        """(B, 3, H, W) images -> torch.mean of the B distances (cell 6)."""
        if not y_pred.is_cuda:
            raise RuntimeError("neurofluid_amd.metrics.LPIPS runs on the GPU; got a CPU tensor")
        p, g = y_pred.detach().float().to(self.device), y_true.detach().float().to(self.device)
        if normalized:
            p, g = p * 2.0 - 1.0, g * 2.0 - 1.0
        total = 0.
        for f0, f1, lin in zip(self.features(p), self.features(g), self.lins):
            n0 = f0 / (f0.pow(2).sum(-1, keepdim=True).sqrt() + 1e-10)
            n1 = f1 / (f1.pow(2).sum(-1, keepdim=True).sqrt() + 1e-10)
            total = total + ((n0 - n1).pow(2) * lin).sum(-1).mean(dim=(1, 2))      # 1x1 "lin" layer, then the spatial average
        return total.mean()
