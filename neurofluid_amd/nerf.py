"""Host-side mirror of /root/reference/models/nerf.py (Embedding :4-38, NeRF :42-124).

The modules only OWN parameters (same names/shapes as the reference, so released checkpoints load
with ``strict=True``); the arithmetic runs in libneurofluid_hip (nf_render_features for the
positional encodings, nf_nerf_mlp_fwd / _bwd for the MLP).  There is no torch fallback forward.
"""
import torch
from torch import nn


class Embedding(nn.Module):
    """Positional encoding descriptor: x -> (x, sin(2^k x), cos(2^k x), ...), k < N_freqs."""

    def __init__(self, in_channels, N_freqs, logscale=True):
        super().__init__()
        if not logscale:
            raise NotImplementedError("only logscale=True is used by the hot path (models/nerf.py:16-17)")
        self.N_freqs = N_freqs
        self.in_channels = in_channels
        self.out_channels = in_channels * (2 * N_freqs + 1)
        self.freq_bands = 2 ** torch.linspace(0, N_freqs - 1, N_freqs)

    def forward(self, x):
        raise RuntimeError("Embedding is fused into nf_render_features; it has no standalone forward")


class NeRF(nn.Module):
    """Parameter container with the reference's layer names (state-dict compatible)."""

    def __init__(self, D=8, W=256, in_channels_xyz=63, in_channels_dir=27, skips=(4,)):
        super().__init__()
        if D != 8 or W != 256 or tuple(skips) != (4,):
            raise NotImplementedError("the MFMA kernel implements the reference's D=8, W=256, skips=[4] network")
        self.D, self.W, self.skips = D, W, list(skips)
        self.in_channels_xyz, self.in_channels_dir = in_channels_xyz, in_channels_dir
        for i in range(D):
            if i == 0:
                layer = nn.Linear(in_channels_xyz, W)
            elif i in self.skips:
                layer = nn.Linear(W + in_channels_xyz, W)
            else:
                layer = nn.Linear(W, W)
            setattr(self, f"xyz_encoding_{i + 1}", nn.Sequential(layer, nn.ReLU(True)))
        self.xyz_encoding_final = nn.Linear(W, W)
        self.dir_encoding = nn.Sequential(nn.Linear(W + in_channels_dir, W // 2), nn.ReLU(True))
        self.sigma = nn.Linear(W, 1)
        self.rgb = nn.Sequential(nn.Linear(W // 2, 3), nn.Sigmoid())

    def linear_layers(self):
        """The 12 Linear modules in the C-ABI order (nf_nerf_params_t)."""
        return [getattr(self, f"xyz_encoding_{i}")[0] for i in range(1, 9)] + [
            self.xyz_encoding_final, self.dir_encoding[0], self.sigma, self.rgb[0]]

    def forward(self, x, sigma_only=False):
        raise RuntimeError("NeRF.forward runs inside RenderNet (fused HIP path); no standalone torch forward")
