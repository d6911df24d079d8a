"""Host-side mirror of /root/reference/models/nerf.py (Embedding :4-38, NeRF :42-124).

Same constructors, parameter names and shapes as the reference (released checkpoints load with ``strict=True``), same
``forward`` signatures and results.  Inside RenderNet the two modules are fused away (nf_render_features emits the
encodings straight into the MLP operand layout, nf_nerf_mlp_fwd* consume it); called on their own — a caller that keeps the
reference's ``models/renderer.py`` and swaps only the modules (INTEGRATION.md level 2) — they run the SAME HIP kernels on
row-major tensors: ``nf_embed_fwd`` / ``nf_embed_bwd`` and the fp32-MFMA MLP (``nf_nerf_mlp_fwd`` / ``nf_nerf_mlp_bwd`` /
``nf_nerf_wgrad`` / ``nf_gemm_f32``), with autograd.  There is no torch fallback: CPU tensors raise.
"""
import ctypes

import torch
from torch import nn

from . import _lib, ops
from ._lib import check, ptr


class _EmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, n_freqs):
        lib = _lib.load()
        ops._require_cuda(x)
        C = x.shape[-1]
        xc = x.detach().contiguous().float()
        rows = xc.numel() // max(C, 1)
        out = torch.empty(*x.shape[:-1], C * (2 * n_freqs + 1), dtype=torch.float32, device=x.device)
        check(lib.nf_embed_fwd(ptr(xc), rows, C, n_freqs, ptr(out), _lib.stream()), "nf_embed_fwd")
        ctx.save_for_backward(xc)
        ctx.n_freqs = n_freqs
        return out

    @staticmethod
    def backward(ctx, g):
        (xc,) = ctx.saved_tensors
        lib = _lib.load()
        C = xc.shape[-1]
        gc = g.detach().contiguous().float()
        dx = torch.empty_like(xc)
        check(lib.nf_embed_bwd(ptr(xc), ptr(gc), xc.numel() // max(C, 1), C, ctx.n_freqs, ptr(dx), _lib.stream()), "nf_embed_bwd")
        return dx, None


class Embedding(nn.Module):
    """x -> (x, sin(2^k x), cos(2^k x), ...), k < N_freqs   (models/nerf.py:21-38)."""

    def __init__(self, in_channels, N_freqs, logscale=True):
        super().__init__()
        if not logscale:
            raise NotImplementedError("only logscale=True is used by the hot path (models/nerf.py:16-17)")
        self.N_freqs = N_freqs
        self.in_channels = in_channels
        self.out_channels = in_channels * (2 * N_freqs + 1)
        self.freq_bands = 2 ** torch.linspace(0, N_freqs - 1, N_freqs)

    def forward(self, x):
        """x: (B, in_channels) -> (B, out_channels), column order of models/nerf.py:33-38."""
        if x.shape[-1] != self.in_channels:
            raise ValueError(f"expected {self.in_channels} channels, got {x.shape[-1]}")
        return _EmbedFn.apply(x, self.N_freqs)


def _nerf_param_struct(layers):
    P = _lib.NerfParams()
    keep = []
    for i, l in enumerate(layers):
        w, b = l.weight.detach().contiguous().float(), l.bias.detach().contiguous().float()
        keep += [w, b]
        P.w[i], P.b[i] = w.data_ptr(), b.data_ptr()
    return P, keep


class _NerfFn(torch.autograd.Function):
    """NeRF.forward on (n, cx + cd) row-major rows: forward on the fp32-MFMA kernel with saved activations, backward =
    nf_nerf_mlp_bwd (data gradient) + nf_nerf_wgrad (15 weight GEMMs + bias sums) + three nf_gemm_f32 for dL/dx."""

    @staticmethod
    def forward(ctx, net, x, sigma_only, *params):
        lib = _lib.load()
        cx, cd = net.in_channels_xyz, net.in_channels_dir
        layers = net.linear_layers()
        dev = x.device
        n = x.shape[0]
        xf = x.detach().float()
        if sigma_only:          # models/nerf.py:100-113: only the position features are given; the view branch is not evaluated
            xf = torch.cat([xf, torch.zeros(n, cd, dtype=torch.float32, device=dev)], 1)
        P, keep = _nerf_param_struct(layers)
        packed = torch.empty(lib.nf_nerf_packed_floats(cx, cd), dtype=torch.float32, device=dev)
        check(lib.nf_nerf_pack(ctypes.byref(P), cx, cd, ptr(packed), _lib.stream()), "nf_nerf_pack")
        need = any(ctx.needs_input_grad)          # (grad mode is off inside Function.forward: ask the ctx, not torch)
        X = ops.rows_to_tiles(xf, cx, cd)
        n_rows = torch.tensor([n], dtype=torch.int32, device=dev)
        row_sample = torch.arange(n, dtype=torch.int32, device=dev)
        out = torch.zeros(max(n, 1), 4, dtype=torch.float32, device=dev)[:n]
        acts = torch.empty(max(n, 1) * 2432, dtype=torch.float32, device=dev) if need else None
        if n > 0:
            check(lib.nf_nerf_mlp_fwd(ptr(packed), cx, cd, ptr(X), ptr(n_rows), n, ptr(row_sample), ptr(out), ptr(acts),
                                      _lib.stream()), "nf_nerf_mlp_fwd")
        ctx.net, ctx.sigma_only, ctx.n = net, sigma_only, n
        ctx.x_needs_grad = bool(ctx.needs_input_grad[1])
        if need:
            # the transposed blob of the backward is packed NOW, from the weights this forward used: an optimiser step between
            # forward and backward must not pair forward-time activations with newer weights
            from .autograd_bwd import _pack_bwd
            packed_t = _pack_bwd(net, cx, cd, dev)
            ctx.save_for_backward(packed, packed_t, X, n_rows, row_sample, out, acts,
                                  layers[0].weight.detach().clone(), layers[4].weight.detach().clone(), layers[9].weight.detach().clone())
        return out[:, 3:4].clone() if sigma_only else out

    @staticmethod
    def backward(ctx, g):
        from .autograd_bwd import DPRE
        lib = _lib.load()
        st = _lib.stream()
        net, n = ctx.net, ctx.n
        cx, cd = net.in_channels_xyz, net.in_channels_dir
        layers = net.linear_layers()
        packed, packed_t, X, n_rows, row_sample, out, acts, W1, W5, Wd = ctx.saved_tensors
        dev = out.device
        # sigma_only (models/nerf.py:100-113) never touches xyz_encoding_final / dir_encoding / rgb: the reference leaves their
        # .grad at None (an optimiser with weight decay or moment decay must not step them), so do we
        unused = (8, 9, 11) if ctx.sigma_only else ()
        if n == 0:
            zeros = [None if i in unused else torch.zeros_like(l.weight) for i, l in enumerate(layers)] + \
                    [None if i in unused else torch.zeros_like(l.bias) for i, l in enumerate(layers)]
            return (None, torch.zeros(0, cx if ctx.sigma_only else cx + cd, device=dev) if ctx.x_needs_grad else None, None) + tuple(zeros)
        d_rs = torch.zeros(n, 4, dtype=torch.float32, device=dev)
        if ctx.sigma_only:
            d_rs[:, 3:4] = g.detach().float()
        else:
            d_rs.copy_(g.detach().float())
        dpre = torch.empty(n, DPRE, dtype=torch.float32, device=dev)
        check(lib.nf_nerf_mlp_bwd_n(ptr(packed), ptr(packed_t), cx, cd, ptr(acts), ptr(n_rows), n, ptr(row_sample), ptr(out),
                                  ptr(d_rs), ptr(dpre), st), "nf_nerf_mlp_bwd_n")
        nsl = 22          # 46 tiles x 22 row slices = 1 012 waves: one 128 x 128 tile per wave, one wave per SIMD
        blob = torch.empty(lib.nf_nerf_wgrad_floats(cx, cd), dtype=torch.float32, device=dev)
        wsp = torch.empty(lib.nf_nerf_wgrad_workspace_floats(cx, cd, nsl), dtype=torch.float32, device=dev)
        colsum = torch.empty(DPRE, dtype=torch.float32, device=dev)
        check(lib.nf_nerf_wgrad(ptr(dpre), ptr(acts), ptr(X), cx, cd, n, nsl, ptr(wsp), ptr(blob), ptr(colsum), st), "nf_nerf_wgrad")
        gw, o = [], 0
        for l in layers:
            k = l.weight.numel()
            gw.append(blob[o:o + k].view_as(l.weight))
            o += k
        gb = [colsum[k * 256:(k + 1) * 256] for k in range(8)]
        gb += [colsum[8 * 256:9 * 256], colsum[9 * 256:9 * 256 + 128], colsum[2435:2436], colsum[2432:2435]]
        dx = None
        for i in unused:
            gw[i] = gb[i] = None
        if ctx.x_needs_grad:
            dx = torch.empty(n, cx + cd, dtype=torch.float32, device=dev)
            ops.gemm(dpre[:, 0:256], W1, out=dx[:, :cx])
            ops.gemm(dpre[:, 4 * 256:5 * 256], W5[:, :cx], out=dx[:, :cx], accumulate=True)
            ops.gemm(dpre[:, 9 * 256:9 * 256 + 128], Wd[:, 256:], out=dx[:, cx:])
            if ctx.sigma_only:
                dx = dx[:, :cx].contiguous()
        return (None, dx, None) + tuple(gw) + tuple(gb)


class NeRF(nn.Module):
    """models/nerf.py:42-124 (D = 8, W = 256, skips = [4]): parameters with the reference's layer names; forward on the
    fp32-MFMA kernels."""

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
        """x: (B, in_channels_xyz + in_channels_dir) -> (B, 4) = [rgb (after the sigmoid), sigma];
        sigma_only: x is (B, in_channels_xyz) -> sigma (B, 1)   (models/nerf.py:83-124)."""
        want = self.in_channels_xyz + (0 if sigma_only else self.in_channels_dir)
        if x.dim() != 2 or x.shape[1] != want:
            raise ValueError(f"NeRF.forward expects (B, {want}) features, got {tuple(x.shape)}")
        ops._require_cuda(x)
        layers = self.linear_layers()
        params = [l.weight for l in layers] + [l.bias for l in layers]
        return _NerfFn.apply(self, x, bool(sigma_only), *params)
