"""
CPU restatement of the particle-driven NeRF renderer (TEST INFRASTRUCTURE — see oracle/__init__.py).

Stage-for-stage restatement in torch-CPU fp32 of
  /root/reference/utils/ray_utils.py   (A0 get_ray_directions/get_rays :85-130, A1 coarse_sample_ray
                                         :232-256, A9 sample_pdf/ImportanceSampling :178-229)
  /root/reference/models/nerf.py       (A5 Embedding :4-38, A6 NeRF :42-124)
  /root/reference/models/renderer.py   (A2 search :112-122, A3 smoothing_position :96-109,
                                         A4 embedding_local_geometry :125-179, A7 mask :233-237,
                                         A8 render_image :182-208, A10 forward :211-270)
Everything is a pure function of tensors + a flat ``state`` dict that uses the reference's
state-dict key names (``nerf_coarse.xyz_encoding_1.0.weight`` ...).  Pinned by tests/golden/*.npz.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import neighbors

# ----------------------------------------------------------------------------------------------
# configuration mirror of configs/warmup.yaml:29-45 (RENDERER node)
# ----------------------------------------------------------------------------------------------
DEFAULT_CFG = dict(
    use_mask=True, N_samples=64, N_importance=128,
    fix_radius=True, particle_radius=0.025, search_raduis_scale=9.0, N_neighbor=20,
    density=True, var=True, smoothed_pos=True, smoothed_dir=True, exclude_ray=True,
    same_smooth_factor=False,
)


def nerf_channels(cfg=DEFAULT_CFG):
    """in_channels_xyz / in_channels_dir as built in models/renderer.py:30-44."""
    cx, cd = 63, 27
    if cfg["density"]:
        cx += 9
    if cfg["var"]:
        cx += 63
    if cfg["smoothed_pos"]:
        cx += 63
    if cfg["smoothed_dir"]:
        cd += 27
    return cx, cd


# ----------------------------------------------------------------------------------------------
# A0  rays
# ----------------------------------------------------------------------------------------------
def get_ray_directions(H, W, focal):
    """utils/ray_utils.py:85-104 (kornia.create_meshgrid(H,W,normalized_coordinates=False) is the
    pixel grid x = 0..W-1 along columns, y = 0..H-1 along rows, float32)."""
    xs = torch.linspace(0, W - 1, W, dtype=torch.float32)
    ys = torch.linspace(0, H - 1, H, dtype=torch.float32)
    j, i = torch.meshgrid(ys, xs, indexing="ij")  # i = x (col), j = y (row)
    return torch.stack([(i - W / 2) / focal, -(j - H / 2) / focal, -torch.ones_like(i)], -1)


def get_rays(directions, c2w):
    """utils/ray_utils.py:107-130."""
    rays_d = directions @ c2w[:, :3].T
    rays_d = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
    rays_o = c2w[:, 3].expand(rays_d.shape)
    return rays_o, rays_d


# ----------------------------------------------------------------------------------------------
# A1  coarse sampling (perturb == 0: the only mode the callers use, trainer/basetrainer.py:284-289;
#     use_disp = linear in disparity, utils/ray_utils.py:236-240, pinned by tests/golden/a1_a10_disp.npz)
# ----------------------------------------------------------------------------------------------
def coarse_z(near, far, n, use_disp=False):
    t = torch.linspace(0, 1, n)
    if use_disp:
        return 1 / (1 / near * (1 - t) + 1 / far * t)
    return near * (1 - t) + far * t


def coarse_sample_ray(near, far, rays, n, use_disp=False, perturb=0, perturb_rand=None):
    """utils/ray_utils.py:232-256.  perturb > 0 (:247-253): every depth is redrawn inside its interval [lower, upper]
    (mid-points of the table) at perturb * U[0,1); perturb_rand = the (R, n) uniform draws, or None = torch.rand from the
    global generator, as the reference draws them."""
    z = coarse_z(near, far, n, use_disp).expand(rays.shape[0], n)
    if perturb > 0:
        mid = 0.5 * (z[:, :-1] + z[:, 1:])
        upper = torch.cat([mid, z[:, -1:]], -1)
        lower = torch.cat([z[:, :1], mid], -1)
        if perturb_rand is None:
            perturb_rand = torch.rand(z.shape)
        z = lower + (upper - lower) * (perturb * perturb_rand)
    xyz = rays[:, None, 0:3] + rays[:, None, 3:6] * z[:, :, None]
    return z, xyz


# ----------------------------------------------------------------------------------------------
# A5 / A6  positional encoding + MLP
# ----------------------------------------------------------------------------------------------
def embed(x, n_freqs):
    """models/nerf.py:21-38 with freq_bands = 2**linspace(0, N-1, N) (:17)."""
    freqs = 2 ** torch.linspace(0, n_freqs - 1, n_freqs)
    out = [x]
    for f in freqs:
        out.append(torch.sin(f * x))
        out.append(torch.cos(f * x))
    return torch.cat(out, -1)


def nerf_forward(state, prefix, x, cx, cd, sigma_only=False):
    """models/nerf.py:83-124; D=8, W=256, skips=[4]."""
    def lin(name, v):
        return F.linear(v, state[f"{prefix}.{name}.weight"], state[f"{prefix}.{name}.bias"])

    if sigma_only:
        xin = x
    else:
        xin, din = torch.split(x, [cx, cd], dim=-1)
    h = xin
    for i in range(8):
        if i == 4:
            h = torch.cat([xin, h], -1)
        h = torch.relu(lin(f"xyz_encoding_{i + 1}.0", h))
    sigma = lin("sigma", h)
    if sigma_only:
        return sigma
    fin = lin("xyz_encoding_final", h)
    hd = torch.relu(lin("dir_encoding.0", torch.cat([fin, din], -1)))
    rgb = torch.sigmoid(lin("rgb.0", hd))
    return torch.cat([rgb, sigma], -1)


# ----------------------------------------------------------------------------------------------
# A2  neighbour search
# ----------------------------------------------------------------------------------------------
def search(ray_particles, particles, radius, K):
    """models/renderer.py:112-122: the reference replicates the particle cloud once per ray and
    calls pytorch3d ball_query; every ray sees the same cloud, so one flat query is identical."""
    R, S, _ = ray_particles.shape
    d, i, nn = neighbors.ball_query_firstk(ray_particles.detach().reshape(-1, 3).numpy(), particles.detach().numpy(), radius, K)
    d, i, nn = torch.from_numpy(d).view(R, S, K), torch.from_numpy(i).view(R, S, K), torch.from_numpy(nn).view(R, S, K, 3)
    if particles.requires_grad:
        # e2e training (A12): the indices are data, `nn` a differentiable gather of the cloud (the same values, bit for bit)
        nn = torch.where((i >= 0).unsqueeze(-1), particles[i.clamp(min=0)], torch.zeros(1))
    return d, i, nn


# ----------------------------------------------------------------------------------------------
# A3 / A4  local geometry features
# ----------------------------------------------------------------------------------------------
def smoothing_position(ray_pos, nn_poses, radius, num_nn=None, exclude_ray=True, same_smooth_factor=False,
                       larger_alpha=0.9, smaller_alpha=0.1):
    """models/renderer.py:96-109.  exclude_ray=True (configs/warmup.yaml:44): the weighted neighbour mean; False (:100-106):
    the ray position blended with it, alpha = 0.9, or 0.1 where `num_nn.le(20)` (the literal 20 of :105, whatever K is)
    unless same_smooth_factor.  Pinned by tests/golden/cfg_incl_ray*.npz."""
    dists = torch.norm(nn_poses - ray_pos.unsqueeze(-2), dim=-1)
    w = torch.clamp(1 - (dists / radius) ** 3, min=0)
    pos = (w.unsqueeze(-1) * nn_poses).sum(-2) / (w.sum(-1, keepdim=True) + 1e-12)
    if not exclude_ray:
        alpha = torch.ones(ray_pos.shape[0], ray_pos.shape[1], 1) * larger_alpha
        if not same_smooth_factor:
            alpha[num_nn.le(20)] = smaller_alpha
        pos = ray_pos * (1 - alpha) + pos * alpha
    return pos, w.sum(-1, keepdim=True)


def particle_direction(p, ro):
    """models/renderer.py:56-60."""
    d = p - ro.expand(p.shape[0], -1)
    return d / torch.norm(d, dim=-1, keepdim=True)


def embedding_local_geometry(dists, neighbors_xyz, radius, ray_particles, rays, ro, cfg=DEFAULT_CFG):
    """models/renderer.py:125-179 -> (feats (R*S, cx+cd), num_nn (R,S,1) int64).
    Column order: [PE10(x) | PE4(density) | PE10(smoothed) | PE10(var) | PE4(ray_dir) | PE4(smoothed_dir)]."""
    R, S, K = dists.shape
    nn_mask = dists.ne(0)
    num_nn = nn_mask.sum(-1, keepdim=True)
    pos_feats = [embed(ray_particles.reshape(-1, 3), 10)]
    dir_feats = [torch.repeat_interleave(embed(rays[:, 3:], 4), repeats=S, dim=0)]
    smoothed, density = smoothing_position(ray_particles, neighbors_xyz, radius, num_nn, cfg.get("exclude_ray", True),
                                           cfg.get("same_smooth_factor", False))
    sdir = particle_direction(smoothed.reshape(-1, 3), ro)
    if cfg["density"]:
        pos_feats.append(embed(density.reshape(-1, 1), 4))
    if cfg["smoothed_pos"]:
        pos_feats.append(embed(smoothed.reshape(-1, 3), 10))
    if cfg["var"]:
        m = nn_mask.unsqueeze(-1).to(neighbors_xyz.dtype)
        v = (neighbors_xyz - ray_particles.unsqueeze(-2)) * m
        mean = v.sum(-2) / (num_nn + 1e-12)
        var = (((v - mean.unsqueeze(-2)) ** 2) * m).sum(-2) / (num_nn + 1e-12)
        pos_feats.append(embed(var.reshape(-1, 3), 10))
    if cfg["smoothed_dir"]:
        dir_feats.append(embed(sdir, 4))
    return torch.cat(pos_feats + dir_feats, dim=1), num_nn


# ----------------------------------------------------------------------------------------------
# A8  alpha compositing
# ----------------------------------------------------------------------------------------------
def render_image(rgbsigma, zvals, rays, white_background=True, noise=None):
    """models/renderer.py:182-208; noise = noise_std * randn(sigmas.shape) (:193-195) or None."""
    rgbs, sigmas = rgbsigma[..., :3], rgbsigma[..., 3]
    if noise is not None:
        sigmas = sigmas + noise
    deltas = zvals[:, 1:] - zvals[:, :-1]
    deltas = torch.cat([deltas, 1e10 * torch.ones_like(deltas[:, :1])], -1)
    deltas = deltas * torch.norm(rays[:, 3:].unsqueeze(1), dim=-1)
    alphas = 1 - torch.exp(-deltas * torch.relu(sigmas))
    shifted = torch.cat([torch.ones_like(alphas[:, :1]), 1 - alphas + 1e-10], -1)
    weights = alphas * torch.cumprod(shifted, -1)[:, :-1]
    wsum = weights.sum(1)
    rgb = torch.sum(weights.unsqueeze(-1) * rgbs, -2)
    depth = torch.sum(weights * zvals, -1)
    if white_background:
        rgb = rgb + 1 - wsum.unsqueeze(-1)
    return rgb, depth, weights


# ----------------------------------------------------------------------------------------------
# A9  importance sampling (det = (perturb == 0), models/renderer.py:250)
# ----------------------------------------------------------------------------------------------
def sample_pdf(bins, weights, n, u=None):
    """utils/ray_utils.py:178-220.  u = None: the det branch (linspace); else the (R, n) uniform draws of :190."""
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if u is None:
        u = torch.linspace(0., 1., steps=n).expand(list(cdf.shape[:-1]) + [n])
    u = u.contiguous()
    inds = torch.searchsorted(cdf.detach(), u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    cdf_lo, cdf_hi = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    bin_lo, bin_hi = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = cdf_hi - cdf_lo
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_lo) / denom
    return bin_lo + t * (bin_hi - bin_lo)


def importance_sampling(zvals, weights, n_importance, rays_o, rays_d, u=None):
    """utils/ray_utils.py:222-229."""
    mid = 0.5 * (zvals[..., 1:] + zvals[..., :-1])
    z_new = sample_pdf(mid, weights[:, 1:-1], n_importance, u).detach()
    z_all, _ = torch.sort(torch.cat([zvals, z_new], -1), -1)
    xyz = rays_o[..., None, :] + rays_d[..., None, :] * z_all[..., :, None]
    return xyz, z_all


# ----------------------------------------------------------------------------------------------
# A10  the whole chunk
# ----------------------------------------------------------------------------------------------
def render_pass(state, prefix, particles, ro, rays, z, xyz, cfg, white_background=True, noise=None):
    radius = cfg["search_raduis_scale"] * cfg["particle_radius"]
    cx, cd = nerf_channels(cfg)
    S = z.shape[1]
    dists, idx, nn = search(xyz, particles, radius, cfg["N_neighbor"])
    feats, num_nn = embedding_local_geometry(dists, nn, radius, xyz, rays, ro, cfg)
    rgbsigma = nerf_forward(state, prefix, feats, cx, cd).view(-1, S, 4)
    mask = torch.all(dists != 0, dim=-1, keepdim=True).float()
    if cfg["use_mask"]:
        rgbsigma = rgbsigma * mask
    rgb, depth, weights = render_image(rgbsigma, z, rays, white_background, noise)
    return dict(rgb=rgb, depth=depth, weights=weights, num_nn=num_nn, mask=mask, rgbsigma=rgbsigma,
                idx=idx, dists=dists, feats=feats)


def render_forward(state, particles, ro, rays, near, far, cfg=DEFAULT_CFG, white_background=True,
                   return_debug=False, use_disp=False, noise_std=0.0, noise=None, perturb=0, perturb_draws=None):
    """models/renderer.py:211-270 -> dict with the reference's keys.  noise_std > 0: sigma noise, one torch.randn draw per pass
    in the reference's order (coarse, fine) from the global CPU generator, or the two tensors `noise` = (n0, n1) (already scaled).
    perturb > 0: jittered coarse depths and random inverse-CDF draws; perturb_draws = (rand (R, N_samples), rand (R, N_importance))
    or None = torch.rand from the global generator at the reference's two call sites (so that, seeded alike, the draws of this
    function and of the reference coincide: coarse jitter, [coarse noise], u, [fine noise])."""
    pr0, pu1 = perturb_draws if perturb_draws is not None else (None, None)
    z0, xyz0 = coarse_sample_ray(near, far, rays, cfg["N_samples"], use_disp, perturb, pr0)
    n0 = n1 = None
    if noise is not None:
        n0, n1 = noise
    elif noise_std:
        n0 = torch.randn(rays.shape[0], cfg["N_samples"]) * noise_std
    p0 = render_pass(state, "nerf_coarse", particles, ro, rays, z0, xyz0, cfg, white_background, n0)
    out = {"rgb0": p0["rgb"], "depth0": p0["depth"], "opacity0": p0["weights"].sum(1),
           "num_nn_0": p0["num_nn"], "mask_0": p0["mask"].sum(1)}
    dbg = {"z0": z0, "weights0": p0["weights"], "rgbsigma0": p0["rgbsigma"]}
    if cfg["N_importance"] > 0:
        if perturb > 0 and pu1 is None:
            pu1 = torch.rand(rays.shape[0], cfg["N_importance"])
        xyz1, z1 = importance_sampling(z0, p0["weights"], cfg["N_importance"], rays[..., :3], rays[..., 3:],
                                       pu1 if perturb > 0 else None)
        if noise is None and noise_std:
            n1 = torch.randn(rays.shape[0], cfg["N_samples"] + cfg["N_importance"]) * noise_std
        p1 = render_pass(state, "nerf_fine", particles, ro, rays, z1, xyz1, cfg, white_background, n1)
        out.update({"rgb1": p1["rgb"], "depth1": p1["depth"], "opacity1": p1["weights"].sum(1),
                    "num_nn_1": p1["num_nn"], "mask_1": p1["mask"].sum(1)})
        dbg.update({"z1": z1, "weights1": p1["weights"], "rgbsigma1": p1["rgbsigma"]})
    if return_debug:
        return out, dbg
    return out


# ----------------------------------------------------------------------------------------------
# deterministic closed-form weights (goldens + benchmarks): throughput is weight-independent,
# and a formula keeps fixtures small.
# ----------------------------------------------------------------------------------------------
def nerf_layer_shapes(cx, cd, W=256):
    shapes = {}
    for i in range(8):
        fan_in = cx if i == 0 else (W + cx if i == 4 else W)
        shapes[f"xyz_encoding_{i + 1}.0"] = (W, fan_in)
    shapes["xyz_encoding_final"] = (W, W)
    shapes["dir_encoding.0"] = (W // 2, W + cd)
    shapes["sigma"] = (1, W)
    shapes["rgb.0"] = (3, W // 2)
    return shapes


def deterministic_nerf_state(prefixes=("nerf_coarse", "nerf_fine"), cfg=DEFAULT_CFG, scale=1.0):
    """w[o,i] = sqrt(2.4/fan_in) * sin(0.37*(o*fan_in+i) + 1.3*layer_no + 0.5*net_no), small sinusoidal
    biases; the sigma bias is positive so densities are not all clipped by the relu."""
    cx, cd = nerf_channels(cfg)
    state = {}
    for n, prefix in enumerate(prefixes):
        for l, (name, (o, i)) in enumerate(nerf_layer_shapes(cx, cd).items()):
            k = torch.arange(o * i, dtype=torch.float64).view(o, i)
            w = math.sqrt(2.4 / i) * scale * torch.sin(0.37 * k + 1.3 * l + 0.5 * n)
            b = 0.05 * torch.cos(0.11 * torch.arange(o, dtype=torch.float64) + l + n)
            if name == "sigma":
                b = b + 2.0
            state[f"{prefix}.{name}.weight"] = w.float()
            state[f"{prefix}.{name}.bias"] = b.float()
    return state


# ----------------------------------------------------------------------------------------------
# synthetic "watercube 400^2" scene
# ----------------------------------------------------------------------------------------------
def watercube_particles(n_side=17, spacing=0.05, corner=(-0.40, -0.40, -0.975), jitter=0.005, seed=10):
    ax = [corner[d] + spacing * np.arange(n_side) for d in range(3)]
    g = np.stack(np.meshgrid(ax[0], ax[1], ax[2], indexing="ij"), -1).reshape(-1, 3)
    g = g + np.random.RandomState(seed).uniform(-jitter, jitter, g.shape)
    return torch.from_numpy(g.astype(np.float32))


def eval_camera():
    """Camera pose (c2w, 3x4) used by the synthetic benchmark scene: the pose *values* of the test
    camera at /root/reference/eval_renderer.py:67-92 (looks at the origin from 10.76 units; rotation
    columns have norm 0.4155, so the normalisation in get_rays matters).  Data, not code."""
    return torch.tensor([
        [0.3597943186759949, 0.09052024036645889, -0.18696719408035278, -4.842308521270752],
        [-0.2077273577451706, 0.15678563714027405, -0.32383665442466736, -8.387124061584473],
        [0.0, 0.37393447756767273, 0.181040421128273, 4.688809871673584]], dtype=torch.float32)


def camera_focal(W, camera_angle_x=0.323):
    return 0.5 * W / math.tan(0.5 * camera_angle_x)


def psnr(a, b):
    """trainer/trainer_renderer.py:19-20."""
    mse = torch.mean((a - b) ** 2).item()
    return float("inf") if mse == 0 else -10.0 * math.log10(mse)
