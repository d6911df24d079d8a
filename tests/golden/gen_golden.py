#!/usr/bin/env python3
"""
Golden-vector generator.  Runs ONLY in the authoring container, where /root/reference exists: it
imports the reference's own Python modules (models/nerf.py, models/renderer.py, utils/ray_utils.py,
models/transmodel.py) and records inputs + the reference's outputs as small .npz fixtures next to
this file.  The reference source never travels: only these data files are committed.

Stand-ins needed to make the import succeed:
  * kornia.create_meshgrid      -> 6-line pixel-grid function (only get_ray_directions uses it)
  * pytorch3d.ops.ball_query    -> oracle.neighbors.ball_query_firstk (documented pytorch3d semantics);
                                   fixtures that depend on it are flagged  standin_ball_query=1
  * open3d.ml.torch             -> dummy module (ContinuousConv cannot run); only integrate_pos_vel /
                                   update_pos_vel / _window_poly6 of ParticleNet are recorded
Usage:  python tests/golden/gen_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)


def install_standins():
    from oracle import neighbors

    kornia = types.ModuleType("kornia")

    def create_meshgrid(H, W, normalized_coordinates=False):
        xs = torch.linspace(0, W - 1, W)
        ys = torch.linspace(0, H - 1, H)
        gy, gx = torch.meshgrid(ys, xs, indexing="ij")
        return torch.stack([gx, gy], -1).unsqueeze(0)

    kornia.create_meshgrid = create_meshgrid
    sys.modules["kornia"] = kornia

    p3d = types.ModuleType("pytorch3d")
    ops = types.ModuleType("pytorch3d.ops")

    def ball_query(p1, p2, radius, K):
        ds, ids, nns = [], [], []
        for b in range(p1.shape[0]):
            d, i, n = neighbors.ball_query_firstk(p1[b].detach().numpy(), p2[b].detach().numpy(), radius, K)
            ds.append(torch.from_numpy(d)); ids.append(torch.from_numpy(i)); nns.append(torch.from_numpy(n))
        return torch.stack(ds), torch.stack(ids), torch.stack(nns)

    ops.ball_query = ball_query
    p3d.ops = ops
    sys.modules["pytorch3d"] = p3d
    sys.modules["pytorch3d.ops"] = ops

    o3d = types.ModuleType("open3d")
    ml = types.ModuleType("open3d.ml")
    mlt = types.ModuleType("open3d.ml.torch")
    layers = types.ModuleType("open3d.ml.torch.layers")

    class ContinuousConv(torch.nn.Module):
        def __init__(self, **kw):
            super().__init__()

    layers.ContinuousConv = ContinuousConv
    mlt.layers = layers
    mlt.ops = types.ModuleType("open3d.ml.torch.ops")
    ml.torch = mlt
    o3d.ml = ml
    for n, m in [("open3d", o3d), ("open3d.ml", ml), ("open3d.ml.torch", mlt)]:
        sys.modules[n] = m


class Node(dict):
    __getattr__ = dict.__getitem__


def renderer_cfg():
    return Node(use_mask=True,
                ray=Node(ray_chunk=1024, N_importance=128, N_samples=64),
                NN_search=Node(fix_radius=True, particle_radius=0.025, search_raduis_scale=9.0, N_neighbor=20),
                encoding=Node(density=True, var=True, smoothed_pos=True, smoothed_dir=True, exclude_ray=True,
                              same_smooth_factor=False))


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path)} bytes)")


def main():
    assert os.path.isdir(REF), "golden vectors can only be regenerated where /root/reference exists"
    install_standins()
    sys.path.insert(0, REF)
    from models.nerf import Embedding, NeRF                       # noqa: E402  (reference code)
    from models.renderer import RenderNet                         # noqa: E402
    from utils import ray_utils                                   # noqa: E402
    from models.transmodel import ParticleNet                     # noqa: E402
    from oracle import render_oracle as ro

    g = torch.Generator().manual_seed(1234)
    c2w = ro.eval_camera()

    # ---- A0 rays
    H, W = 12, 16
    focal = ro.camera_focal(400)
    dirs = ray_utils.get_ray_directions(H, W, focal)
    rays_o, rays_d = ray_utils.get_rays(dirs, c2w)
    save("a0_rays", H=H, W=W, focal=focal, c2w=c2w, directions=dirs, rays_o=rays_o.contiguous(), rays_d=rays_d)

    # a few rays through the synthetic cube at 400x400
    dirs400 = ray_utils.get_ray_directions(400, 400, focal)
    o4, d4 = ray_utils.get_rays(dirs400, c2w)
    rays400 = torch.cat([o4, d4], -1)
    sel = torch.cat([rays400[200, 180:212], rays400[150:182:4, 205], rays400[10, 0:8]], 0).contiguous()  # 48 rays

    # ---- A1 coarse sampling
    z, xyz = ray_utils.coarse_sample_ray(9.0, 13.0, sel[:5], 64, False, 0)
    save("a1_coarse", rays=sel[:5], near=9.0, far=13.0, z=z.contiguous(), xyz=xyz)

    # ---- A5 embeddings
    x3 = torch.randn(7, 3, generator=g) * 2
    x1 = torch.rand(7, 1, generator=g) * 20
    save("a5_embed", x3=x3, x1=x1, e3_10=Embedding(3, 10)(x3), e3_4=Embedding(3, 4)(x3), e1_4=Embedding(1, 4)(x1))

    # ---- A6 NeRF MLP with closed-form weights
    state = ro.deterministic_nerf_state()
    cx, cd = ro.nerf_channels()
    net = NeRF(in_channels_xyz=cx, in_channels_dir=cd)
    net.load_state_dict({k[len("nerf_coarse."):]: v for k, v in state.items() if k.startswith("nerf_coarse.")}, strict=True)
    xin = torch.randn(40, cx + cd, generator=g).clamp(-1, 1)
    with torch.no_grad():
        save("a6_nerf", x=xin, out=net(xin), sigma_only=net(xin[:, :cx], sigma_only=True))

    # ---- renderer stages on a small cloud
    P = ro.watercube_particles()
    rn = RenderNet(renderer_cfg(), near=9.0, far=13.0)
    rn.load_state_dict(state, strict=True)
    ro_cam = rn.set_ro(c2w)
    with torch.no_grad():
        rays8 = sel[:8]
        z0, xyz0 = ray_utils.coarse_sample_ray(9.0, 13.0, rays8, 64, False, 0)
        dists, idx, nn, radius = rn.search(xyz0, P, True)
        num_nn = dists.ne(0).sum(-1, keepdim=True)
        sm, dens = rn.smoothing_position(xyz0, nn, radius, num_nn, exclude_ray=True)
        save("a3_smoothing", ray_pos=xyz0, nn=nn, radius=radius, smoothed=sm, density=dens, standin_ball_query=1)
        pos_f, dir_f, num_nn2 = rn.embedding_local_geometry(dists, idx, nn, radius, xyz0, rays8, ro_cam)
        feats = torch.cat(pos_f + dir_f, dim=1)
        save("a4_features", dists=dists, nn=nn, radius=radius, ray_pos=xyz0, rays=rays8, ro=ro_cam,
             feats=feats, num_nn=num_nn2, standin_ball_query=1)

        # A8 composite on random rgbsigma
        rs = torch.rand(6, 64, 4, generator=g)
        rs[..., 3] = (rs[..., 3] - 0.3) * 30
        zz = torch.sort(torch.rand(6, 64, generator=g) * 4 + 9, -1)[0]
        rgb, depth, wts = rn.render_image(rs, zz, sel[:6], 0., True)
        rgb_nb, _, _ = rn.render_image(rs, zz, sel[:6], 0., False)
        save("a8_composite", rgbsigma=rs, z=zz, rays=sel[:6], rgb=rgb, depth=depth, weights=wts, rgb_nobg=rgb_nb)

        # A9 importance sampling
        zc = z0[:6].contiguous()
        w = torch.rand(6, 64, generator=g) ** 4
        w[0] = 0.0
        w[1, 20:30] = 0.5
        w[1, :20] = 0
        w[1, 30:] = 0
        xyz1, z1 = ray_utils.ImportanceSampling(zc, w, 128, sel[:6, :3], sel[:6, 3:], det=True)
        save("a9_importance", z0=zc, weights=w, rays=sel[:6], z1=z1, xyz1=xyz1)

        # A10 whole forward: 48 rays x 4913 particles
        out = rn(P, ro_cam, sel, focal, c2w)
        save("a10_forward", particles=P, rays=sel, ro=ro_cam, c2w=c2w, near=9.0, far=13.0, standin_ball_query=1,
             **{k: v for k, v in out.items()})

    # ---- C1 training-step loss + gradients (reference autograd) on 16 rays
    rn.zero_grad()
    tgt = torch.rand(16, 3, generator=g)
    out = rn(P, ro_cam, sel[:16], focal, c2w)
    loss = torch.nn.functional.mse_loss(out["rgb0"], tgt) + torch.nn.functional.mse_loss(out["rgb1"], tgt)
    loss.backward()
    gsel = {}
    for name in ["nerf_coarse.xyz_encoding_1.0.weight", "nerf_coarse.sigma.weight", "nerf_fine.xyz_encoding_5.0.weight",
                 "nerf_fine.dir_encoding.0.weight", "nerf_fine.rgb.0.bias", "nerf_fine.xyz_encoding_final.bias"]:
        p = dict(rn.named_parameters())[name]
        gsel["grad__" + name.replace(".", "__")] = p.grad.clone()
    gnorm = {("gnorm__" + n.replace(".", "__")): p.grad.norm() for n, p in rn.named_parameters()}
    save("c1_trainstep", particles=P, rays=sel[:16], target=tgt, loss=loss.detach(), standin_ball_query=1, **gsel, **gnorm)

    # ---- B1/B3 transition-model pieces that run without Open3D
    pn = ParticleNet(gravity=(0, 0, -9.81))
    pos = torch.randn(9, 3, generator=g)
    vel = torch.randn(9, 3, generator=g)
    p2, v2 = pn.integrate_pos_vel(pos, vel)
    delta = torch.randn(9, 3, generator=g) * 0.01
    p3, v3 = pn.update_pos_vel(pos, p2, delta)
    R = torch.linspace(-0.2, 1.3, 16)
    save("b1_integrate", pos=pos, vel=vel, gravity=pn.gravity, dt=pn.time_step, pos_new=p2, vel_new=v2, delta=delta,
         pos_corr=p3, vel_corr=v3, R=R, window=pn._window_poly6(R), filter_extent=float(pn.filter_extent))


if __name__ == "__main__":
    main()
