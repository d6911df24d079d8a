#!/usr/bin/env python3
"""Golden vectors of the renderer's NON-DEFAULT configurations (models/renderer.py:30-44, :96-109, :125-179): every
single-flag ablation of `encoding.{density,var,smoothed_pos,smoothed_dir}` (the reference's one published quality number,
BASELINE.md §1, comes from `wo-smoothed_dir`), all four off, `N_neighbor` 8 / 32, `(N_samples, N_importance)` = (32, 64) /
(64, 0), and the three `encoding.exclude_ray=False` branches of `smoothing_position` (:100-106: alpha by `num_nn.le(20)`
— always true at K = 20, both values at K = 32 — and `same_smooth_factor`).

For each variant the reference's own `RenderNet` (imported from /root/reference, same stand-ins as gen_golden.py) is
run on 24 rays through the synthetic water cube and the file `cfg_<name>.npz` records: the result dict of `forward`, and
the loss + the gradient norm of EVERY parameter + dL/d(particle positions) of the reference's autograd for
`mse(rgb0) + mse(rgb1)` (N_importance = 0: rgb0 only).  The ball-query stand-in used HERE returns `nn` as a
differentiable gather of `p2` (indices from the oracle's first-K search, which are data), so that the reference's
autograd reaches the particles through A3 / A4 exactly as it does through pytorch3d's op.
Runs only where /root/reference exists.  Usage:  python tests/golden/gen_golden_configs.py"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg      # noqa: E402

# name -> (encoding overrides, NN_search overrides, ray overrides[, top-level overrides])
VARIANTS = {
    "wo_density": (dict(density=False), {}, {}),
    "wo_var": (dict(var=False), {}, {}),
    "wo_smoothed_pos": (dict(smoothed_pos=False), {}, {}),
    "wo_smoothed_dir": (dict(smoothed_dir=False), {}, {}),
    "plain": (dict(density=False, var=False, smoothed_pos=False, smoothed_dir=False), {}, {}),
    "k8": ({}, dict(N_neighbor=8), {}),
    "k32": ({}, dict(N_neighbor=32), {}),
    "s32_64": ({}, {}, dict(N_samples=32, N_importance=64)),
    "s64_0": ({}, {}, dict(N_importance=0)),
    "incl_ray": (dict(exclude_ray=False), {}, {}),
    # use_mask=False: with the mask, a sample with num_nn <= 20 < K is zeroed and the alpha = 0.1 branch would never show
    "incl_ray_k32": (dict(exclude_ray=False), dict(N_neighbor=32), {}, dict(use_mask=False)),
    "incl_ray_same": (dict(exclude_ray=False, same_smooth_factor=True), dict(N_neighbor=32), {}),
}


def variant_cfg(name):
    enc, nns, ray, *top = VARIANTS[name]
    cfg = gg.renderer_cfg()
    cfg["encoding"].update(enc)
    cfg["NN_search"].update(nns)
    cfg["ray"].update(ray)
    for t in top:
        cfg.update(t)
    return cfg


def oracle_cfg(name):
    """The same variant as the flat dict the oracle takes (oracle/render_oracle.py DEFAULT_CFG)."""
    from oracle import render_oracle as ro
    enc, nns, ray, *top = VARIANTS[name]
    cfg = dict(ro.DEFAULT_CFG)
    for part in (enc, nns, ray, *top):
        cfg.update(part)
    return cfg


def install_differentiable_ball_query():
    """pytorch3d.ops.ball_query stand-in whose `nn` output is a differentiable gather of p2."""
    from oracle import neighbors
    ops = types.ModuleType("pytorch3d.ops")

    def ball_query(p1, p2, radius, K):
        ds, ids, nns = [], [], []
        for b in range(p1.shape[0]):
            d, i, _ = neighbors.ball_query_firstk(p1[b].detach().numpy(), p2[b].detach().numpy(), radius, K)
            i = torch.from_numpy(i)
            ds.append(torch.from_numpy(d))
            ids.append(i)
            nns.append(torch.where((i >= 0).unsqueeze(-1), p2[b][i.clamp(min=0)], torch.zeros(1)))
        return torch.stack(ds), torch.stack(ids), torch.stack(nns)

    ops.ball_query = ball_query
    sys.modules["pytorch3d"].ops = ops
    sys.modules["pytorch3d.ops"] = ops


def select_rays(ray_utils, ro):
    c2w = ro.eval_camera()
    focal = ro.camera_focal(400)
    dirs400 = ray_utils.get_ray_directions(400, 400, focal)
    o4, d4 = ray_utils.get_rays(dirs400, c2w)
    rays400 = torch.cat([o4, d4], -1)
    # 16 neighbouring pixels through the cube's middle, 6 down a column, 2 that miss the fluid
    return torch.cat([rays400[200, 184:200], rays400[150:174:4, 205], rays400[10, 0:2]], 0).contiguous(), c2w, focal


def main():
    assert os.path.isdir(gg.REF), "golden vectors can only be regenerated where /root/reference exists"
    gg.install_standins()
    install_differentiable_ball_query()
    sys.path.insert(0, gg.REF)
    from models.renderer import RenderNet      # noqa: E402  (reference code)
    from utils import ray_utils                # noqa: E402
    from oracle import render_oracle as ro

    sel, c2w, focal = select_rays(ray_utils, ro)
    tgt = torch.rand(sel.shape[0], 3, generator=torch.Generator().manual_seed(77))
    for name in VARIANTS:
        cfg = variant_cfg(name)
        ocfg = oracle_cfg(name)
        state = ro.deterministic_nerf_state(cfg=ocfg)
        rn = RenderNet(cfg, near=9.0, far=13.0)
        rn.load_state_dict(state, strict=True)
        P = ro.watercube_particles().clone().requires_grad_(True)
        ro_cam = rn.set_ro(c2w)
        out = rn(P, ro_cam, sel, focal, c2w)
        loss = torch.nn.functional.mse_loss(out["rgb0"], tgt)
        if "rgb1" in out:
            loss = loss + torch.nn.functional.mse_loss(out["rgb1"], tgt)
        loss.backward()
        gnorm = {("gnorm__" + n.replace(".", "__")): p.grad.norm() for n, p in rn.named_parameters() if p.grad is not None}
        dP = P.grad if P.grad is not None else torch.zeros_like(P)      # `plain`: no path to the particles
        gg.save("cfg_" + name, rays=sel, ro=ro_cam, target=tgt, loss=loss.detach(), dparticles=dP, standin_ball_query=1, **{k: v.detach() for k, v in out.items()}, **gnorm)


if __name__ == "__main__":
    main()
