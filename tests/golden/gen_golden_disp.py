#!/usr/bin/env python3
"""Golden vectors of the renderer's optional modes: `noise_std > 0` (a10_noise.npz) and `use_disp=True` mode (coarse depths linear in disparity, utils/ray_utils.py:236-240),
recorded from the reference's own Python exactly like gen_golden.py (same stand-ins, same scene), in a file of its own so
that gen_golden.py keeps regenerating its fixtures bit for bit.  Runs only where /root/reference exists.
Usage:  python tests/golden/gen_golden_disp.py"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg      # noqa: E402


def main():
    assert os.path.isdir(gg.REF), "golden vectors can only be regenerated where /root/reference exists"
    gg.install_standins()
    sys.path.insert(0, gg.REF)
    from models.renderer import RenderNet      # noqa: E402  (reference code)
    from utils import ray_utils                # noqa: E402
    from oracle import render_oracle as ro

    c2w = ro.eval_camera()
    focal = ro.camera_focal(400)
    dirs400 = ray_utils.get_ray_directions(400, 400, focal)
    o4, d4 = ray_utils.get_rays(dirs400, c2w)
    rays400 = torch.cat([o4, d4], -1)
    sel = torch.cat([rays400[200, 180:196], rays400[150:182:4, 205]], 0).contiguous()        # 24 rays through the cube
    z, xyz = ray_utils.coarse_sample_ray(9.0, 13.0, sel[:5], 64, True, 0)
    state = ro.deterministic_nerf_state()
    P = ro.watercube_particles()
    rn = RenderNet(gg.renderer_cfg(), near=9.0, far=13.0)
    rn.load_state_dict(state, strict=True)
    ro_cam = rn.set_ro(c2w)
    with torch.no_grad():
        out = rn(P, ro_cam, sel, focal, c2w, use_disp=True)
    # noise_std > 0 (models/renderer.py:193-195): the reference draws torch.randn(sigmas.shape) once per pass from the global
    # generator — seeded here, so that the oracle can repeat the draws
    torch.manual_seed(4321)
    with torch.no_grad():
        outn = rn(P, ro_cam, sel, focal, c2w, noise_std=0.5)
    gg.save("a10_noise", particles=P, rays=sel, ro=ro_cam, seed=4321, noise_std=0.5, standin_ball_query=1,
            **{k: v for k, v in outn.items()})
    gg.save("a1_a10_disp", rays5=sel[:5], near=9.0, far=13.0, z=z.contiguous(), xyz=xyz, particles=P, rays=sel, ro=ro_cam,
            standin_ball_query=1, **{k: v for k, v in out.items()})


if __name__ == "__main__":
    main()
