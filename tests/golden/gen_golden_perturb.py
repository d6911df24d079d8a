#!/usr/bin/env python3
"""Golden vectors of `perturb > 0` (models/renderer.py:225, :250; utils/ray_utils.py:186-190, :247-253): what the reference's own
RenderNet.forward returned for seeded draws — perturb alone, and perturb together with noise_std (which interleaves the four
draws of a call: coarse jitter, coarse noise, inverse-CDF u, fine noise).  Same stand-ins and scene as gen_golden.py, a file of
its own so that the other generators keep regenerating their fixtures bit for bit.  Runs only where /root/reference exists.
Usage:  python tests/golden/gen_golden_perturb.py"""
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
    state = ro.deterministic_nerf_state()
    P = ro.watercube_particles()
    rn = RenderNet(gg.renderer_cfg(), near=9.0, far=13.0)
    rn.load_state_dict(state, strict=True)
    ro_cam = rn.set_ro(c2w)
    torch.manual_seed(777)
    zc, _ = ray_utils.coarse_sample_ray(9.0, 13.0, sel[:5], 64, False, 0.75)
    torch.manual_seed(1234)
    with torch.no_grad():
        out = rn(P, ro_cam, sel, focal, c2w, perturb=1.0)
    torch.manual_seed(4321)
    with torch.no_grad():
        outn = rn(P, ro_cam, sel, focal, c2w, perturb=0.5, noise_std=0.25)
    gg.save("a10_perturb", particles=P, rays=sel, ro=ro_cam, seed=1234, perturb=1.0, standin_ball_query=1,
            seed_coarse=777, perturb_coarse=0.75, z_coarse=zc.contiguous(),
            **{k: v for k, v in out.items()})
    gg.save("a10_perturb_noise", particles=P, rays=sel, ro=ro_cam, seed=4321, perturb=0.5, noise_std=0.25, standin_ball_query=1,
            **{k: v for k, v in outn.items()})


if __name__ == "__main__":
    main()
