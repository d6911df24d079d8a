"""How much does dL/d(particle positions) of the renderer (both passes) move when the positions move by ONE fp32 ulp?
Oracle only (CPU): calibrates the tolerance of tests/test_gpu_render.py::test_fine_pass_particle_gradients_vs_oracle_autograd
and of tests/test_gpu_coupled.py — the positional encodings multiply 1-ulp differences of the smoothed positions (HIP vs
CPU summation order) by up to 512 before the MLP sees them, and the gradient inherits that."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from oracle import render_oracle as ro  # noqa: E402
from test_gpu_render import _fluid_rays, _oracle_render_diff  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rays, roc = _fluid_rays(n)
tgt = torch.rand(rays.shape[0], 3, generator=torch.Generator().manual_seed(5))
st = ro.deterministic_nerf_state()
P0 = ro.watercube_particles()
z0, xyz0 = ro.coarse_sample_ray(9.0, 13.0, rays, 64)         # the fine depths, the way render_forward derives them
p0 = ro.render_pass(st, "nerf_coarse", P0, roc, rays, z0, xyz0, ro.DEFAULT_CFG)
_, z1 = ro.importance_sampling(z0, p0["weights"].detach(), 128, rays[:, :3], rays[:, 3:])


def grad(P):
    Pc = P.clone().requires_grad_(True)
    _oracle_render_diff(st, Pc, roc, rays, z1, tgt).backward()
    return Pc.grad


g0 = grad(P0)
gen = torch.Generator().manual_seed(0)
for trial in range(4):
    sign = (torch.randint(0, 2, P0.shape, generator=gen) * 2 - 1).float()
    g1 = grad(torch.nextafter(P0, P0 + sign))
    print(f"trial {trial}: relative change of dL/dpos under a 1-ulp move of every coordinate: "
          f"{float((g1 - g0).norm() / g0.norm()):.3e}  (touched {int((g0.abs().sum(1) > 0).sum())} particles)")
