"""How much do the fine-net gradients move when the particle positions move by ONE fp32 ulp?  (dev tool: calibrates the
tolerance of tests/test_gpu_render.py::test_fine_net_grads_same_samples — the positional encodings multiply such
noise by up to 512 before it reaches the MLP.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import load_golden
from neurofluid_amd.renderer import RenderNet
from oracle import render_oracle as ro
import bench
dev = torch.device("cuda:0")
g = load_golden("c1_trainstep")
T = lambda a: torch.from_numpy(np.asarray(a)).to(dev)
P, rays, tgt = T(g["particles"]), T(g["rays"]), T(g["target"])
roc = T(load_golden("a10_forward")["ro"])
def grads(Pp):
    net = RenderNet(bench.renderer_cfg(), 9.0, 13.0); net.load_state_dict(ro.deterministic_nerf_state()); net = net.to(dev)
    out = net(Pp, roc, rays, None, None)
    torch.nn.functional.mse_loss(out["rgb1"], tgt).backward()
    return {k: p.grad.clone() for k, p in net.named_parameters() if k.startswith("nerf_fine")}, out["mask_1"].clone()
g0, m0 = grads(P)
gen = torch.Generator().manual_seed(0)
for trial in range(3):
    sign = (torch.randint(0, 2, P.shape, generator=gen) * 2 - 1).to(dev).float()
    Pp = torch.nextafter(P, P + sign)                      # every coordinate one ulp up or down
    g1, m1 = grads(Pp)
    same = bool(torch.equal(m0, m1))
    worst = max((float((g1[k] - g0[k]).norm() / g0[k].norm()), k) for k in g0)
    print(f"trial {trial}: masks identical {same}; worst relative gradient change {worst[0]:.2e} at {worst[1]}")
