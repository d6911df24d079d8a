import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from neurofluid_amd.renderer import RenderNet
from neurofluid_amd.transmodel import ParticleNet
from neurofluid_amd.render_loop import render_image
dev = torch.device("cuda:0")
scene = bench.build_scene(dev)
net = RenderNet(bench.renderer_cfg(), 9.0, 13.0); net.load_state_dict(scene["nerf_state"]); net = net.to(dev)
pn = ParticleNet(gravity=(0, 0, -9.81)); pn.load_state_dict(scene["trans_state"]); pn = pn.to(dev)
P0 = scene["P"].to(dev); box, bn = scene["box"].to(dev), scene["bn"].to(dev)
rays = scene["rays"].to(dev); roc = scene["c2w"][:, 3].to(dev)
pos, vel = P0.clone(), torch.zeros_like(P0)
ts = []
for i in range(14):
    torch.cuda.synchronize(); t = time.perf_counter()
    with torch.no_grad():
        pos, vel, _ = pn(pos, vel, box, bn)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        out = render_image(net, P0, rays.shape[0], roc, rays, None, None, iseval=True, ray_chunk=160000)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append(((t1 - t) * 1e3, (t2 - t1) * 1e3))
print(" ".join(f"{a:.2f}+{b:.1f}" for a, b in ts))
