"""Host-side phase timing of the warm-up training step (dev tool; thread sampler, no extra processes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from neurofluid_amd import train_step as ts
from neurofluid_amd.renderer import RenderNet

dev = torch.device("cuda:0")
scene = bench.build_scene(400)
net = RenderNet(bench.renderer_cfg(), 9.0, 13.0); net.load_state_dict(scene["nerf_state"]); net = net.to(dev)
H = W = 400
rays = scene["rays"].view(H, W, 6).to(dev); cw = scene["c2w"].to(dev)
g = torch.Generator().manual_seed(1)
views = [dict(cw=cw, rays=rays, rgb=torch.rand(H * W, 3, generator=g).to(dev)) for _ in range(4)]
P = scene["P"].to(dev)
for p in net.parameters(): p.requires_grad_(True)
opt = torch.optim.Adam(net.parameters(), lr=5e-4)
rng = np.random.RandomState(10)
mode = sys.argv[1] if len(sys.argv) > 1 else "thread"
sampler = None if mode == "inline" else ts.PixelSampler(rng, 4, 1024, lambda s: 160000, 1000)
marks = {}
def mark(name, t0):
    marks[name] = marks.get(name, 0.0) + time.perf_counter() - t0
N = 30
for it in range(N + 5):
    if it == 5:
        torch.cuda.synchronize(); marks.clear(); T0 = time.perf_counter()
    step = 1000 + it
    t = time.perf_counter(); sels = sampler.next(step) if sampler else [rng.choice(160000, size=[1024], replace=False) for _ in range(4)]; mark("sampler.next", t)
    t = time.perf_counter(); coords = ts.random_sample_coords(H, W, step, 500)
    sc_all = coords[torch.from_numpy(np.concatenate(sels))].long().to(dev); mark("coords+h2d", t)
    t = time.perf_counter()
    rays_l, rgbs_l, ro_l = [], [], []
    for vi, v in enumerate(views):
        sc = sc_all[vi * 1024:(vi + 1) * 1024]
        rays_l.append(v["rays"][sc[:, 0], sc[:, 1]]); rgbs_l.append(v["rgb"].view(H, W, -1)[sc[:, 0], sc[:, 1]])
        ro_l.append(net.set_ro(v["cw"]).expand(1024, 3))
    ro_c, rays_c = torch.cat(ro_l).contiguous(), torch.cat(rays_l); mark("gathers", t)
    t = time.perf_counter(); out = net(P, ro_c, rays_c, None, None); mark("forward (host)", t)
    t = time.perf_counter()
    total = 0.
    for i, rgbs in enumerate(rgbs_l):
        sl = slice(i * 1024, (i + 1) * 1024)
        total = total + torch.nn.functional.mse_loss(out["rgb0"][sl], rgbs) + torch.nn.functional.mse_loss(out["rgb1"][sl], rgbs)
    mark("loss", t)
    t = time.perf_counter(); opt.zero_grad(); total.backward(); mark("backward (host)", t)
    t = time.perf_counter(); opt.step(); mark("adam", t)
torch.cuda.synchronize()
tot = time.perf_counter() - T0
print(mode, "step %.2f ms" % (tot / N * 1e3))
for k, v in marks.items():
    print("  %-18s %7.2f ms" % (k, v / N * 1e3))
if sampler: sampler.close()
