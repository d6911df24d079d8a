"""Renders the bench's 400x400 frame N times with a given MLP path (dev tool; run under tools/prof.sh for kernel stats).
usage: python tools/frame_prof.py [fp32|fp16|split] [frames] [watercube400|honeycone800|bunny800]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from neurofluid_amd.renderer import RenderNet
from neurofluid_amd.render_loop import render_image

dtype = sys.argv[1] if len(sys.argv) > 1 else "fp16"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 10
scene_name = sys.argv[3] if len(sys.argv) > 3 else "watercube400"
side = 800 if scene_name.endswith("800") else 400
dev = torch.device("cuda:0")
sc = bench.build_scene(side)
cfg = bench.renderer_cfg(); cfg["mlp_dtype"] = dtype
net = RenderNet(cfg, 9.0, 13.0); net.load_state_dict(sc["nerf_state"], strict=True); net = net.to(dev)
P0, roc, rays = sc["P"].to(dev), sc["c2w"][:, 3].to(dev), sc["rays"].to(dev)
if side == 800:
    from neurofluid_amd import synthetic
    P0 = synthetic.shaped_particles(scene_name[:-3], order="random").to(dev)


def frame():
    with torch.no_grad():
        net.invalidate_grid()
        return render_image(net, P0, rays.shape[0], roc, rays, None, None, iseval=True, ray_chunk=rays.shape[0], gather=False,
                            device_chunk=1 << 22)


for _ in range(3):
    frame()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(frames):
    frame()
torch.cuda.synchronize()
print("%s: %.3f ms per frame" % (dtype, (time.perf_counter() - t0) / frames * 1e3))
