"""200-frame coupled rollout (BASELINE config 5 shape: transition step + full-frame render per frame), dev tool:
frames/s, rays/s, allocator stability, finiteness.  usage: rollout_perf.py [frames] [side] [fp32|fp16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from neurofluid_amd.synthetic import watercube_scene
from neurofluid_amd.renderer import RenderNet
from neurofluid_amd.transmodel import ParticleNet
from neurofluid_amd.render_loop import render_image

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
side = int(sys.argv[2]) if len(sys.argv) > 2 else 400
mode = sys.argv[3] if len(sys.argv) > 3 else "fp32"
dev = torch.device("cuda:0")
scene = watercube_scene(side, side)
cfg = bench.renderer_cfg(); cfg["mlp_dtype"] = mode
net = RenderNet(cfg, 9.0, 13.0); net.load_state_dict(scene["nerf_state"]); net = net.to(dev)
pn = ParticleNet(gravity=(0, 0, -9.81)); pn.load_state_dict(scene["trans_state"]); pn = pn.to(dev)
pos, vel = scene["P"].to(dev), torch.zeros_like(scene["P"]).to(dev)
box, bn = scene["box"].to(dev), scene["bn"].to(dev)
rays = scene["rays"].to(dev); roc = scene["c2w"][:, 3].to(dev)
n = rays.shape[0]
torch.cuda.synchronize(); t0 = time.time(); marks = []
with torch.no_grad():
    for f in range(frames):
        pos, vel, _ = pn(pos, vel, box, bn)
        out = render_image(net, pos, n, roc, rays, None, None, iseval=True, ray_chunk=n)
        if f % 25 == 0 or f == frames - 1:
            torch.cuda.synchronize()
            ok = bool(torch.isfinite(out["pred_rgbs_1"]).all()) and bool(torch.isfinite(pos).all())
            marks.append((f, round(time.time() - t0, 2), round(torch.cuda.memory_reserved() / 2**30, 2),
                          round(float(out["mask_1"].sum()) / 1e6, 2), ok))
torch.cuda.synchronize(); dt = time.time() - t0
print(f"{mode} {side}x{side}: {frames} frames in {dt:.2f} s = {frames/dt:.1f} frames/s, {frames*n/dt/1e6:.2f} M rays/s")
print("(frame, t[s], reserved GiB, active fine rows [M], finite):", marks)
