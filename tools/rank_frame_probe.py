"""Dev tool: what ONE rank of an N-rank strong-scaling run does per frame (its chunks of the 400x400 image, replicated transition step, grid
rebuild; no collective), timed on one GPU — wall per frame next to the sum of its kernels' HIP-event times, to see whether a rank at N = 8
is bound by its GPU work or by the host that enqueues it.  usage: python tools/rank_frame_probe.py [world=8] [frames=30]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from neurofluid_amd.renderer import RenderNet
from neurofluid_amd.transmodel import ParticleNet
from neurofluid_amd.render_loop import render_image
from neurofluid_amd import synthetic

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
scene = bench.build_scene(400)
net = RenderNet(bench.renderer_cfg(), 9.0, 13.0); net.load_state_dict(scene["nerf_state"], strict=True); net = net.to(dev)
pn = ParticleNet(gravity=(0, 0, -9.81)); pn.load_state_dict(scene["trans_state"], strict=True); pn = pn.to(dev)
P0, box, bn = scene["P"].to(dev), scene["box"].to(dev), scene["bn"].to(dev)
roc = scene["c2w"][:, 3].to(dev)
cam = (400, 400, synthetic.camera_focal(400), scene["c2w"].to(dev))
for rank in (0, world - 1):
    st = {"pos": P0.clone(), "vel": torch.zeros_like(P0), "k": 0}

    def frame(tm=None):
        with torch.no_grad():
            if st["k"] % 8 == 0:
                st["pos"], st["vel"] = P0.clone(), torch.zeros_like(P0)
            st["k"] += 1
            st["pos"], st["vel"], _ = pn(st["pos"], st["vel"], box, bn)
            return render_image(net, st["pos"], 160000, roc, None, None, None, iseval=True, ray_chunk=1024, rank=rank, world=world,
                                gather=False, device_chunk=1 << 22, camera=cam, timings=tm) if world > 1 else \
                render_image(net, st["pos"], 160000, roc, None, None, None, iseval=True, ray_chunk=1024, gather=False, device_chunk=1 << 22, camera=cam, timings=tm)
    # gather=False at world > 1 still all-gathers rgb: this probe has no process group -> patch the gather away
    import neurofluid_amd.dist as nfd
    nfd.gather_chunks = lambda local, *a, **k: local
    for _ in range(8):
        frame()
    torch.cuda.synchronize()
    tm = {}
    t0 = time.perf_counter()
    for _ in range(frames):
        frame(tm)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    ev = sum(a.elapsed_time(b) for a, b in tm["render"]) / frames
    print(f"world {world} rank {rank}: {t_all / frames * 1e3:.2f} ms per frame wall, host loop {t_host / frames * 1e3:.2f} ms per frame, "
          f"render events {ev:.2f} ms per frame")
