"""Wall time of the warm-up training step (bench.py's train workload) and the host-side share of it (dev tool).
usage: python tools/train_hostprof.py [fused]   — 'fused' builds Adam with fused=True"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from neurofluid_amd import train_step as ts
from neurofluid_amd.renderer import RenderNet

dev = torch.device("cuda:0")
scene = bench.build_scene(400)
net = RenderNet(bench.renderer_cfg(), 9.0, 13.0); net.load_state_dict(scene["nerf_state"]); net = net.to(dev)
if len(sys.argv) > 1 and sys.argv[1] == "fused":
    _Adam = torch.optim.Adam
    torch.optim.Adam = lambda params, **kw: _Adam(params, fused=True, **kw)
if os.environ.get("NF_FAKE_DRAW") == "1":      # what-if: a pixel draw that costs nothing (NOT the reference's stream)
    class _Cheap:
        def __init__(self, rng): self.rng = rng
        def get_state(self): return self.rng.get_state()
        def set_state(self, s): self.rng.set_state(s)
        def choice(self, n, size, replace=False): return self.rng.randint(0, n, size=size)
    _PS = ts.PixelSampler
    ts.PixelSampler = lambda rng, *a, **k: _PS(_Cheap(rng), *a, **k)
if os.environ.get("NF_ONE_STREAM") == "1":
    from neurofluid_amd import autograd_bwd
    autograd_bwd.TWO_STREAM_BACKWARD = False
step = ts.make_train_step(net, scene, dev)
for _ in range(8):
    step()
torch.cuda.synchronize()
N = 40
t0 = time.perf_counter(); host = 0.0
for _ in range(N):
    t = time.perf_counter(); step(); host += time.perf_counter() - t
torch.cuda.synchronize()
print("%s: step %.2f ms wall, %.2f ms inside step() on the host (incl. the wait for its row counts)" %
      (sys.argv[1] if len(sys.argv) > 1 else "default", (time.perf_counter() - t0) / N * 1e3, host / N * 1e3))
if os.environ.get("NF_CPROFILE") == "1":
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(40):
        step()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
