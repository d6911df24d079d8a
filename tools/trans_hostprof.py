"""Host-side profile of the fused transition step (dev tool): where do the microseconds between launches go?"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neurofluid_amd.synthetic import watercube_scene
from neurofluid_amd.transmodel import ParticleNet
dev = torch.device("cuda:0")
scene = watercube_scene(8, 8)
pn = ParticleNet(gravity=(0, 0, -9.81)); pn.load_state_dict(scene["trans_state"], strict=True); pn = pn.to(dev)
P0 = scene["P"].to(dev); box, bn = scene["box"].to(dev), scene["bn"].to(dev)
pos, vel = P0.clone(), torch.zeros_like(P0)
with torch.no_grad():
    for _ in range(20): pos, vel, _ = pn(pos, vel, box, bn)
    torch.cuda.synchronize()
    # host-only time: enqueue 200 steps without waiting
    t = time.perf_counter()
    for _ in range(200): pos, vel, _ = pn(pos, vel, box, bn)
    th = time.perf_counter() - t
    torch.cuda.synchronize(); tt = time.perf_counter() - t
    print(f"host enqueue {th/200*1e6:.1f} us/step, wall {tt/200*1e6:.1f} us/step")
    # the C call alone (8 launches + event record), GPU idle before each call
    import ctypes
    from neurofluid_amd import _lib
    st = pn._fused
    lib = _lib.load()
    nn = torch.empty(P0.shape[0], device=dev); pc, vc = torch.empty_like(pos), torch.empty_like(pos)
    ts = []
    for _ in range(50):
        torch.cuda.synchronize()
        t = time.perf_counter()
        lib.nf_trans_step(st["Sref"], pos.data_ptr(), vel.data_ptr(), nn.data_ptr(), pc.data_ptr(), vc.data_ptr(), st["flag_dev"],
                          12345, torch.cuda.current_stream().cuda_stream)
        ts.append(time.perf_counter() - t)
    torch.cuda.synchronize()
    print(f"nf_trans_step host time: median {sorted(ts)[25]*1e6:.1f} us")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): pos, vel, _ = pn(pos, vel, box, bn)
    pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
