"""Rollout parity of the transition model, HIP vs oracle (run on the GPU box):
  free:    both sides roll out on their own from the same initial state (the bar of BASELINE.json: mean L2 <= 1e-4)
  stepped: every frame the oracle steps from the HIP path's previous state (per-step error, no chaotic growth)
usage: python tools/rollout_parity.py [cloud=watercube|bunny|honeycone] [frames] [order]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from neurofluid_amd import synthetic  # noqa: E402
from neurofluid_amd.transmodel import ParticleNet  # noqa: E402
from oracle import trans_oracle as to  # noqa: E402

cloud = sys.argv[1] if len(sys.argv) > 1 else "watercube"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 50
order = sys.argv[3] if len(sys.argv) > 3 else "random"
dev = torch.device("cuda:0")
st = to.deterministic_transition_state()
pn = ParticleNet(gravity=(0, 0, -9.81))
pn.load_state_dict(st, strict=True)
pn = pn.to(dev)
P = synthetic.watercube_particles() if cloud == "watercube" else synthetic.shaped_particles(cloud, order=order)
box, bn = to.watercube_box()
boxd, bnd = box.to(dev), bn.to(dev)
p_h, v_h = P.to(dev), torch.zeros_like(P).to(dev)
p_o, v_o = P.clone(), torch.zeros_like(P)
t0 = time.time()
print(f"{cloud} ({order}) {P.shape[0]} particles, {frames} frames")
for f in range(frames):
    ph_prev, vh_prev = p_h.cpu(), v_h.cpu()
    with torch.no_grad():
        p_h, v_h, n_h = pn(p_h, v_h, boxd, bnd)
    p_o, v_o, n_o = to.particle_net_forward(st, p_o, v_o, box, bn)
    p_s, v_s, n_s = to.particle_net_forward(st, ph_prev, vh_prev, box, bn)
    free = float((p_h.cpu() - p_o).norm(dim=-1).mean())
    stepped = float((p_h.cpu() - p_s).norm(dim=-1).mean())
    smax = float((p_h.cpu() - p_s).norm(dim=-1).max())
    same = bool(torch.equal(n_h.cpu(), n_s))
    if f % 5 == 4 or f == frames - 1 or f < 3:
        print(f"frame {f + 1:3d}: free mean L2 {free:.3e}   stepped mean L2 {stepped:.3e} max {smax:.3e} "
              f"counts equal {same}  nbrs/particle {float(n_h.mean()):.1f}  zmin {float(p_h[:, 2].min()):.3f} "
              f"[{time.time() - t0:.0f} s]", flush=True)
