"""Timing of ParticleNet.forward alone on the synthetic watercube (dev tool): particle-steps/s.
usage: tools/trans_perf.py [steps] [unfused|split|grid|all_pairs]     (steps from the initial cloud, as bench.py measures: the rate depends
on the state of the rollout — a cloud that has fallen and piled up has more neighbours per particle)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neurofluid_amd.synthetic import watercube_scene
from neurofluid_amd.transmodel import ParticleNet

dev = torch.device("cuda:0")
scene = watercube_scene(8, 8)
pn = ParticleNet(gravity=(0, 0, -9.81))
pn.load_state_dict(scene["trans_state"], strict=True)
pn = pn.to(dev)
P0 = scene["P"].to(dev)
box, bn = scene["box"].to(dev), scene["bn"].to(dev)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
mode = sys.argv[2] if len(sys.argv) > 2 else "fp32"
# "frozen" as a third argument: every step starts from the SAME state (P0, zero velocities).  That is what A/B builds that corrupt
# the step's OUTPUT (ablation switches) must be timed with: in a rollout their garbage positions change the neighbour lists — and with
# them the work — of every following step
frozen = len(sys.argv) > 3 and sys.argv[3] == "frozen"
if mode == "unfused":
    pn.fused_inference = False
elif mode == "split":
    pn.conv_arith = "split"
elif mode in ("grid", "all_pairs"):
    pn.fused_search = mode
for it in range(3):
    pos, vel = P0.clone(), torch.zeros_like(P0)
    torch.cuda.synchronize(); t = time.time()
    with torch.no_grad():
        for _ in range(steps):
            if frozen:
                pn(pos, vel, box, bn)
            else:
                pos, vel, _ = pn(pos, vel, box, bn)
    torch.cuda.synchronize(); dt = time.time() - t
    print(f"iter {it} [{mode}{' frozen' if frozen else ''}]: {dt/steps*1e6:.1f} us/step, {P0.shape[0]*steps/dt/1e6:.2f} M particle-steps/s  overflows {getattr(pn, 'fused_overflows', 0)}")
