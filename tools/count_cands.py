import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from neurofluid_amd import ops
from neurofluid_amd.renderer import RenderNet
dev = torch.device("cuda:0")
scene = bench.build_scene(dev)
net = RenderNet(bench.renderer_cfg(), 9.0, 13.0); net.load_state_dict(scene["nerf_state"]); net = net.to(dev)
P = scene["P"].to(dev); rays = scene["rays"].to(dev); roc = scene["c2w"][:, 3].to(dev)
from neurofluid_amd.autograd import _run_passes
with torch.no_grad():
    p0, p1, *_ = _run_passes(net, P, roc, rays, True, True, False)
print("coarse: samples", p0.R * p0.S, "cand", int(p0.counters[0]), "active", int(p0.counters[1]))
print("fine:   samples", p1.R * p1.S, "cand", int(p1.counters[0]), "active", int(p1.counters[1]))
nn = p1.num_nn.view(-1)
import numpy as np
h = torch.bincount(nn.clamp(max=20).long(), minlength=21).cpu().numpy()
print("fine num_nn histogram:", h.tolist())
