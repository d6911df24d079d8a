import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from conftest import load_golden
from neurofluid_amd import ops
from oracle import render_oracle as ro
dev = torch.device("cuda:0")
g = load_golden("a4_features")
T = lambda a: torch.from_numpy(np.asarray(a))
P = ro.watercube_particles().to(dev)
rays = T(g["rays"]).to(dev)
feats = ops.debug_features(P, rays, 9.0, 13.0, 64, 0.225, 20, 15, T(g["ro"]).to(dev))
ref = T(g["feats"]).view(rays.shape[0], 64, -1)
rows = feats["row_sample"].cpu().long()
got = feats["features"].cpu().double()
r, s = rows // 64, rows % 64
d = (got - ref[r, s].double()).abs()
print("rows", rows.numel(), "max", float(d.max()), "mean", float(d.mean()))
# positional part of xyz (cols 0..62): x(3), then per freq sin(3),cos(3)
for k in (0, 5, 9):
    c0 = 3 + 6 * k
    print("freq", k, "xyz cols max", float(d[:, c0:c0 + 6].max()), "mean", float(d[:, c0:c0 + 6].mean()))
# recompute the reference PE from the golden's own x columns in float64 to separate PE error from input noise
x = got[:, 0:3].float()
for k in (0, 5, 9):
    c0 = 3 + 6 * k
    arg = (x * (2.0 ** k)).double()
    e = torch.cat([(got[:, c0:c0 + 3] - torch.sin(arg)).abs(), (got[:, c0 + 3:c0 + 6] - torch.cos(arg)).abs()], 1)
    e32 = torch.cat([(torch.sin((x * 2.0 ** k)).double() - torch.sin(arg)).abs(), (torch.cos((x * 2.0 ** k)).double() - torch.cos(arg)).abs()], 1)
    print("freq", k, "PE error vs exact: max %.3e mean %.3e | torch fp32 sin/cos vs exact: max %.3e" % (float(e.max()), float(e.mean()), float(e32.max())))
