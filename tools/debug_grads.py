import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import load_golden
from test_gpu_render import make_net, T
dev = torch.device("cuda:0")
g = load_golden("c1_trainstep")
net = make_net(dev)
P, rays, tgt = T(g["particles"], dev), T(g["rays"], dev), T(g["target"], dev)
roc = T(load_golden("a10_forward")["ro"], dev)
out = net(P, roc, rays, None, None)
loss = torch.nn.functional.mse_loss(out["rgb0"], tgt) + torch.nn.functional.mse_loss(out["rgb1"], tgt)
loss.backward()
print("loss", float(loss), float(g["loss"]))
params = dict(net.named_parameters())
for key, ref in sorted(g.items()):
    if key.startswith("gnorm__"):
        name = key[len("gnorm__"):].replace("__", ".")
        gn, rn = float(params[name].grad.norm()), float(ref)
        print(f"{name:45s} gnorm {gn:.6e} ref {rn:.6e} rel {abs(gn-rn)/max(rn,1e-30):.2e}")
for key, ref in sorted(g.items()):
    if key.startswith("grad__"):
        name = key[len("grad__"):].replace("__", ".")
        got = params[name].grad.cpu(); ref = T(ref)
        d = (got - ref).abs()
        print(f"{name:45s} maxdiff {float(d.max()):.3e} scale {float(ref.abs().max()):.3e} shape {tuple(ref.shape)} argmax {np.unravel_index(int(d.argmax()), ref.shape)}")
        if "encoding_5" in name:
            print("   xyz-part maxdiff", float(d[:, :198].max()), " h-part maxdiff", float(d[:, 198:].max()))
