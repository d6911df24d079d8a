"""dev: one training step from identical state, eager vs graph replay: which gradients / parameters differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neurofluid_amd import train_step as ts
from neurofluid_amd.renderer import RenderNet
from neurofluid_amd.synthetic import watercube_scene
dev = torch.device("cuda:0")
scene = watercube_scene(400, 400)
nets, steps = [], []
for graph in ("0", "1"):
    os.environ["NF_TRAIN_GRAPH"] = graph
    net = RenderNet(dict(use_mask=True, ray=dict(ray_chunk=1024, N_importance=128, N_samples=64),
                         NN_search=dict(fix_radius=True, particle_radius=0.025, search_raduis_scale=9.0, N_neighbor=20),
                         encoding=dict(density=True, var=True, smoothed_pos=True, smoothed_dir=True, exclude_ray=True, same_smooth_factor=False)), 9.0, 13.0)
    net.load_state_dict(scene["nerf_state"], strict=True)
    net = net.to(dev)
    step = ts.make_train_step(net, scene, dev)
    for _ in range(3):
        step()
    nets.append(net); steps.append(step)
torch.cuda.synchronize()
print("after 3 eager steps on both: max param diff", max(float((a - b).abs().max()) for (k, a), (_, b) in zip(nets[0].state_dict().items(), nets[1].state_dict().items())))
for it in range(2):
    l0 = steps[0](); l1 = steps[1]()
    if steps[1].graphed is not None:
        steps[1].graphed.verify()
    torch.cuda.synchronize()
    print("step", it, "loss", float(l0), float(l1), "captures", steps[1].graphed.captures)
    bad = 0
    for (k, a), (_, b) in zip(nets[0].named_parameters(), nets[1].named_parameters()):
        gd = float((a.grad - b.grad).abs().max()); pd = float((a.detach() - b.detach()).abs().max())
        if gd or pd:
            bad += 1
            print("  %-40s grad diff %.3e (|g| %.3e)  param diff %.3e" % (k, gd, float(a.grad.abs().max()), pd))
    print("  differing tensors:", bad)
