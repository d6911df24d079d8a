import sys, os, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from neurofluid_amd.renderer import RenderNet
from neurofluid_amd.train_step import make_train_step
dev = torch.device("cuda:0")
scene = bench.build_scene(dev)
net = RenderNet(bench.renderer_cfg(), 9.0, 13.0); net.load_state_dict(scene["nerf_state"]); net = net.to(dev)
step = make_train_step(net, scene, dev)
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
