"""Timing of the train_e2e.py step (BASELINE configs[2]) at the watercube scale on a synthetic on-disk dataset:
4913 particles, 400x400 images, the config's view count x ray_chunk rays per step (dev tool)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import configs
from neurofluid_amd.datasets import write_synthetic_dataset
from neurofluid_amd.trainers import E2ETrainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
root = tempfile.mkdtemp(prefix="nf_e2e_")
write_synthetic_dataset(os.path.join(root, "data", "watercube"), n_frames=steps + 6, img=400, n_side=17)
cfg = configs.end2end_training_config(["--expdir", os.path.join(root, "exps"), "--expname", "perf", "--dataset", "watercube"])
ds = configs.dataset_config()["watercube"]
for split in ("train", "test"):
    ds[split].path = os.path.join(root, "data", "watercube")
    ds[split].start_index, ds[split].end_index = 0, steps + 6
cfg.update(ds)
for node in (cfg.TRAIN, cfg.TEST):
    node.imgW = node.imgH = 400
cfg.TRAIN.save_interval = 10 ** 9
cfg.TRAIN.epochs = 10
tr = E2ETrainer(cfg)
tr.keep_frame_cache = True      # steady state of a multi-epoch run: frames stay on the device between train() calls
print("views", tr.train_view_names, "ray_chunk", cfg.RENDERER.ray.ray_chunk, "frames", len(tr.dataset))
tr.train(max_steps=len(tr.dataset))        # warm-up: one pass over every frame (the dataset caches decoded frames)
torch.cuda.synchronize()
blocks = []
for _ in range(5):          # median of 5 blocks (a block is `steps` steps from the start of an epoch)
    tr.start_step = 0
    t0 = time.time()
    tr.train(max_steps=steps)
    torch.cuda.synchronize()
    blocks.append((time.time() - t0) / steps)
print("blocks (ms/step):", [round(b * 1e3, 2) for b in blocks])
dt = sorted(blocks)[2]
nv = len(tr.train_view_names)
print(f"train_e2e step: {dt*1e3:.2f} ms  ({nv} views x {cfg.RENDERER.ray.ray_chunk} rays + transition fwd/bwd, "
      f"{nv * cfg.RENDERER.ray.ray_chunk / dt:.0f} rays/s)")

# ---- phase breakdown (synchronised timers; adds sync overhead)
import collections
acc = collections.defaultdict(float)
def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t = time.time()
        r = fn(*a, **k)
        torch.cuda.synchronize(); acc[name] += time.time() - t
        return r
    return w
ds_get = tr.dataset.__class__.__getitem__
tr.dataset.__class__.__getitem__ = timed("dataset[i]", ds_get)
tr._to_dev = timed("to_dev", tr._to_dev)
tr.trainsition_step_for_training = timed("transition fwd", tr.trainsition_step_for_training)
tr.render_image = timed("render fwd", tr.render_image)
tr.sample_pixels = timed("sample_pixels", tr.sample_pixels)
tr.update_step = timed("backward+optim", tr.update_step)
tr.start_step = 0
t0 = time.time()
tr.train(max_steps=steps)
torch.cuda.synchronize()
tot = time.time() - t0
for k, v in acc.items():
    print(f"  {k:18s} {v/steps*1e3:8.2f} ms/step")
print(f"  total              {tot/steps*1e3:8.2f} ms/step")
