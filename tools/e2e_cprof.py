import os, sys, tempfile, time, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import configs
from neurofluid_amd.datasets import write_synthetic_dataset
from neurofluid_amd.trainers import E2ETrainer
steps = 30
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
tr.train(max_steps=len(tr.dataset))
torch.cuda.synchronize()
tr.start_step = 0
pr = cProfile.Profile(); pr.enable()
t0 = time.time()
tr.train(max_steps=steps)
torch.cuda.synchronize()
pr.disable()
print("ms/step", (time.time() - t0) / steps * 1e3)
pstats.Stats(pr).sort_stats("tottime").print_stats(38)
