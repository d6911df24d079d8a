"""Which host-side ops launch the small kernels of a train_e2e.py step (dev tool): torch.profiler with Python stacks,
ops grouped by the innermost neurofluid_amd / torch.optim frame.  usage: python tools/e2e_opcount.py [steps]"""
import collections, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import configs
from neurofluid_amd.datasets import write_synthetic_dataset
from neurofluid_amd.trainers import E2ETrainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
root = tempfile.mkdtemp(prefix="nf_e2e_")
write_synthetic_dataset(os.path.join(root, "data", "watercube"), n_frames=12, img=400, n_side=17)
cfg = configs.end2end_training_config(["--expdir", os.path.join(root, "exps"), "--expname", "perf", "--dataset", "watercube"])
ds = configs.dataset_config()["watercube"]
for split in ("train", "test"):
    ds[split].path = os.path.join(root, "data", "watercube")
    ds[split].start_index, ds[split].end_index = 0, 12
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
from torch.profiler import record_function
from neurofluid_amd import dist as nfdist


def rf(name, fn):
    def w(*a, **k):
        with record_function("PH:" + name):
            return fn(*a, **k)
    return w


def update_step(loss, global_step):          # E2ETrainer.update_step, phase by phase
    clip = tr.options.TRAIN.grad_clip_value
    with record_function("PH:zero_grad"):
        tr.optimizer.zero_grad()
        if tr.separate:
            tr.transition_optimizer.zero_grad()
    with record_function("PH:backward"):
        loss.backward()
    with record_function("PH:clip"):
        if clip != 0:
            torch.nn.utils.clip_grad_norm_(tr.renderer.parameters(), clip)
            torch.nn.utils.clip_grad_norm_(tr.transition_model.parameters(), clip)
    with record_function("PH:optim"):
        tr.optimizer.step()
        if tr.separate:
            tr.transition_optimizer.step()
        for sch in tr.schedulers:
            sch.step()


tr.update_step = update_step
tr.trainsition_step_for_training = rf("transition_fwd", tr.trainsition_step_for_training)
tr.renderer.forward = rf("render_fwd", tr.renderer.forward)
tr._frame_on_device = rf("frame_on_device", tr._frame_on_device)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.train(max_steps=steps)
    torch.cuda.synchronize()
by = collections.Counter()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU]
phases = sorted((e.time_range.start, e.time_range.end, e.name[3:]) for e in evs if e.name.startswith("PH:"))
fns = sorted((e.time_range.start, e.time_range.end, e.name) for e in evs if e.name.endswith("Backward") or e.name.endswith("Backward0"))
for ev in evs:
    if not ev.kernels or ev.name.startswith("PH:"):
        continue
    t = ev.time_range.start
    ph = [n for a, b, n in phases if a <= t <= b]
    node = [n for a, b, n in fns if a <= t <= b and n != ev.name]
    by[((ph[-1] if ph else "(outside)") + (" / " + node[-1][:30] if node else ""), ev.name[:40])] += len(ev.kernels)
for (frame, name), n in sorted(by.items(), key=lambda kv: -kv[1])[:70]:
    print("%6.1f /step  %-40s %s" % (n / steps, name, frame))
if os.environ.get("NF_OPSEQ"):           # the launch sequence of the last step, in host order
    last = [p_ for p_ in phases if p_[2] == "frame_on_device"][-1][0]
    for ev in sorted(evs, key=lambda e: e.time_range.start):
        if ev.kernels and not ev.name.startswith("PH:") and ev.time_range.start >= last:
            t = ev.time_range.start
            ph = [n for a, b, n in phases if a <= t <= b]
            node = [n for a, b, n in fns if a <= t <= b and n != ev.name]
            print("%9.0f %-16s %-28s %-28s %s" % (t - last, ph[-1] if ph else "-", node[-1][:28] if node else "", ev.name[:28],
                                              str(ev.input_shapes)[:70]))
print("total kernels per step: %.0f" % (sum(by.values()) / steps))
