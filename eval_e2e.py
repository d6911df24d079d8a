"""roll out + render the test frames — same entry point as the reference's eval_e2e.py.  Launch with
`python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 eval_e2e.py ...` to shard ray chunks over N GPUs."""
from configs import dataset_config, end2end_training_config
from neurofluid_amd.trainers import E2EEvaluator

if __name__ == '__main__':
    cfg = end2end_training_config()
    cfg.update(dataset_config()[cfg.dataset])
    res = E2EEvaluator(cfg).eval()
    print({k: (sum(v) / max(len(v), 1)) for k, v in res.items()})
