"""end-to-end training (transition model + renderer) — same entry point and flags as the reference's train_e2e.py."""
from configs import dataset_config, end2end_training_config
from neurofluid_amd.trainers import E2ETrainer

if __name__ == '__main__':
    cfg = end2end_training_config()
    cfg.update(dataset_config()[cfg.dataset])
    print(cfg.dump())
    E2ETrainer(cfg).train()
