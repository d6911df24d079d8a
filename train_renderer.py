"""warm-up of the renderer — same entry point and flags as the reference's train_renderer.py
(`python train_renderer.py --dataset watercube [--config ...] [--expdir ...] [--expname ...] [--resume_from ...]`)."""
from configs import dataset_config, warmup_training_config
from neurofluid_amd.trainers import RendererTrainer

if __name__ == '__main__':
    cfg = warmup_training_config()
    cfg.update(dataset_config()[cfg.dataset])
    print(cfg.dump())
    RendererTrainer(cfg).train()
