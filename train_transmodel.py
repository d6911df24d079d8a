"""supervised fine-tuning of the transition model — same entry point as the reference's train_transmodel.py."""
from configs import transmodel_config
from neurofluid_amd.trainers import TransModelTrainer

if __name__ == '__main__':
    TransModelTrainer(transmodel_config()).train()
