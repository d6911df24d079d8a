"""render GT particle frames with a warm-up checkpoint — same entry point as the reference's eval_renderer.py."""
from configs import warmup_training_config
from neurofluid_amd.trainers import RendererEvaluation

if __name__ == '__main__':
    RendererEvaluation(warmup_training_config()).eval()
