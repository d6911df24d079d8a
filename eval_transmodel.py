"""roll out the transition model only — same entry point as the reference's eval_transmodel.py."""
from configs import transmodel_config
from neurofluid_amd.trainers import TransModelEvaluation

if __name__ == '__main__':
    res = TransModelEvaluation(transmodel_config()).eval()
    print({k: (sum(v) / max(len(v), 1)) for k, v in res.items()})
