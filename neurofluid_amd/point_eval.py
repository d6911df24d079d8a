"""FluidErrors (/root/reference/utils/point_eval.py:31-81): per-frame statistics of |pred - gt| and of the
gt -> nearest-prediction distance, in units of 1e-3.  CPU metric off the timed path (scipy cKDTree), kept for the
evaluators' reports; SURVEY §8f ranks a GPU version as future work."""
import json

import numpy as np
from scipy.spatial import cKDTree


def _stats(x):
    s = {'mean': np.mean(x), 'mse': np.mean(x ** 2), 'var': np.var(x), 'min': np.min(x), 'max': np.max(x),
         'median': np.median(x)}
    s = {k: float(v) * 1000 for k, v in s.items()}
    s['num_particles'] = x.shape[0]
    return s


class FluidErrors:
    def __init__(self):
        self.errors = {}

    def cal_errors(self, pred_pos, gt_pos, time_idx):
        if not np.isfinite(pred_pos).all() or not np.isfinite(gt_pos).all():
            print('positions contain nonfinite values')
            return None
        errs = _stats(np.linalg.norm(pred_pos - gt_pos, axis=-1))
        g2p, _ = cKDTree(pred_pos).query(gt_pos)
        errs.update({'gt2pred_' + k: v for k, v in _stats(g2p).items()})
        self.errors.setdefault(time_idx, {}).update(errs)
        return errs['gt2pred_mean']

    def save(self, path):
        with open(path, 'w') as f:
            json.dump(list(self.errors.items()), f, indent=4)

    def load(self, path):
        with open(path) as f:
            self.errors = {(tuple(k) if isinstance(k, list) else k): v for k, v in json.load(f)}
