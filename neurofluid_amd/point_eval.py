"""FluidErrors (/root/reference/utils/point_eval.py:31-81): per-frame statistics of |pred - gt| and of the
gt -> nearest-prediction distance, in units of 1e-3.

The reference moves both clouds to the host and runs scipy's cKDTree every frame inside the rollout loops
(eval_e2e.py:86, eval_transmodel.py:102-106).  Here the clouds stay on the device: the nearest-neighbour
distances come from the exact brute-force kernel (nf_nearest), the statistics are reduced on the device in
float64, and one small vector of scalars crosses to the host per frame."""
import json

import numpy as np
import torch

from . import ops

_KEYS = ('mean', 'mse', 'var', 'min', 'max', 'median')


def _stats_vec(x):
    """x: (n,) float64 device tensor -> 6 statistics (numpy conventions: population variance, median = mean of
    the two middle values for even n)."""
    n = x.shape[0]
    srt = torch.sort(x).values
    med = (srt[(n - 1) // 2] + srt[n // 2]) * 0.5
    return torch.stack([x.mean(), (x * x).mean(), x.var(unbiased=False), srt[0], srt[-1], med])


def _as_device(a, device):
    t = torch.as_tensor(a) if not torch.is_tensor(a) else a
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


class FluidErrors:
    def __init__(self, device=None):
        self.errors = {}
        self.device = device

    def cal_errors(self, pred_pos, gt_pos, time_idx):
        dev = self.device
        if dev is None:
            dev = pred_pos.device if torch.is_tensor(pred_pos) and pred_pos.is_cuda else torch.device("cuda")
        pred, gt = _as_device(pred_pos, dev), _as_device(gt_pos, dev)
        err = (pred - gt).double().norm(dim=-1)
        g2p = ops.nearest(pred, gt)
        vec = torch.cat([_stats_vec(err), _stats_vec(g2p), torch.stack([pred.isfinite().all(), gt.isfinite().all()]).double()])
        vec = vec.cpu().numpy()                      # the one host transfer of the frame
        if vec[12] == 0 or vec[13] == 0:
            print('positions contain nonfinite values')
            return None
        errs = {k: float(v) * 1000 for k, v in zip(_KEYS, vec[:6])}
        errs['num_particles'] = int(pred.shape[0])
        errs.update({'gt2pred_' + k: float(v) * 1000 for k, v in zip(_KEYS, vec[6:12])})
        errs['gt2pred_num_particles'] = int(gt.shape[0])
        self.errors.setdefault(time_idx, {}).update(errs)
        return errs['gt2pred_mean']

    def save(self, path):
        with open(path, 'w') as f:
            json.dump(list(self.errors.items()), f, indent=4)

    def load(self, path):
        with open(path) as f:
            self.errors = {(tuple(k) if isinstance(k, list) else k): v for k, v in json.load(f)}
