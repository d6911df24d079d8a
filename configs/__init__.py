"""Config surface of the reference (`/root/reference/configs/__init__.py`): the same five CLI flags
(:10-15), the same factory names, the same yaml keys — re-implemented as a small attribute-dict over PyYAML
(yacs is not a dependency).  Behavioural differences, on purpose: `os.makedirs(..., exist_ok=True)` (the reference
crashes when the experiment directory exists, :85) and `update()` works after freezing (the entry points call it
on a frozen node, train_e2e.py:15-16)."""
import argparse
import os

import yaml

_HERE = os.path.dirname(os.path.realpath(__file__))


class Node(dict):
    """dict with attribute access, recursive; mirrors the bits of yacs.CfgNode the code base uses."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = Node(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def merge_from_file(self, path):
        """`_include: other.yaml` (relative to the including file) is merged first, then the file's own keys."""
        with open(path) as f:
            d = yaml.safe_load(f) or {}
        inc = d.pop('_include', None)
        for name in ([inc] if isinstance(inc, str) else (inc or [])):
            self.merge_from_file(os.path.join(os.path.dirname(os.path.abspath(path)), name))
        self.merge(d)

    def merge(self, d):
        for k, v in d.items():
            if isinstance(v, dict) and isinstance(self.get(k), Node):
                self[k].merge(v)
            else:
                self[k] = Node(v) if isinstance(v, dict) else v

    def update(self, d=None, **kw):
        self.merge(dict(d or {}, **kw))

    def freeze(self):
        return self

    def clone(self):
        return Node(self.to_dict())

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, Node) else v) for k, v in self.items()}

    def dump(self):
        return yaml.safe_dump(self.to_dict(), default_flow_style=False)


def make_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--expdir', type=str, default='exps', help='experiment dir')
    p.add_argument('--expname', type=str, default='debug', help='experiment name')
    p.add_argument('--dataset', type=str, default='', help='dataset')
    p.add_argument('--config', type=str, default='', help='default config file')
    p.add_argument('--resume_from', type=str, default='', help='path of ckpt to be load')
    return p


def default_config():
    return Node()


def get_config(config_file, merge=True):
    cfg = Node()
    cfg.merge_from_file(config_file)
    return cfg


def save_config(cfg, savepath):
    with open(savepath, 'w') as f:
        f.write(cfg.dump())


def dataset_config():
    return get_config(os.path.join(_HERE, 'dataset.yaml'))


def _training_config(default_yaml, argv=None):
    args = vars(make_parser().parse_args(argv))
    cfg = Node()
    cfg.merge_from_file(args['config'] or os.path.join(_HERE, default_yaml))
    cfg.update(args)
    out = os.path.join(args['expdir'], args['expname'])
    os.makedirs(out, exist_ok=True)
    save_config(cfg, os.path.join(out, 'config.yaml'))
    return cfg


def end2end_training_config(argv=None):
    return _training_config('end2end.yaml', argv)


def warmup_training_config(argv=None):
    return _training_config('warmup.yaml', argv)


def transmodel_config(argv=None):
    return _training_config('transmodel.yaml', argv)
