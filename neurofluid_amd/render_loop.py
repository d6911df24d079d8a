"""The chunk loop of /root/reference/trainer/basetrainer.py:264-309 (BaseTrainer.render_image), with
optional ray-chunk sharding across ranks (neurofluid_amd/dist.py)."""
import torch

from . import dist as nfdist


def render_image(renderer, particle_pos, N_ray, ro, rays, focal_length=None, cw=None, iseval=False, ray_chunk=1024,
                 rank=0, world=1, gather=True):
    """Same result dict as the reference: pred_rgbs_0/1 (N_ray,3), num_nn_0/1 (N_ray*S), mask_0/1 (N_ray,1) if iseval.
    With world > 1 every rank renders chunks k = rank, rank+world, ... and the RGB tiles are all-gathered;
    num_nn / mask stay local unless gather=True."""
    n_imp = renderer.N_importance
    n_chunks = (N_ray + ray_chunk - 1) // ray_chunk
    mine = nfdist.my_chunks(n_chunks, rank, world)
    keys = ["rgb0", "num_nn_0"] + (["mask_0"] if iseval else [])
    if n_imp > 0:
        keys += ["rgb1", "num_nn_1"] + (["mask_1"] if iseval else [])
    parts = {k: [] for k in keys}
    for k in mine:
        res = renderer(particle_pos, ro, rays[k * ray_chunk:(k + 1) * ray_chunk], focal_length, cw)
        for key in keys:
            v = res[key]
            parts[key].append(v.view(v.shape[0], -1) if key.startswith("num_nn") else v)
    names = {"rgb0": "pred_rgbs_0", "rgb1": "pred_rgbs_1"}
    ret = {}
    if world == 1:
        for key in keys:
            t = torch.cat(parts[key], dim=0)
            ret[names.get(key, key)] = t.reshape(-1) if key.startswith("num_nn") else t
        return ret
    share = nfdist.share_size(n_chunks, world)
    dev = rays.device
    for key in keys:
        if not gather and not key.startswith("rgb"):
            continue
        width = parts[key][0].shape[-1] if parts[key] else None
        if width is None:   # this rank owns no chunk: learn the width from the key
            S0, S1 = renderer.N_samples, renderer.N_samples + n_imp
            width = {"rgb0": 3, "rgb1": 3, "mask_0": 1, "mask_1": 1, "num_nn_0": S0, "num_nn_1": S1}[key]
        dtype = parts[key][0].dtype if parts[key] else (torch.int64 if key.startswith("num_nn") else torch.float32)
        local = torch.zeros(share * ray_chunk, width, dtype=dtype, device=dev)
        for s, t in enumerate(parts[key]):
            local[s * ray_chunk:s * ray_chunk + t.shape[0]] = t
        full = nfdist.gather_chunks(local, n_chunks, ray_chunk, N_ray, rank, world)
        ret[names.get(key, key)] = full.reshape(-1) if key.startswith("num_nn") else full
    return ret
