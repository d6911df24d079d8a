"""The chunk loop of /root/reference/trainer/basetrainer.py:264-309 (BaseTrainer.render_image), with
optional ray-chunk sharding across ranks (neurofluid_amd/dist.py)."""
import torch

from . import dist as nfdist
from .autograd import LazyResults

_OWN_IDX = {}


def _own_ray_index(n_chunks, ray_chunk, N_ray, rank, world, device):
    """Indices of the rays of this rank's chunks (k = rank, rank + world, ...) in ownership order; cached."""
    key = (n_chunks, ray_chunk, N_ray, rank, world, str(device))
    idx = _OWN_IDX.get(key)
    if idx is None:
        parts = [torch.arange(k * ray_chunk, min((k + 1) * ray_chunk, N_ray)) for k in nfdist.my_chunks(n_chunks, rank, world)]
        idx = (torch.cat(parts) if parts else torch.zeros(0, dtype=torch.int64)).to(device)
        if len(_OWN_IDX) > 64:
            _OWN_IDX.clear()
        _OWN_IDX[key] = idx
    return idx


def _fit_device_chunk(renderer, device_chunk, ray_chunk, device):
    """Rays per fused call, bounded by what the device can hold.  With `use_mask` only active samples get row buffers (their number is
    data: the OutOfMemoryError path of render_image handles a call that does not fit); WITHOUT the mask every sample is a row
    (1 KB of MLP operand + the 8 + 4 K bytes of its row lists + 21 B of per-sample arrays), so the size is known up front and the
    call is cut to 60 % of the free memory.  A bound learnt from an earlier failure (`_device_chunk_limit`) is kept."""
    device_chunk = min(device_chunk, int(getattr(renderer, "_device_chunk_limit", device_chunk)))
    if device.type == "cuda" and not getattr(renderer, "use_mask", True):
        S1 = renderer.N_samples + renderer.N_importance         # the arena is reused by the passes: the larger one counts
        per_ray = S1 * (1024 + 8 + 4 * renderer.num_neighbor + 21) + 64
        free, _ = torch.cuda.mem_get_info(device)
        fit = int(free * 0.6) // per_ray // ray_chunk * ray_chunk
        device_chunk = min(device_chunk, max(ray_chunk, fit))
    return max(int(device_chunk), 1)


def render_image(renderer, particle_pos, N_ray, ro, rays, focal_length=None, cw=None, iseval=False, ray_chunk=1024,
                 rank=0, world=1, gather=True, device_chunk=None, camera=None, timings=None):
    """Same result dict as the reference: pred_rgbs_0/1 (N_ray,3), num_nn_0/1 (N_ray*S), mask_0/1 (N_ray,1) if iseval.

    ray_chunk    the reference's chunk (the unit of its loop, and the unit that is dealt to the ranks);
    device_chunk rays per fused renderer call (default: ray_chunk).  Rays are independent and results are
                 chunk-independent bit for bit (tests/test_gpu_render.py), so several reference chunks go into one call.
    With world > 1 the chunks are interleaved over the ranks (chunk k -> rank k mod world: the fluid covers a minority
    of the pixels, so the cost follows the active samples; fine interleaving balances it) and every rank renders ALL
    its chunks in as few fused calls as device_chunk allows — 1024-ray balance at full-GPU launch sizes.  The RGB
    tiles are all-gathered; num_nn / mask stay local unless gather=True.

    camera = (H, W, focal, c2w) with rays=None: every rank GENERATES the rays of its own chunks on the device (nf_get_rays_chunks,
    SURVEY 8e: no ray tensor is materialised per rank, nothing is scattered) — bit-identical to indexing the full tensor.
    timings: a dict that receives ("render", "gather") pairs of torch.cuda.Event (start, end) of this call, for the per-rank
    breakdown of `bench.py --gpus N`."""
    n_imp = renderer.N_importance
    n_chunks = (N_ray + ray_chunk - 1) // ray_chunk
    device_chunk = max(int(device_chunk or ray_chunk), 1)
    keys = ["rgb0", "num_nn_0"] + (["mask_0"] if iseval else [])
    if n_imp > 0:
        keys += ["rgb1", "num_nn_1"] + (["mask_1"] if iseval else [])
    per_ray_ro = ro is not None and ro.dim() == 2          # fused multi-view calls carry one camera position per ray
    ev = None
    if timings is not None:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ev[0].record()
    if rays is None:
        from . import ray_utils
        H_, W_, focal_, c2w_ = camera
        assert H_ * W_ == N_ray and not per_ray_ro
        my_rays = ray_utils.get_rays_own_chunks(H_, W_, focal_, c2w_, ray_chunk, rank, world, device=particle_pos.device)
        my_ro = ro
    elif world > 1:
        own = _own_ray_index(n_chunks, ray_chunk, N_ray, rank, world, rays.device)
        my_rays = rays.index_select(0, own)
        my_ro = ro.index_select(0, own) if per_ray_ro else ro
    else:
        my_rays, my_ro = rays[:N_ray], ro
    parts = {k: [] for k in keys}
    device_chunk = _fit_device_chunk(renderer, device_chunk, ray_chunk, my_rays.device)
    i = 0
    while i < my_rays.shape[0]:
        try:
            res = renderer(particle_pos, my_ro[i:i + device_chunk] if per_ray_ro else my_ro, my_rays[i:i + device_chunk],
                           focal_length, cw)
        except torch.cuda.OutOfMemoryError:
            # the row buffers of a fused call scale with its ACTIVE samples, which only the device knows: a call that does not
            # fit is redone at half the size (results are chunk-independent bit for bit) and the module remembers the bound
            if device_chunk <= ray_chunk:
                raise
            device_chunk = max(ray_chunk, device_chunk // 2 // ray_chunk * ray_chunk)
            renderer._device_chunk_limit = device_chunk
            if getattr(renderer, "_workspace", None) is not None:
                renderer._workspace = None          # drop the arena that grew towards the failed size
            torch.cuda.empty_cache()
            continue
        i += device_chunk
        for key in keys:
            raw = res.raw_int32(key) if key.startswith("num_nn") and hasattr(res, "raw_int32") else None
            if raw is not None:             # the kernels' int32 counts: widened to the reference's int64 when first read
                parts[key].append(raw[0].view(raw[1][0], -1))
                continue
            v = res[key]
            parts[key].append(v.view(v.shape[0], -1) if key.startswith("num_nn") else v)
    names = {"rgb0": "pred_rgbs_0", "rgb1": "pred_rgbs_1"}
    ret = LazyResults()
    if ev is not None:
        ev[1].record()
        timings.setdefault("render", []).append((ev[0], ev[1]))
    if world == 1:
        for key in keys:
            t = parts[key][0] if len(parts[key]) == 1 else torch.cat(parts[key], dim=0)     # one fused call: no copy
            if key.startswith("num_nn") and t.dtype == torch.int32:
                ret.set_lazy(key, t, (t.numel(),))
            else:
                ret[names.get(key, key)] = t.reshape(-1) if key.startswith("num_nn") else t
        return ret
    share = nfdist.share_size(n_chunks, world)
    dev = my_rays.device
    S0, S1 = renderer.N_samples, renderer.N_samples + n_imp
    widths = {"rgb0": 3, "rgb1": 3, "mask_0": 1, "mask_1": 1, "num_nn_0": S0, "num_nn_1": S1}
    for key in keys:
        if not gather and not key.startswith("rgb"):
            continue
        # the slab's dtype must not depend on what THIS rank rendered (a rank that owns no chunk of a small image has nothing to look at,
        # and every rank must bring the same bytes to the all-gather): neighbour counts travel as int32, everything else as fp32
        dtype = torch.int32 if key.startswith("num_nn") else torch.float32
        # own chunks in ownership order; only the image's last chunk can be ragged and it is the last of its owner, so
        # the rendered rows are a prefix of this rank's (share * ray_chunk)-row slab
        local = torch.zeros(share * ray_chunk, widths[key], dtype=dtype, device=dev)
        if parts[key]:
            t = parts[key][0] if len(parts[key]) == 1 else torch.cat(parts[key], dim=0)
            local[:t.shape[0]] = t                  # (copy_ converts an int64 count / another float type)
        full = nfdist.gather_chunks(local, n_chunks, ray_chunk, N_ray, rank, world)
        if key.startswith("num_nn") and full.dtype == torch.int32:
            ret.set_lazy(key, full, (full.numel(),))
        else:
            ret[names.get(key, key)] = full.reshape(-1) if key.startswith("num_nn") else full
    if ev is not None:
        ev[2].record()
        timings.setdefault("gather", []).append((ev[1], ev[2]))
    return ret
