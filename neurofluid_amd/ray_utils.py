"""Host-side mirror of /root/reference/utils/ray_utils.py for the functions the hot path calls
(get_ray_directions :85-104, get_rays :107-130, coarse_sample_ray :232-256, sample_pdf/ImportanceSampling :178-229).
Device versions run in libneurofluid_hip; the *_cpu helpers exist for dataset loading on hosts without a GPU
(they are plain torch and are NOT on the timed path)."""
import torch

from . import _lib
from ._lib import check, ptr


def get_ray_directions(H, W, focal):
    """(H,W,3) camera-space directions ((i-W/2)/f, -(j-H/2)/f, -1); torch ops on the default device."""
    xs = torch.linspace(0, W - 1, W)
    ys = torch.linspace(0, H - 1, H)
    j, i = torch.meshgrid(ys, xs, indexing="ij")
    return torch.stack([(i - W / 2) / focal, -(j - H / 2) / focal, -torch.ones_like(i)], -1)


def get_rays(directions, c2w):
    rays_d = directions @ c2w[:, :3].T
    rays_d = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
    return c2w[:, 3].expand(rays_d.shape), rays_d


def get_rays_cpu(H, W, focal, c2w):
    o, d = get_rays(get_ray_directions(H, W, focal), c2w)
    return torch.cat([o, d], -1)


def get_rays_device(H, W, focal, c2w, row0=0, nrows=None, device=None):
    """A0 on the GPU: rays (nrows*W, 6) for image rows [row0, row0+nrows) — lets every rank generate only its tiles."""
    lib = _lib.load()
    device = device or c2w.device
    nrows = H - row0 if nrows is None else nrows
    c = c2w.detach().to(device).contiguous().float()
    out = torch.empty(nrows * W, 6, dtype=torch.float32, device=device)
    check(lib.nf_get_rays(H, W, float(focal), ptr(c), row0, nrows, ptr(out), _lib.stream()), "nf_get_rays")
    return out


def get_rays_own_chunks(H, W, focal, c2w, ray_chunk, rank, world, device=None):
    """A0 for ONE RANK of a ray-chunk-sharded render (dist.my_chunks: chunk k -> rank k mod world): the rays of this rank's chunks in
    ownership order, generated on the device in one launch (nf_get_rays_chunks) — no (H*W, 6) tensor per rank, nothing scattered
    (SURVEY 8e).  Bit-identical to `get_rays_device(H, W, focal, c2w).index_select(0, own)`."""
    lib = _lib.load()
    device = device or c2w.device
    n_ray = H * W
    n_chunks = (n_ray + ray_chunk - 1) // ray_chunk
    own = list(range(rank, n_chunks, world))
    n_own = sum(min((k + 1) * ray_chunk, n_ray) - k * ray_chunk for k in own)
    c = c2w.detach().to(device).contiguous().float()
    out = torch.empty(n_own, 6, dtype=torch.float32, device=device)
    check(lib.nf_get_rays_chunks(H, W, float(focal), ptr(c), ray_chunk, rank, world, n_own, ptr(out), _lib.stream()), "nf_get_rays_chunks")
    return out
