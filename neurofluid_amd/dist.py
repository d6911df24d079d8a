"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).

The renderer shards by RAY CHUNK (the seam is the chunk loop, /root/reference/trainer/basetrainer.py:282-289):
chunk k of an image goes to rank k mod world (interleaved, because the fluid covers a minority of the pixels and
with sample compaction the cost follows the active samples: contiguous eighths give 3.4x load imbalance,
interleaving 1.02x — SURVEY §8d).  Particles stay replicated (59 KB); the only data-path collective is the
final all-gather of the rendered tiles.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* when launched by torch.distributed.run.
    Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def my_chunks(n_chunks, rank, world):
    """Interleaved chunk ownership: k -> rank k mod world."""
    return list(range(rank, n_chunks, world))


def share_size(n_chunks, world):
    return (n_chunks + world - 1) // world


def gather_chunks(local, n_chunks, chunk, total_rows, rank, world):
    """local: (share*chunk, C) rows of this rank's chunks in ownership order (padded with zeros).
    Returns the (total_rows, C) tensor in original ray order on every rank (all-gather)."""
    share = share_size(n_chunks, world)
    C = local.shape[-1]
    assert local.shape[0] == share * chunk
    if world == 1:
        return local[:total_rows]
    out = torch.empty(world * share * chunk, C, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    # out[r, s, i] holds chunk (s*world + r), row i  ->  reorder to chunk-major
    out = out.view(world, share, chunk, C).permute(1, 0, 2, 3).reshape(share * world * chunk, C)
    return out[:total_rows]


def allreduce_grads(params, world):
    """Data-parallel training over rays: one flat-bucket all-reduce (5.35 MB for the renderer)."""
    if world == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= world
    o = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[o:o + n].view_as(g))
        o += n


def mean_over_ranks_hook(world):
    """Tensor hook for the e2e step: the ranks render different rays of the SAME predicted particles, so dL/d(pred_pos)
    (Np x 3 fp32 = 59 KB) is all-reduced (mean) BEFORE it is back-propagated into the replicated transition model — whose
    692 902-parameter gradients are then identical on every rank and need no all-reduce of their own (2.8 MB saved per step).
    Usage: pred_pos.register_hook(mean_over_ranks_hook(world)); allreduce_grads() then takes the renderer's parameters only."""
    def hook(g):
        if world == 1:
            return g
        g = g.contiguous().clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        return g / world
    return hook


def replicas_in_sync(params, world, resync_from=0):
    """Checks that a module every rank updates WITHOUT a gradient all-reduce (the e2e trainer's transition model: its only upstream
    gradient is averaged by mean_over_ranks_hook, so equal replicas stay equal only as long as its forward and backward are bitwise
    deterministic on every rank — no float atomics, no rank-local summation order) still holds the same bits everywhere.
    An exact, order-free checksum of the parameter bits (int64 sum of the int32 words) is all-reduced as (max, -min); on a mismatch
    the parameters are broadcast from `resync_from` (None: leave them).  Returns True when the replicas agreed.  One 16-byte
    collective: called at checkpoint time (trainers.E2ETrainer), not per step."""
    params = [p for p in params]
    if world == 1 or not params:
        return True
    dev = params[0].device
    cs = torch.zeros((), dtype=torch.int64, device=dev)
    for p in params:
        cs = cs + p.detach().contiguous().view(-1).view(torch.int32).to(torch.int64).sum()
    pair = torch.stack([cs, -cs])
    dist.all_reduce(pair, op=dist.ReduceOp.MAX)
    same = bool((pair[0] == -pair[1]).item())
    if not same and resync_from is not None:
        for p in params:
            dist.broadcast(p.data, src=resync_from)
    return same
