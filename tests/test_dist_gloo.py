"""world_size-2 gloo tests (CPU) of the multi-GPU host path: interleaved chunk sharding + tile all-gather of
render_image, and the data-parallel gradient all-reduce."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class FakeRenderer(torch.nn.Module):
    """Stands in for RenderNet on CPU: per-ray deterministic outputs, so sharded == unsharded can be checked."""
    N_importance = 128
    N_samples = 64

    def forward(self, particles, ro, rays, focal=None, cw=None):
        R = rays.shape[0]
        base = rays[:, :3].sum(1, keepdim=True)
        return {"rgb0": base.repeat(1, 3), "rgb1": base.repeat(1, 3) * 2,
                "num_nn_0": (base.long() % 7).view(R, 1, 1).repeat(1, 64, 1),
                "num_nn_1": (base.long() % 5).view(R, 1, 1).repeat(1, 192, 1),
                "mask_0": base, "mask_1": base + 1}


def _worker(rank, world, port, n_rays, chunk, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from neurofluid_amd import dist as nfdist
    from neurofluid_amd.render_loop import render_image
    r, w, _ = nfdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    rays = torch.arange(n_rays * 6, dtype=torch.float32).view(n_rays, 6)
    net = FakeRenderer()
    ref = render_image(net, None, n_rays, None, rays, iseval=True, ray_chunk=chunk)
    got = render_image(net, None, n_rays, None, rays, iseval=True, ray_chunk=chunk, rank=rank, world=world)
    ok = all(torch.equal(ref[k], got[k]) for k in ref) and set(ref) == set(got)
    # gradient all-reduce = mean over ranks
    p = torch.nn.Parameter(torch.zeros(5))
    p.grad = torch.full((5,), float(rank + 1))
    nfdist.allreduce_grads([p], world)
    ok = ok and torch.allclose(p.grad, torch.full((5,), (1 + world) / 2))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_rays,chunk", [(1000, 128), (1024, 256), (130, 64)])
def test_sharded_render_image_equals_unsharded(n_rays, chunk):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_rays, chunk, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]
