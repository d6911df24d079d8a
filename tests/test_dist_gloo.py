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
    # reference chunks dealt to the ranks, several of them fused into one renderer call per rank
    got = render_image(net, None, n_rays, None, rays, iseval=True, ray_chunk=chunk, rank=rank, world=world, device_chunk=3 * chunk)
    ok = ok and all(torch.equal(ref[k], got[k]) for k in ref)
    got = render_image(net, None, n_rays, None, rays, iseval=True, ray_chunk=chunk, device_chunk=1 << 20)
    ok = ok and all(torch.equal(ref[k], got[k]) for k in ref)
    # gradient all-reduce = mean over ranks
    p = torch.nn.Parameter(torch.zeros(5))
    p.grad = torch.full((5,), float(rank + 1))
    nfdist.allreduce_grads([p], world)
    ok = ok and torch.allclose(p.grad, torch.full((5,), (1 + world) / 2))
    # replicas that are kept equal by determinism alone (the e2e trainer's transition model): equal bits pass, a drifted replica is seen
    # on EVERY rank and re-seeded from rank 0
    m = torch.nn.Linear(4, 3)
    with torch.no_grad():
        for t in m.parameters():
            t.copy_(torch.arange(t.numel(), dtype=torch.float32).view_as(t) * 0.25)
    ok = ok and nfdist.replicas_in_sync(m.parameters(), world)
    if rank == world - 1:
        with torch.no_grad():
            m.bias[1] += 1e-7 * 4           # one ulp-sized difference on one rank
    ok = ok and not nfdist.replicas_in_sync(m.parameters(), world)
    ok = ok and nfdist.replicas_in_sync(m.parameters(), world) and float(m.bias[1]) == 0.25
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_rays,chunk,world", [(1000, 128, 2), (1024, 256, 2), (130, 64, 2), (1000, 128, 3), (130, 64, 4)])
def test_sharded_render_image_equals_unsharded(n_rays, chunk, world):
    """(world 3: an odd rank count, ragged last chunk; world 4 with 3 chunks: one rank owns NO chunk and still takes part in the gather)"""
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
    assert sorted(res) == [(r, True) for r in range(world)]


def _dp_worker(rank, world, port, q):
    """Data-parallel training semantics of BaseTrainer (trainers.py): identical model init on every rank, DIFFERENT
    pixel selections per rank (seed + rank), gradient all-reduce -> identical weights after the step."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import numpy as np
    from neurofluid_amd import dist as nfdist
    from neurofluid_amd.trainers import BaseTrainer
    from configs import Node
    tr = BaseTrainer.__new__(BaseTrainer)            # the constructor needs a GPU; the seeding logic does not
    tr.rank, tr.world, tr.local_rank = nfdist.init_from_env(backend="gloo")
    tr.options = Node({"TRAIN": {"precrop_iters": 0}})
    tr.seed_everything(10)
    model = torch.nn.Linear(6, 3)                    # "init_fn": built from the common seed
    w0 = model.weight.detach().clone()
    tr.seed_data_streams()
    H = W = 32
    rays = torch.arange(H * W * 6, dtype=torch.float32).view(H, W, 6) / (H * W * 6)
    rgbs = torch.rand(H * W, 3, generator=torch.Generator().manual_seed(0))
    sel_rays, sel_rgb = tr.sample_pixels(rays, rgbs, H, W, global_step=5, ray_chunk=64)
    first_pixel = float(sel_rays[0, 0])
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    loss = torch.nn.functional.mse_loss(model(sel_rays), sel_rgb)
    opt.zero_grad()
    loss.backward()
    nfdist.allreduce_grads(list(model.parameters()), world)
    opt.step()
    gathered = [None] * world
    dist.all_gather_object(gathered, (w0, model.weight.detach().clone(), first_pixel, sel_rays.clone()))
    same_init = all(torch.equal(g[0], gathered[0][0]) for g in gathered)
    same_final = all(torch.equal(g[1], gathered[0][1]) for g in gathered)
    different_pixels = not torch.equal(gathered[0][3], gathered[1][3])
    moved = not torch.equal(gathered[0][0], gathered[0][1])
    q.put((rank, bool(same_init and same_final and different_pixels and moved)))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_ranks_draw_different_pixels_same_weights():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_single_process_keeps_reference_stream():
    """world == 1: seed_data_streams leaves the reference's single np.random stream untouched."""
    import numpy as np
    from neurofluid_amd.trainers import BaseTrainer
    tr = BaseTrainer.__new__(BaseTrainer)
    tr.rank, tr.world = 0, 1
    tr.seed_everything(10)
    tr.seed_data_streams()
    a = np.random.choice(1000, size=[8], replace=False)
    np.random.seed(10)
    assert np.array_equal(a, np.random.choice(1000, size=[8], replace=False))


def _e2e_hook_worker(rank, world, port, q):
    """SURVEY 8e, e2e training: all-reducing dL/d(pred_pos) (a tensor hook, 59 KB at 4 913 particles) in front of the REPLICATED
    transition backward gives every rank the gradients that all-reducing the transition model's own parameter gradients gives."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from neurofluid_amd import dist as nfdist
    nfdist.init_from_env(backend="gloo")
    torch.manual_seed(3)                                  # identical replicas
    trans = torch.nn.Sequential(torch.nn.Linear(3, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3))      # "transition model"
    rend = torch.nn.Linear(3, 3)                          # "renderer"
    pos = torch.randn(50, 3, generator=torch.Generator().manual_seed(1))
    sel = torch.randperm(50, generator=torch.Generator().manual_seed(100 + rank))[:20]               # this rank's "rays"
    tgt = torch.rand(20, 3, generator=torch.Generator().manual_seed(200 + rank))

    def loss_of(pred):
        return torch.nn.functional.mse_loss(rend(pred[sel]), tgt) + 0.1 * pred.abs().mean()          # rgb loss + a boundary-like term

    # (a) the scheme of rounds 1-4: all-reduce every parameter gradient
    for p in list(trans.parameters()) + list(rend.parameters()):
        p.grad = None
    loss_of(trans(pos)).backward()
    nfdist.allreduce_grads(list(trans.parameters()) + list(rend.parameters()), world)
    ref = [p.grad.clone() for p in list(trans.parameters()) + list(rend.parameters())]
    # (b) hook on pred_pos + renderer-only all-reduce
    for p in list(trans.parameters()) + list(rend.parameters()):
        p.grad = None
    pred = trans(pos)
    pred.register_hook(nfdist.mean_over_ranks_hook(world))
    loss_of(pred).backward()
    nfdist.allreduce_grads(list(rend.parameters()), world)
    got = [p.grad.clone() for p in list(trans.parameters()) + list(rend.parameters())]
    ok = all(torch.allclose(a, b, rtol=1e-5, atol=1e-7) for a, b in zip(ref, got))
    gathered = [None] * world
    dist.all_gather_object(gathered, [g.clone() for g in got])
    ok = ok and all(torch.equal(a, b) for a, b in zip(gathered[0], gathered[1]))       # identical on every rank, without a second all-reduce
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_e2e_dpos_allreduce_equals_parameter_allreduce():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_e2e_hook_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]
