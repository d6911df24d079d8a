"""C1-C3 on the GPU: the trainer / evaluator counterparts run end-to-end on a tiny synthetic dataset written in the
reference's on-disk format (32x32 images, 343 particles)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    from neurofluid_amd.datasets import write_synthetic_dataset
    root = tmp_path_factory.mktemp("nf")
    write_synthetic_dataset(str(root / "data" / "watercube"), n_frames=4, img=32, n_side=7)
    return root


def _cfg(workdir, factory, name):
    import configs
    cfg = factory(["--expdir", str(workdir / "exps"), "--expname", name, "--dataset", "watercube"])
    ds = configs.dataset_config()["watercube"]
    for split in ("train", "test"):
        ds[split].path = str(workdir / "data" / "watercube")
        ds[split].start_index, ds[split].end_index = 0, 4
    cfg.update(ds)
    for node in (cfg.TRAIN, cfg.TEST):
        node.imgW = node.imgH = 32
    cfg.RENDERER.ray.ray_chunk = 128
    cfg.RENDERER.device_ray_chunk = 512
    cfg.TRAIN.save_interval = 2
    return cfg


def test_renderer_trainer_and_eval(workdir):
    import configs
    from neurofluid_amd.trainers import RendererTrainer, RendererEvaluation
    cfg = _cfg(workdir, configs.warmup_training_config, "warm")
    cfg.TRAIN.N_iters = 4
    tr = RendererTrainer(cfg)
    before = [p.detach().clone() for p in tr.renderer.parameters()]
    loss = tr.train()
    assert np.isfinite(float(loss))
    # steps 0-2 ran eagerly (they learn the row capacities), step 3 was the first replay of the captured step
    assert tr._graph_step is not None and tr._graph_step.captures >= 1 and tr._graph_step.steps_total >= 1
    assert any(not torch.equal(a, b.detach()) for a, b in zip(before, tr.renderer.parameters()))
    ck = workdir / "exps" / "warm" / "models" / "3.pt"
    assert ck.exists()
    sd = torch.load(ck)
    assert set(sd) == {"step", "renderer_state_dict", "optimizer_state_dict"}
    assert "nerf_coarse.xyz_encoding_1.0.weight" in sd["renderer_state_dict"]
    # resume + full-image evaluation from the fixed camera (eval_renderer.py)
    cfg2 = _cfg(workdir, configs.warmup_training_config, "warm_eval")
    cfg2.resume_from = str(ck)
    cfg2.TEST.data_path = str(workdir / "data" / "watercube" / "view_1" / "train" / "particles")
    cfg2.TEST.end_index = 2
    ev = RendererEvaluation(cfg2)
    out = ev.eval(max_frames=2)
    assert len(out) == 2 and out[0]["pred_rgbs_1"].shape == (32 * 32, 3)
    assert (workdir / "exps" / "warm_eval" / "render_GT").exists()


def test_e2e_trainer_and_evaluator(workdir):
    import configs
    from neurofluid_amd.trainers import E2ETrainer, E2EEvaluator
    cfg = _cfg(workdir, configs.end2end_training_config, "e2e")
    cfg.TRAIN.epochs = 1
    tr = E2ETrainer(cfg)
    t_before = [p.detach().clone() for p in tr.transition_model.parameters()]
    loss = tr.train(max_steps=3)
    assert np.isfinite(float(loss))
    changed = sum(not torch.equal(a, b.detach()) for a, b in zip(t_before, tr.transition_model.parameters()))
    assert changed >= 10, "transition-model parameters must receive gradients through the renderer"
    tr.save_checkpoint(7)
    cfg2 = _cfg(workdir, configs.end2end_training_config, "e2e_eval")
    cfg2.resume_from = str(workdir / "exps" / "e2e" / "models" / "7.pt")
    res = E2EEvaluator(cfg2).eval()
    assert len(res["pred2gt"]) == 3 and all(np.isfinite(res["pred2gt"])) and len(res["psnr"]) == 6
    assert (workdir / "exps" / "e2e_eval" / "images" / "fine" / "view_5" / "Pred" / "00001.png").exists()


def test_graph_replayed_e2e_step_equals_eager(workdir):
    """trainer_e2e.py's step replayed as ONE HIP graph (e2e_graph.GraphedE2EStep: transition forward, render, loss, backward through both
    models, both parameter groups' Adam, the carried state) against the eager step from the same seed: after 11 steps over a 3-frame
    sequence (the state restarts at frame 0 inside the replayed range) every parameter of both models, the carried state and the
    last loss are BIT-equal; the graph did replay (steps 0-2 are eager: they learn the capacities)."""
    import configs
    from neurofluid_amd.trainers import E2ETrainer

    def run(name, graph):
        cfg = _cfg(workdir, configs.end2end_training_config, name)
        cfg.TRAIN.epochs = 10
        cfg.TRAIN.save_interval = 6         # an evaluation + checkpoint in the middle of the replayed range
        cfg.TRAIN.log_interval = 3          # logged steps verify (and read the step's own prediction) before the next replay
        cfg.TRAIN.e2e_graph = graph
        tr = E2ETrainer(cfg)
        loss = tr.train(max_steps=11)
        torch.cuda.synchronize()
        return tr, float(loss)
    ea, le = run("e2e_eager", False)
    gr, lg = run("e2e_graph", True)
    g = gr._graph_step
    assert g is not None and g.captures >= 1 and g.steps_total >= 6, (g and (g.captures, g.steps_total))
    assert ea.__dict__.get("_graph_step") is None
    assert le == lg
    for m in ("renderer", "transition_model"):
        for (name, a), (_, b) in zip(getattr(ea, m).named_parameters(), getattr(gr, m).named_parameters()):
            assert torch.equal(a.detach(), b.detach()), (m, name)
    assert torch.equal(ea.pos_for_next_step, gr.pos_for_next_step) and torch.equal(ea.vel_for_next_step, gr.vel_for_next_step)
    # a capacity the graph was captured with is exceeded: the poisoned steps are redone, the trajectory is still the eager one
    gr.renderer.train_row_cap = {k: max(64, v // 64) for k, v in gr.renderer.train_row_cap.items()}
    g._recapture = True
    gr.train(max_steps=4)              # (all four through the graph: the eager warm-up steps are per trainer, not per call)
    ea.train(max_steps=4)
    torch.cuda.synchronize()
    assert g.redone_steps >= 1
    for m in ("renderer", "transition_model"):
        for (name, a), (_, b) in zip(getattr(ea, m).named_parameters(), getattr(gr, m).named_parameters()):
            assert torch.equal(a.detach(), b.detach()), ("after redo", m, name)


def test_transmodel_eval_and_train(workdir):
    import configs
    from neurofluid_amd.trainers import TransModelEvaluation, TransModelTrainer
    cfg = configs.transmodel_config(["--expdir", str(workdir / "exps"), "--expname", "trans"])
    cfg.TEST.datapath = str(workdir / "data" / "watercube")
    cfg.TEST.end_index = 4
    res = TransModelEvaluation(cfg).eval()       # BASELINE config 1: 2..3-step rollout harness
    assert len(res["pred2gt"]) == 3 and all(np.isfinite(res["pred2gt"]))
    assert (workdir / "exps" / "trans" / "res.json").exists()
    cfg.TRAIN.datapath.train = cfg.TRAIN.datapath.eval = str(workdir / "data" / "watercube")
    cfg.TRAIN.end_index = 4
    cfg.TRAIN.N_iters = 2                      # EPOCHS over the 2 three-frame windows (trainer_transmodel.py:167-168)
    cfg.TRAIN.save_interval = 1
    cfg.TRAIN.grad_clip_value = 1.0
    tr = TransModelTrainer(cfg)
    assert len(tr.dataset) == 2 and len(tr.test_dataset) == 2
    # the loss of one sample, numerically: 0.5*wmse_1 + 0.5*wmse_2 + boundary_1 + boundary_2 (:179-189) vs the oracle
    from oracle import trans_oracle as to
    item = tr.test_dataset[0]                  # not rotated
    with torch.no_grad():
        loss, parts = tr.sample_loss(tr._to_dev(item))
    st = {k: v.detach().cpu() for k, v in tr.transition_model.state_dict().items()}
    q1, w1, m1 = to.particle_net_forward(st, item["particles_pos_0"], item["particles_vel_0"], item["box"], item["box_normals"])
    q2, w2, m2 = to.particle_net_forward(st, q1, w1, item["box"], item["box_normals"])

    def wmse(pred, gt, n):
        return torch.mean(torch.exp(-n / 40) * torch.sqrt(torch.sum((pred - gt) ** 2, -1) + 1e-12) ** 0.5)

    def bnd(p):
        r = 0.025
        c = torch.stack((p[:, 0].clamp(-1 + r, 1 - r), p[:, 1].clamp(-1 + r, 1 - r), p[:, 2].clamp(-1 + r, 2.4552 - r)), 1)
        return torch.nn.functional.l1_loss(p, c)

    ref = 0.5 * wmse(q1, item["particles_pos_1"], m1) + 0.5 * wmse(q2, item["particles_pos_2"], m2) + bnd(q1) + bnd(q2)
    assert abs(float(loss) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref))), (float(loss), float(ref))
    before = [p.detach().clone() for p in tr.transition_model.parameters()]
    loss = tr.train()
    assert np.isfinite(float(loss))
    assert any(not torch.equal(a, b.detach()) for a, b in zip(before, tr.transition_model.parameters()))
    # one checkpoint per epoch, named by the global step, 'step' = epoch index; rollout dumps of eval()
    ck = workdir / "exps" / "trans" / "models" / "4.pt"
    assert ck.exists() and (workdir / "exps" / "trans" / "models" / "2.pt").exists()
    sd = torch.load(ck)
    assert set(sd) == {"step", "model_state_dict", "optimizer_state_dict"} and sd["step"] == 1
    assert (workdir / "exps" / "trans" / "particles" / "4" / "pred_1.obj").exists()
    # resume restores model AND optimizer state (trainer_transmodel.py:111-115)
    cfg.resume_from = str(ck)
    tr2 = TransModelTrainer(cfg)
    st2 = tr2.optimizer.state_dict()["state"]
    assert len(st2) > 0 and int(next(iter(st2.values()))["step"]) == 4
    for a, b in zip(tr.transition_model.parameters(), tr2.transition_model.parameters()):
        assert torch.equal(a.detach(), b.detach())


def _sharded_eval_worker(rank, world, port, root, q):
    """One rank of a 2-rank job on ONE GPU (gloo): real HIP kernels, real sharding / gather control flow."""
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch
    import torch.distributed as dist
    import configs
    from neurofluid_amd import dist as nfdist
    nfdist.init_from_env(backend="gloo")
    from neurofluid_amd.trainers import E2EEvaluator
    cfg = configs.end2end_training_config(["--expdir", os.path.join(root, "exps"), "--expname", f"shard_r{rank}", "--dataset", "watercube"])
    ds = configs.dataset_config()["watercube"]
    for split in ("train", "test"):
        ds[split].path = os.path.join(root, "data", "watercube")
        ds[split].start_index, ds[split].end_index = 0, 3
    cfg.update(ds)
    for node in (cfg.TRAIN, cfg.TEST):
        node.imgW = node.imgH = 48
    cfg.RENDERER.ray.ray_chunk = 128          # 2304 rays = 18 chunks -> 9 per rank
    cfg.RENDERER.device_ray_chunk = 512
    ev = E2EEvaluator(cfg)
    assert (ev.rank, ev.world) == (rank, world)
    torch.manual_seed(0)                      # identical replicas
    for p in list(ev.renderer.parameters()) + list(ev.transition_model.parameters()):
        torch.nn.init.normal_(p, std=0.05) if p.dim() > 1 else torch.nn.init.zeros_(p)
    res = ev.eval(dump=False)
    data = ev._to_dev(ev.test_dataset[0])
    cw = data['cw_1'][0]
    rays = data['rays_1'][0].reshape(-1, 6)
    with torch.no_grad():
        sharded = ev.render_image(data['particles_pos'], rays.shape[0], ev.renderer.set_ro(cw), rays, None, cw, iseval=True)
        ev.world, ev.rank = 1, 0              # the same call unsharded, on this rank alone
        single = ev.render_image(data['particles_pos'], rays.shape[0], ev.renderer.set_ro(cw), rays, None, cw, iseval=True)
    ok = all(torch.equal(sharded[k], single[k]) for k in single) and set(sharded) == set(single)
    q.put((rank, bool(ok), [round(x, 6) for x in res["psnr"]]))
    dist.barrier()
    dist.destroy_process_group()


def test_tile_sharded_eval_two_ranks_one_gpu(workdir):
    """BASELINE config 4's control flow (eval_e2e.py with ray chunks interleaved over ranks + gather) with the REAL
    kernels: 2 ranks sharing this GPU over gloo render the same frames; the sharded image must equal the unsharded one
    bit for bit on every rank, and both ranks must report identical PSNRs (they gathered the same image)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_eval_worker, args=(r, 2, port, str(workdir), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [r[:2] for r in res] == [(0, True), (1, True)]
    assert res[0][2] == res[1][2] and len(res[0][2]) > 0


def _dp_train_worker(rank, world, port, root, backend, q):
    """One rank of a data-parallel train_renderer.py run: identical replicas, rank-decorrelated pixel draws, gradient all-reduce.
    backend "gloo": the ranks share device 0 (what a one-GPU box can run); "nccl": one device per rank, the all-reduce over RCCL."""
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank if backend == "nccl" else 0), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    import configs
    from neurofluid_amd import dist as nfdist
    nfdist.init_from_env(backend=backend)
    from neurofluid_amd.trainers import RendererTrainer
    cfg = configs.warmup_training_config(["--expdir", os.path.join(root, "exps"), "--expname", f"dp_{backend}_r{rank}", "--dataset", "watercube"])
    ds = configs.dataset_config()["watercube"]
    for split in ("train", "test"):
        ds[split].path = os.path.join(root, "data", "watercube")
        ds[split].start_index, ds[split].end_index = 0, 4
    cfg.update(ds)
    for node in (cfg.TRAIN, cfg.TEST):
        node.imgW = node.imgH = 32
    cfg.RENDERER.ray.ray_chunk = 128
    cfg.TRAIN.save_interval = 10 ** 9
    cfg.TRAIN.N_iters = 3
    tr = RendererTrainer(cfg)
    assert (tr.rank, tr.world) == (rank, world) and dist.get_backend() == backend
    before = [p.detach().clone() for p in tr.renderer.parameters()]
    same_start = nfdist.replicas_in_sync(tr.renderer.parameters(), world, resync_from=None)       # same seed on every rank
    loss = tr.train()
    moved = any(not torch.equal(a, b.detach()) for a, b in zip(before, tr.renderer.parameters()))
    # the ranks drew DIFFERENT pixels (seed + rank), so their losses differ, yet after the all-reduced steps the weights are the same bits
    losses = [None] * world
    dist.all_gather_object(losses, float(loss))
    same_end = nfdist.replicas_in_sync(tr.renderer.parameters(), world, resync_from=None)
    q.put((rank, bool(same_start and same_end and moved and len(set(losses)) == world), losses))
    dist.barrier()
    dist.destroy_process_group()


def _run_dp(workdir, backend):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_train_worker, args=(r, 2, port, str(workdir), backend, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [r[:2] for r in res] == [(0, True), (1, True)], res


def test_data_parallel_train_renderer_two_ranks_one_gpu(workdir):
    """BASELINE config 2's N > 1 control flow (trainer_renderer.py:94-143 under data parallelism) with the REAL kernels: two ranks
    share this GPU over gloo, draw different pixels, all-reduce the 5.35 MB of renderer gradients, step Adam — three steps later
    both replicas hold the same bits (dist.replicas_in_sync) although their losses differed."""
    _run_dp(workdir, "gloo")


def test_rccl_data_parallel_train_renderer_two_devices(workdir):
    """The same over real RCCL / xGMI (one device per rank): runs wherever two devices are visible, skips on the one-GPU boxes."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two MI355X devices (RCCL refuses two ranks on one device)")
    _run_dp(workdir, "nccl")


def test_hip_adam_matches_torch_adam():
    """make_adam's optimiser (HipAdam: one nf_adam_step launch per step) against torch.optim.Adam's default implementation on the
    same parameters and gradients: two groups with their own learning rates (trainer_e2e.py:83-139), tensors from 1 element to
    several chunks, weight decay, 6 steps — parameters and both moments within 2 ulp-level bounds (the two run the same operations;
    torch's division / sqrt and the kernel's may round the last bit differently), `step` counters equal; then the state dicts
    interchange: HipAdam continues from a torch.optim.Adam checkpoint and vice versa, and portable_optimizer_state() of both agree."""
    from neurofluid_amd.train_step import make_adam, HipAdam, portable_optimizer_state, load_optimizer_state
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(5)
    shapes = [(1,), (3, 128), (256, 198), (256, 454), (4, 4, 4, 96, 64), (64,), (5000,)]

    def params():
        g = torch.Generator().manual_seed(11)
        return [torch.nn.Parameter((torch.randn(*s, generator=g) * 0.1).to(dev)) for s in shapes]
    pa, pb = params(), params()
    groups = lambda ps: [{"params": ps[:4], "lr": 3e-4}, {"params": ps[4:], "lr": 1e-3, "weight_decay": 1e-2}]      # noqa: E731
    oa = make_adam(groups(pa), betas=(0.9, 0.999), eps=1e-8)
    ob = torch.optim.Adam(groups(pb), betas=(0.9, 0.999), eps=1e-8, foreach=False, fused=False)
    assert isinstance(oa, HipAdam)

    def run(steps, oa, ob, pa, pb):
        for _ in range(steps):
            for x, y in zip(pa, pb):
                gr = (torch.randn(x.shape, generator=gen) * 0.05).to(dev)
                x.grad, y.grad = gr.clone(), gr.clone()
            oa.step(); ob.step()
        for k, (x, y) in enumerate(zip(pa, pb)):
            torch.testing.assert_close(x.detach(), y.detach(), rtol=2e-6, atol=2e-8, msg=f"param {k}")
            sa, sb = oa.state[x], ob.state[y]
            # (one rounding of a value of ~1e-2 is 2e-9: elements that nearly cancel differ by that much in absolute terms)
            torch.testing.assert_close(sa["exp_avg"], sb["exp_avg"], rtol=2e-6, atol=1e-8)
            torch.testing.assert_close(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=2e-6, atol=1e-11)
            assert float(sa["step"]) == float(sb["step"])
    run(6, oa, ob, pa, pb)
    # interchange: each continues from the OTHER's checkpoint
    sda, sdb = portable_optimizer_state(oa), portable_optimizer_state(ob)
    assert sda["param_groups"] == sdb["param_groups"] or [{k: v for k, v in g.items() if k not in ("fused", "foreach")} for g in sda["param_groups"]] == \
        [{k: v for k, v in g.items() if k not in ("fused", "foreach")} for g in sdb["param_groups"]]
    oa2 = make_adam(groups(pa), betas=(0.9, 0.999), eps=1e-8)
    ob2 = torch.optim.Adam(groups(pb), betas=(0.9, 0.999), eps=1e-8, foreach=False, fused=False)
    load_optimizer_state(oa2, sdb)
    load_optimizer_state(ob2, sda)
    run(3, oa2, ob2, pa, pb)
    assert float(oa2.state[pa[0]]["step"]) == 9.0
    # the RAW state_dict() (no portable_optimizer_state): every parameter has its own `step` tensor — HipAdam shares one object per
    # group internally, and an aliased counter in a checkpoint makes a plain Adam advance it once per parameter — so a
    # torch.optim.Adam that loads the raw dict through a torch.save round trip takes ONE step per step
    import io
    raw = oa2.state_dict()
    assert len({id(st["step"]) for st in raw["state"].values()}) == len(raw["state"])
    buf = io.BytesIO(); torch.save(raw, buf); buf.seek(0)
    ob3 = torch.optim.Adam(groups(pb), betas=(0.9, 0.999), eps=1e-8, foreach=False, fused=False)
    ob3.load_state_dict(torch.load(buf, weights_only=False))
    for y in pb:
        y.grad = torch.zeros_like(y)
    ob3.step()
    assert all(float(ob3.state[y]["step"]) == 10.0 for y in pb)
    # a parameter without a gradient is skipped, like torch does
    for x in pa:
        x.grad = None
    before = [x.detach().clone() for x in pa]
    oa2.step()
    assert all(torch.equal(a, b.detach()) for a, b in zip(before, pa))


@pytest.mark.gpu
def test_fused_e2e_loss_matches_torch_ops():
    """train_step.e2e_loss (nf_e2e_loss: the end-to-end step's loss and its gradients in one launch) against the torch expression
    it replaces — the reference's own: sum over views of MSELoss(rgb0_v) + MSELoss(rgb1_v) + w * L1Loss(pos, clip(pos))
    (trainer/trainer_e2e.py:264-280, trainer/basetrainer.py:108-116): loss to 1e-6 relative, every gradient to 1e-6 of its scale
    (one-pass fp32 sums in a different order), an upstream gradient other than 1 included; without the fine pass and without the
    boundary term too."""
    from neurofluid_amd.train_step import e2e_loss
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    V, R = 3, 257
    lo, hi = (-0.975, -0.975, -0.975), (0.975, 0.975, 2.4302)
    for fine, wb in ((True, 0.3), (False, 0.3), (True, 0.0)):
        rgb0 = torch.rand(V * R, 3, generator=g).to(dev).requires_grad_(True)
        rgb1 = torch.rand(V * R, 3, generator=g).to(dev).requires_grad_(True)
        rgbs = torch.rand(V * R, 3, generator=g).to(dev)
        pos = ((torch.rand(4913, 3, generator=g) - 0.5) * 2.2 + torch.tensor([0.0, 0.0, 0.7])).to(dev).requires_grad_(True)
        out = {"rgb0": rgb0, "rgb1": rgb1}
        loss = e2e_loss(out, rgbs, V, fine, pos, (lo, hi), wb)
        (loss * 1.7).backward()
        got = [loss.detach().clone(), rgb0.grad.clone(), rgb1.grad.clone() if fine else None, pos.grad.clone() if wb else None]
        for t in (rgb0, rgb1, pos):
            t.grad = None
        ref = 0.
        for v in range(V):
            sl = slice(v * R, (v + 1) * R)
            ref = ref + torch.nn.functional.mse_loss(rgb0[sl], rgbs[sl])
            if fine:
                ref = ref + torch.nn.functional.mse_loss(rgb1[sl], rgbs[sl])
        if wb:
            clipped = torch.clamp(pos, torch.tensor(lo, device=dev), torch.tensor(hi, device=dev))
            ref = ref + torch.nn.functional.l1_loss(pos, clipped) * wb
        (ref * 1.7).backward()
        assert abs(float(got[0]) - float(ref)) <= 1e-6 * abs(float(ref))
        assert float((got[1] - rgb0.grad).abs().max()) <= 1e-6 * float(rgb0.grad.abs().max())
        if fine:
            assert float((got[2] - rgb1.grad).abs().max()) <= 1e-6 * float(rgb1.grad.abs().max())
        else:
            assert rgb1.grad is None
        if wb:
            assert float((pos.grad != 0).float().mean()) > 0.01            # some particles are outside the box
            assert float((got[3] - pos.grad).abs().max()) <= 1e-6 * float(pos.grad.abs().max())


@pytest.mark.gpu
def test_gather_view_pixels_one_launch_equals_per_view_indexing():
    """train_step.gather_view_pixels on GPU tensors (nf_gather_view_pixels: one upload + one launch for all views' rays, colours and camera
    positions) against indexing every view on its own (trainer/basetrainer.py:186-193): the same rows, bit for bit, for 1 / 4 / 16 views, RGB and
    RGBA colours, at the training size (400 x 400, 1 024 pixels per view); more than 16 views take the per-view path; a selection outside the
    image is refused on the host."""
    from neurofluid_amd.train_step import gather_view_pixels
    dev = torch.device("cuda:0")
    H = W = 400
    g = torch.Generator().manual_seed(3)
    coords = torch.stack(torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij"), -1).reshape(-1, 2)
    for V, C, rc in ((1, 3, 1024), (4, 3, 1024), (16, 4, 333), (17, 3, 64)):
        rays = [torch.randn(H, W, 6, generator=g).to(dev) for _ in range(V)]
        rgbs = [torch.rand(H * W, C, generator=g).to(dev) for _ in range(V)]
        cws = [torch.randn(3, 4, generator=g).to(dev) for _ in range(V)]
        sels = [np.random.RandomState(10 + v).choice(H * W, rc, replace=False) for v in range(V)]
        r, c, ro = gather_view_pixels(rays, rgbs, cws, coords, sels, H, W)
        assert r.shape == (V * rc, 6) and c.shape == (V * rc, C) and ro.shape == (V * rc, 3)
        for v in range(V):
            yx = coords[sels[v]].long().to(dev)
            assert torch.equal(r[v * rc:(v + 1) * rc], rays[v][yx[:, 0], yx[:, 1]])
            assert torch.equal(c[v * rc:(v + 1) * rc], rgbs[v][yx[:, 0] * W + yx[:, 1]])
            assert torch.equal(ro[v * rc:(v + 1) * rc], cws[v][:, 3].expand(rc, 3))
    bad = torch.cat([coords, torch.tensor([[float(H), 0.0]])])
    with pytest.raises(IndexError):
        gather_view_pixels(rays[:1], rgbs[:1], cws[:1], bad, [np.array([H * W])], H, W)
