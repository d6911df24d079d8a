#!/usr/bin/env python3
"""bench.py — NeuroFluid hot path on MI355X: rays/sec (+ particle-steps/sec), synthetic watercube 400^2.

  python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run, one rank per GPU)

One "step" (weak scaling, per-GPU work fixed) = the coupled per-frame body of the reference's e2e loop
(eval_e2e.py:58-134): one ParticleNet transition step on the 4 913-particle cloud (replicated on every rank),
then the full coarse+fine render of N 400x400 views (N = number of GPUs), 1024-ray-granular chunks interleaved
over the ranks, RGB tiles all-gathered over RCCL.  With --workload train the step is instead one
train_renderer.py optimiser step (4 views x 1024 rays per rank, forward + backward + Adam, gradients
all-reduced) — BASELINE.json configs[1].
Prints ONE JSON line on rank 0 (contract in the task statement) incl. `roofline` (dominant kernel = the fp32-MFMA
NeRF MLP, timed with HIP events on its stream) and `cpu_baseline` (the oracle = CPU port, timed on the host cores).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

MLP_FLOP_PER_ROW = 1331968          # 665 984 MAC per sample
PARTICLE_STEP_FLOP = 1385088
F32_MATRIX_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz


def renderer_cfg():
    return dict(use_mask=True, ray=dict(ray_chunk=1024, N_importance=128, N_samples=64),
                NN_search=dict(fix_radius=True, particle_radius=0.025, search_raduis_scale=9.0, N_neighbor=20),
                encoding=dict(density=True, var=True, smoothed_pos=True, smoothed_dir=True, exclude_ray=True,
                              same_smooth_factor=False))


def build_scene(dev):
    from neurofluid_amd.synthetic import watercube_scene      # scene + closed-form weights (host-side, not timed)
    return watercube_scene(400, 400)


def cpu_baseline(scene):
    """The oracle (CPU port of the reference path) on a bounded, representative sample (about 10-30 s of CPU work):
    every 40th ray of the 400^2 image (4000 rays, same hit ratio as the full frame) + 5 transition steps.
    torch intra-op threads are capped at 16: the oracle's ops are small and lose time beyond that."""
    from oracle import render_oracle as ro, trans_oracle as to
    from neurofluid_amd import effective_cpus
    cores = min(effective_cpus(), 16)
    torch.set_num_threads(cores)
    rays = scene["rays"][::40].contiguous()
    t0 = time.time()
    ro.render_forward(scene["nerf_state"], scene["P"], scene["c2w"][:, 3], rays, 9.0, 13.0)
    dt = time.time() - t0
    t1 = time.time()
    p, v = scene["P"], torch.zeros_like(scene["P"])
    nsteps = 5
    for _ in range(nsteps):
        p, v, _ = to.particle_net_forward(scene["trans_state"], p, v, scene["box"], scene["bn"])
    dts = time.time() - t1
    return {"value": rays.shape[0] / dt, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": f"oracle.render_forward on every 40th ray of the 400x400 frame ({rays.shape[0]} rays, {dt:.1f} s); "
                      f"oracle.particle_net_forward x{nsteps} on 4913 particles ({dts:.1f} s); "
                      f"{cores} torch threads ({os.cpu_count()} host cores visible, CPU budget {effective_cpus()})",
            "particle_steps_per_sec": scene["P"].shape[0] * nsteps / dts}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="render", choices=["render", "train"])
    ap.add_argument("--chunk", type=int, default=0,
                    help="device ray chunk; 0 = one whole 400x400 view per fused call (results are chunk-independent)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from neurofluid_amd import dist as nfdist, ops
    from neurofluid_amd.renderer import RenderNet
    from neurofluid_amd.transmodel import ParticleNet
    from neurofluid_amd.render_loop import render_image
    import torch.distributed as dist

    # dev switch: NF_BENCH_SINGLE_DEVICE=1 runs an N-rank job on ONE GPU over gloo, to exercise the multi-rank control
    # flow (sharding, collectives, timing protocol) where only one device exists; it is not a performance mode
    single_dev = os.environ.get("NF_BENCH_SINGLE_DEVICE") == "1"
    rank, world, local = nfdist.init_from_env("gloo" if single_dev else None)
    if single_dev:
        local = 0
    if args.chunk <= 0:
        args.chunk = 400 * 400      # weak scaling: chunk k = view k -> rank k mod N, identical load on every rank
    assert world == args.gpus or (args.gpus == 1 and world == 1), f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    scene = build_scene(dev)
    net = RenderNet(renderer_cfg(), 9.0, 13.0)
    net.load_state_dict(scene["nerf_state"], strict=True)
    net = net.to(dev)
    pn = ParticleNet(gravity=(0, 0, -9.81))
    pn.load_state_dict(scene["trans_state"], strict=True)
    pn = pn.to(dev)
    P0 = scene["P"].to(dev)
    box, bn = scene["box"].to(dev), scene["bn"].to(dev)
    roc = scene["c2w"][:, 3].to(dev)
    n_views = world
    rays = scene["rays"].to(dev).repeat(n_views, 1).contiguous()      # N views of the synthetic camera (weak scaling)
    n_rays = rays.shape[0]

    ops.PROFILE = None
    state = {"pos": P0.clone(), "vel": torch.zeros_like(P0)}

    def step_render():
        with torch.no_grad():
            state["pos"], state["vel"], _ = pn(state["pos"], state["vel"], box, bn)
            # every step renders the same (initial) cloud so that step time is stationary; the transition
            # step above is real work on the evolving state
            out = render_image(net, P0, n_rays, roc, rays, None, None, iseval=True, ray_chunk=args.chunk,
                               rank=rank, world=world, gather=False)
        return out

    if args.workload == "train":
        from neurofluid_amd.train_step import make_train_step
        step_fn = make_train_step(net, scene, dev, rank, world)
    else:
        step_fn = step_render

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ops.PROFILE = {"mlp": [], "rows": []}     # warm the HIP-event path too (its first use loads runtime components)
    for _ in range(args.warmup):
        step_fn()
    sync()
    if ops.PROFILE["mlp"]:
        ops.PROFILE["mlp"][0][0].elapsed_time(ops.PROFILE["mlp"][0][1])
    ops.PROFILE = {"mlp": [], "rows": []}
    t0 = time.perf_counter()
    host_marks = []
    for _ in range(args.steps):
        out = step_fn()
        host_marks.append(time.perf_counter() - t0)
        if os.environ.get('NF_BENCH_DEBUG'):
            host_marks.append(-torch.cuda.memory_reserved() / 1e9)
    sync()
    dt = time.perf_counter() - t0
    prof = ops.PROFILE
    ops.PROFILE = None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    rays_per_step = n_rays if args.workload == "render" else 4096 * world
    value = rays_per_step * args.steps / dt

    # ---- roofline of the dominant kernel (this rank's launches)
    mlp_ms = sum(a.elapsed_time(b) for a, b in prof["mlp"])
    rows = sum(prof["rows"])
    mult = 1.0 if args.workload == "render" else 1.0
    n_launch = max(len(prof["mlp"]), 1)
    achieved = rows * MLP_FLOP_PER_ROW * mult / (mlp_ms * 1e-3) / 1e12 if mlp_ms > 0 else 0.0
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "round1_mlp_pmc.json")
    if args.workload == "render" and os.path.exists(pmc):
        # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command
        # (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, tools/summarize_profiles.py)
        traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
    roofline = {"bound": "mfma", "kernel": "k_mlp_fwd_l (fp32 v_mfma_f32_32x32x2_f32, weights through an LDS ring)", "achieved": achieved,
                "peak": F32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / F32_MATRIX_PEAK_TFLOPS,
                "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC)", "launches": len(prof["mlp"]), "avg_launch_ms": mlp_ms / n_launch,
                "executed_rows_per_step": rows / args.steps,
                "flop_per_row": MLP_FLOP_PER_ROW, "mlp_ms_per_step": mlp_ms / args.steps}

    # ---- transition model alone (particle-steps/sec), rank 0 state
    tp, tv = P0.clone(), torch.zeros_like(P0)
    for _ in range(3):
        with torch.no_grad():
            tp, tv, _ = pn(tp, tv, box, bn)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    nts = 20
    for _ in range(nts):
        with torch.no_grad():
            tp, tv, _ = pn(tp, tv, box, bn)
    torch.cuda.synchronize()
    pstep = P0.shape[0] * nts / (time.perf_counter() - t1)

    # ---- extra (NOT the headline, which stays fp32): the same render step with the fp16-MFMA MLP (BASELINE config 5)
    fp16_extra = None
    if args.workload == "render":
        cfg16 = renderer_cfg(); cfg16["mlp_dtype"] = "fp16"
        net16 = RenderNet(cfg16, 9.0, 13.0)
        net16.load_state_dict(scene["nerf_state"], strict=True)
        net16 = net16.to(dev)

        def step16():
            with torch.no_grad():
                return render_image(net16, P0, n_rays, roc, rays, None, None, iseval=True, ray_chunk=args.chunk,
                                    rank=rank, world=world, gather=False)
        out16 = step16()
        sync()
        t2 = time.perf_counter()
        for _ in range(3):
            out16 = step16()
        sync()
        dt16 = (time.perf_counter() - t2) / 3
        mine = nfdist.my_chunks((n_rays + args.chunk - 1) // args.chunk, rank, world)
        a = out["pred_rgbs_1"] if world == 1 else None
        psnr16 = None
        if world == 1:
            mse = torch.mean((out16["pred_rgbs_1"] - out["pred_rgbs_1"]) ** 2).item()
            psnr16 = (-10.0 * math.log10(mse)) if mse > 0 else float("inf")
        fp16_extra = {"rays_per_sec": n_rays / dt16, "ms_per_step": dt16 * 1e3, "dtype": "f16 MFMA, f32 accumulate",
                      "psnr_vs_f32_path_db": psnr16, "note": "render only (no transition step); not the headline value"}

    # ---- extra: BASELINE configs[1] (train_renderer.py step: 4 views x 1024 rays, fwd + bwd + Adam) on this rank
    train_extra = None
    if args.workload == "render":
        from neurofluid_amd.train_step import make_train_step
        net_t = RenderNet(renderer_cfg(), 9.0, 13.0)
        net_t.load_state_dict(scene["nerf_state"], strict=True)
        net_t = net_t.to(dev)
        tstep = make_train_step(net_t, scene, dev, rank, world)
        for _ in range(3):
            tstep()
        sync()
        t3 = time.perf_counter()
        for _ in range(10):
            tstep()
        sync()
        dtt = (time.perf_counter() - t3) / 10
        train_extra = {"workload": "train_renderer.py step: 4 views x 1024 rays per rank, forward + backward + Adam (+ grad all-reduce)",
                       "ms_per_step": dtt * 1e3, "rays_per_sec": 4096 * world / dtt}

    if rank == 0:
        res = {"metric": ("rays/sec (renderer coarse+fine forward) coupled with one transition step per frame, watercube 400^2"
                          if args.workload == "render" else
                          "rays/sec of the train_renderer.py optimiser step (forward + backward + Adam), watercube 400^2"),
               "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": ("eval_e2e per-frame body: ParticleNet step (4913 particles, replicated) + full 400x400 "
                                       "coarse+fine render of %d view(s), 160000 rays each, chunks of %d rays interleaved over "
                                       "%d rank(s)" % (n_views, args.chunk, world)) if args.workload == "render" else
                                      "train_renderer.py step: 4 views x 1024 rays per rank, fwd+bwd+Adam",
                          "particles": int(P0.shape[0]), "image": "400x400", "N_samples": 64, "N_importance": 128,
                          "K": 20, "use_mask": True, "device_ray_chunk": args.chunk},
               "particle_steps_per_sec": pstep,
               "particle_steps_note": "ParticleNet.forward alone on one GPU; the 4913-particle step does not shard (replicas only: "
                                      "every rank advances the same state), so this figure is per replica, not multiplied by N",
               "roofline": roofline, "fp16_mfma_path": fp16_extra, "train_step": train_extra}
        if os.environ.get("NF_BENCH_DEBUG"):
            res["host_marks_ms"] = [round(m * 1e3, 2) for m in host_marks]
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(scene)
            res["speedup_vs_cpu_port"] = value / res["cpu_baseline"]["value"]
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
