"""RCCL on the box: the N > 1 data path of bench.py / the trainers is `backend="nccl"` (= RCCL); the multi-rank logic is
covered by the gloo tests (tests/test_dist_gloo.py, test_tile_sharded_eval_two_ranks_one_gpu), this one checks that the
collectives the path uses actually run through RCCL on an MI355X of this image (a 1-rank group: a second rank on the same
device is refused by RCCL, and the boxes have one GPU)."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_single_rank_collectives():
    code = textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r)
        import torch, torch.distributed as dist
        from neurofluid_amd import dist as nfdist
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29641")
        torch.cuda.set_device(0)
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
        assert dist.get_backend() == "nccl"
        x = torch.arange(12, dtype=torch.float32, device="cuda").view(4, 3)
        out = torch.empty(4, 3, device="cuda")
        dist.all_gather_into_tensor(out, x.contiguous())          # the RGB-tile gather of render_image
        assert torch.equal(out, x)
        flat = torch.ones(1 << 20, device="cuda")
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)                 # the flat gradient bucket of allreduce_grads
        assert float(flat.sum()) == float(1 << 20)
        t = torch.tensor([1.5], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                    # bench.py's max-over-ranks step time
        dist.barrier()
        torch.cuda.synchronize()
        # the same collectives through the package's helpers at world = 1 (early-outs) and a forced 1-rank gather
        assert torch.equal(nfdist.gather_chunks(x, 1, 4, 4, 0, 1), x)
        dist.destroy_process_group()
        print("rccl ok")
    """ % ROOT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def _bench(args, env_extra, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          timeout=timeout, env=env, cwd=ROOT)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` as a driver types it (no torchrun): bench.py starts the ranks itself and rank 0 prints the
    one JSON line (strong ray-tile scaling, chunk k -> rank k mod 2).  The box has ONE device, so the two ranks share it over
    gloo (NF_BENCH_SINGLE_DEVICE=1): launch path, sharding, collectives and accounting are what is checked, not the rate."""
    import json
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--no-extras", "--no-cpu-baseline"], {"NF_BENCH_SINGLE_DEVICE": "1"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["scaling"] == "strong" and res["steps"] == 1
    assert res["max_over_mean"] is not None and 1.0 <= res["max_over_mean"] < 1.1
    assert len(res["load_balance"]["executed_rows_per_rank_per_step"]) == 2
    assert res["value"] > 0 and "single_device_emulation" in res
    # the per-rank breakdown a sub-linear scaling curve is diagnosed from: every rank reports its phases and its own finish
    br = res["per_rank_breakdown"]
    assert [r["rank"] for r in br["per_rank"]] == [0, 1] and br["finish_skew_ms_per_step"] >= 0
    for r in br["per_rank"]:
        assert r["transition_ms_per_step"] > 0 and r["grid_ms_per_step"] > 0 and r["render_ms_per_step"] > 0
        assert r["gather_ms_per_step"] >= 0 and r["own_wall_ms_per_step"] > 0


def test_bench_more_ranks_than_devices_is_refused_with_a_message():
    import torch
    n = torch.cuda.device_count() + 1
    r = _bench(["--gpus", str(n), "--steps", "1", "--warmup", "1", "--no-extras", "--no-cpu-baseline"], {"NF_BENCH_SINGLE_DEVICE": "0"}, timeout=300)
    assert r.returncode == 2
    assert "device(s) visible" in r.stderr and "AssertionError" not in r.stderr and "Traceback" not in r.stderr


def test_own_chunk_rays_bit_equal_to_the_indexed_full_tensor():
    """a rank generates the rays of ITS chunks (nf_get_rays_chunks) instead of indexing an (H*W, 6) tensor — the same bits,
    for every rank of worlds 1 / 2 / 3 / 8, ragged last chunk included (400 x 400 = 156.25 chunks of 1024)."""
    import torch
    from neurofluid_amd import ray_utils, synthetic
    from neurofluid_amd import dist as nfdist
    dev = torch.device("cuda:0")
    c2w = synthetic.eval_camera().to(dev)
    for H, W, chunk in ((400, 400, 1024), (37, 53, 64)):
        focal = synthetic.camera_focal(W)
        full = ray_utils.get_rays_device(H, W, focal, c2w)
        n_chunks = (H * W + chunk - 1) // chunk
        for world in (1, 2, 3, 8):
            for rank in range(world):
                own = torch.cat([torch.arange(k * chunk, min((k + 1) * chunk, H * W)) for k in nfdist.my_chunks(n_chunks, rank, world)]).to(dev)
                got = ray_utils.get_rays_own_chunks(H, W, focal, c2w, chunk, rank, world, device=dev)
                assert torch.equal(got, full.index_select(0, own)), (H, W, world, rank)


def test_rccl_two_ranks_two_devices():
    """The N > 1 path over real RCCL / xGMI: skipped on the one-GPU boxes of this pool, runs wherever two devices are visible —
    `bench.py --gpus 2` (strong ray-tile scaling, own-chunk ray generation, RGB all-gather through RCCL)."""
    import json
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two MI355X devices (RCCL refuses two ranks on one device)")
    r = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline"], {"NF_BENCH_SINGLE_DEVICE": "0"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert res["n_gpus"] == 2 and "single_device_emulation" not in res and res["value"] > 0
    assert len(res["per_rank_breakdown"]["per_rank"]) == 2
