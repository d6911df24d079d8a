"""Host-side logic that needs no GPU: operand layout transforms, pixel sampling, LR schedule, chunk sharding."""
import os
import sys

import numpy as np
import pytest
import torch

from neurofluid_amd import dist as nfdist, ops
from neurofluid_amd.train_step import ExponentialLR, random_sample_coords


def test_tile_layout_roundtrip_and_definition():
    g = torch.Generator().manual_seed(0)
    for n in (1, 31, 32, 33, 100):
        x = torch.randn(n, 252, generator=g)
        X = ops.rows_to_tiles(x, 198, 54)
        assert X.numel() == (n + 31) // 32 * 32 * 256
        assert torch.equal(ops.tiles_to_rows(X, n, 198, 54), x)
    # definition check (include/neurofluid_hip.h): X[tile][q][h*32+j][e] = feature 8q+4h+e of row tile*32+j,
    # dir-like features start at group q = QX
    x = torch.arange(40 * 252, dtype=torch.float32).view(40, 252)
    X = ops.rows_to_tiles(x, 198, 54).view(2, 32, 64, 4)
    assert X[1, 3, 1 * 32 + 5, 2] == x[32 + 5, 8 * 3 + 4 * 1 + 2]
    assert X[0, 25, 0 * 32 + 7, 1] == x[7, 198 + 1]
    assert X[0, 24, 1 * 32 + 7, 2] == 0        # pad: features 198,199 of the pos-like block


def test_random_sample_coords_matches_reference_semantics():
    full = random_sample_coords(400, 400, 501, 500)
    assert full.shape == (160000, 2) and full[401].tolist() == [1.0, 1.0]
    crop = random_sample_coords(400, 400, 500, 500)          # step <= precrop_iters -> central half
    assert crop.shape == (200 * 200, 2)
    assert crop.min().item() == 100 and crop.max().item() == 299


def test_exponential_lr():
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=5e-4)
    sch = ExponentialLR(opt, decay_epochs=10000, gamma=0.1)
    for _ in range(100):
        opt.step(); sch.step()
    assert abs(opt.param_groups[0]["lr"] - 5e-4 * 0.1 ** (100 / 10000)) < 1e-12


def test_chunk_ownership_is_interleaved_and_complete():
    for n_chunks, world in ((157, 8), (625, 8), (5, 8), (16, 2)):
        owned = [nfdist.my_chunks(n_chunks, r, world) for r in range(world)]
        assert sorted(c for o in owned for c in o) == list(range(n_chunks))
        assert max(len(o) for o in owned) == nfdist.share_size(n_chunks, world)
        assert all(o == list(range(r, n_chunks, world)) for r, o in enumerate(owned))


def test_gather_chunks_single_rank_identity():
    x = torch.arange(10 * 3, dtype=torch.float32).view(10, 3)
    pad = torch.zeros(12, 3); pad[:10] = x
    assert torch.equal(nfdist.gather_chunks(pad, 3, 4, 10, 0, 1), x)


def test_pixel_sampler_keeps_the_rng_stream():
    """The background prefetch must hand out exactly the selections sequential rng.choice calls would."""
    from neurofluid_amd.train_step import PixelSampler
    ref_rng = np.random.RandomState(10)
    ref = [[ref_rng.choice(160000, size=[1024], replace=False) for _ in range(4)] for _ in range(3)]
    ps = PixelSampler(np.random.RandomState(10), 4, 1024, lambda s: 160000, first_step=7)
    for i in range(3):
        got = ps.next(7 + i)
        assert all(np.array_equal(a, b) for a, b in zip(got, ref[i]))
    ps.close()


def test_synthetic_scene_matches_the_oracle_definitions():
    """bench.py builds its scene from the product-side module; it must be the scene the oracle/goldens use."""
    from neurofluid_amd import synthetic as sy
    from oracle import render_oracle as ro, trans_oracle as to
    assert torch.equal(sy.watercube_particles(), ro.watercube_particles())
    assert torch.equal(sy.eval_camera(), ro.eval_camera())
    a, b = sy.deterministic_nerf_state(), ro.deterministic_nerf_state()
    assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)
    a, b = sy.deterministic_transition_state(), to.deterministic_transition_state()
    assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)
    assert all(torch.equal(x, y) for x, y in zip(sy.watercube_box(), to.watercube_box()))
    sc = sy.watercube_scene(8, 8)
    d = ro.get_ray_directions(8, 8, sy.camera_focal(8))
    o, dd = ro.get_rays(d, sy.eval_camera())
    assert torch.equal(sc["rays"], torch.cat([o, dd], -1).view(-1, 6))


def test_row_capacity_buckets_and_scratch_arena():
    """Row-sized buffers are allocated at bucketed capacities (<= 12.5 % slack, monotone) and the inference scratch
    arena only ever grows: a drifting active-row count must not reach the device allocator every frame."""
    from neurofluid_amd import ops
    prev = 0
    for n in [0, 1, 31, 8192, 8193, 60000, 100001, 1570000, 2950000, 3000000]:
        c = ops._round_rows(n)
        assert c >= max(n, 1) and c % 8192 == 0 and c >= prev
        if n > 65536:
            assert c <= n * 1.125 + 8192
        prev = c
    assert len({ops._round_rows(n) for n in range(2_900_000, 3_000_000, 1000)}) <= 2
    ws = ops.Workspace()
    dev = torch.device("cpu")
    a = ws.get("x", 1000, torch.float32, dev)
    p0 = ws._buf["x"].data_ptr()
    b = ws.get("x", 1100, torch.float32, dev)          # within the 25 % headroom: same storage
    assert ws._buf["x"].data_ptr() == p0 and b.shape == (1100,) and a.dtype == torch.float32
    ws.get("x", 5000, torch.float32, dev)               # grows
    assert ws._buf["x"].numel() >= 5000 * 4
    big = ws._buf["x"].numel()
    ws.get("x", 10, torch.int32, dev)                   # never shrinks, dtype views share the bytes
    assert ws._buf["x"].numel() == big


def test_host_cpu_budget():
    """The intra-op pool is capped to the CPU budget of the container (cgroup quota / affinity), not to the number
    of cores the node shows."""
    import os
    import neurofluid_amd
    n = neurofluid_amd.effective_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    assert torch.get_num_threads() <= max(16, n)
    os.environ["NF_HOST_THREADS"] = "2"
    try:
        assert neurofluid_amd.limit_host_threads() <= max(2, torch.get_num_threads())
    finally:
        del os.environ["NF_HOST_THREADS"]


def test_pixel_sampler_readahead_is_invisible_and_errors_surface():
    """PixelSampler (train_step.py): same selections as drawing inline from the stream; close() joins the worker and
    rewinds the stream to before the first unconsumed draw; a worker exception is re-raised by next()."""
    import numpy as np
    import pytest
    from neurofluid_amd.train_step import PixelSampler
    rng = np.random.RandomState(3)
    s = PixelSampler(rng, n_views=2, ray_chunk=16, n_pixels_of_step=lambda step: 100, first_step=7)
    got = [s.next(7), s.next(8), s.next(9)]
    s.close()
    assert not s.t.is_alive()
    after = rng.randint(1 << 30)
    ref = np.random.RandomState(3)
    want = [[ref.choice(100, size=[16], replace=False) for _ in range(2)] for _ in range(3)]
    for a, b in zip(got, want):
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert after == ref.randint(1 << 30)          # the read-ahead draws were rewound
    # the module-level stream works the same way (the trainers pass np.random itself)
    np.random.seed(11)
    s = PixelSampler(np.random, 1, 8, lambda step: 50, 0)
    a = s.next(0)
    s.close()
    b = np.random.randint(1 << 30)
    np.random.seed(11)
    assert np.array_equal(a[0], np.random.choice(50, size=[8], replace=False)) and b == np.random.randint(1 << 30)
    # fewer pixels than ray_chunk: ValueError in the worker -> raised in the consumer, no hang
    s = PixelSampler(np.random.RandomState(0), 1, 64, lambda step: 10, 0)
    with pytest.raises(ValueError):
        s.next(0)
    s.close()


def test_dataset_readers_vs_reference_readers(tmp_path):
    """the on-disk formats.  tests/golden/f1_dataset.npz holds what the REFERENCE's own BlenderDataset /
    ParticleDataset (datasets/dataset.py, datasets/dataset_splishsplash_rawdata.py) returned for a data set written by
    write_synthetic_dataset with fixed arguments (tests/golden/gen_golden_dataset.py); the build's readers must return the
    same arrays for the same files: items, frame pairing (i, i+1), RGBA-on-white blending, half-resolution resize, focal,
    sliding windows, z-rotation drawn from the global numpy stream."""
    import numpy as np
    import torch
    from conftest import load_golden
    from neurofluid_amd.datasets import BlenderDataset, ParticleDataset, write_synthetic_dataset
    g = load_golden("f1_dataset")
    root = str(tmp_path / "watercube")
    write_synthetic_dataset(root, n_frames=4, img=8, n_side=3, views=("view_0", "view_1"), splits=("train",), camera_angle_x=0.323,
                            seed=10)
    cfg = {"data_type": "splishsplash"}
    ds = BlenderDataset(root, cfg, imgW=8, imgH=8, start_index=0, end_index=4, imgscale=1.0, viewnames=["view_0", "view_1"], split="train")
    assert len(ds) == int(g["blender_len"])
    for idx in (0, 2):
        item = ds[idx]
        keys = [k.split("__")[1] for k in g if k.startswith(f"blender_{idx}__")]
        assert set(keys) == set(item), (sorted(keys), sorted(item))
        for k in keys:
            got = np.asarray(item[k]) if not torch.is_tensor(item[k]) else item[k].numpy()
            ref = g[f"blender_{idx}__{k}"]
            assert got.shape == ref.shape, k
            if k.startswith("rays"):          # directions: same torch ops as the reference -> same bits expected; allow 1 ulp
                np.testing.assert_allclose(got, ref, rtol=0, atol=1.2e-7, err_msg=k)
            else:
                np.testing.assert_array_equal(got, ref, err_msg=k)
    ds2 = BlenderDataset(root, cfg, imgW=8, imgH=8, start_index=1, end_index=3, imgscale=2.0, viewnames=["view_1"], split="train")
    item = ds2[0]
    np.testing.assert_array_equal(item["rgb"].numpy(), g["blender_half__rgb"])
    np.testing.assert_allclose(item["rays"].numpy(), g["blender_half__rays"], rtol=0, atol=1.2e-7)
    np.testing.assert_allclose(np.asarray(item["focal"]), g["blender_half__focal"], rtol=1e-12)
    pd = ParticleDataset(root, "blender", 0, 4, random_rot=False, window=3)
    assert len(pd) == int(g["particles_len"])
    for k, v in pd[1].items():
        np.testing.assert_array_equal(v.numpy(), g[f"particles_1__{k}"], err_msg=k)
    np.random.seed(123)
    pr = ParticleDataset(root, "blender", 0, 4, random_rot=True, window=2)
    assert len(pr) == int(g["particles_rot_len"])
    for k, v in pr[0].items():
        np.testing.assert_array_equal(v.numpy(), g[f"particles_rot0__{k}"], err_msg=k)


def test_native_pixel_draw_matches_numpy_bit_for_bit():
    """nf_host_choice_mt19937 (csrc/nf_host.hip): np.random.choice(n, size, replace=False) of the legacy RandomState
    (trainer/trainer_renderer.py:119) outside the GIL — same indices, same generator state afterwards (the next values of
    the stream agree, including a cached gaussian), for full-frame / centre-crop / degenerate sizes and across the
    generator's 624-word refills and power-of-two mask boundaries."""
    import ctypes
    import numpy as np
    from neurofluid_amd import _lib
    lib = _lib.load()
    cases = [(0, 160000, 1024), (1, 40000, 1024), (7, 5, 5), (3, 1, 1), (3, 2, 1), (11, 1024, 1024), (5, 100000, 1),
             (9, 2 ** 17, 100), (9, 2 ** 17 + 1, 100), (13, 3, 2), (21, 640000, 4096)]
    for seed, n, size in cases:
        r = np.random.RandomState(seed)
        r.standard_normal(3)                      # leaves a cached gaussian in the state
        st = r.get_state()
        want = r.choice(n, size=[size], replace=False)
        key, pos, out = st[1].copy(), ctypes.c_int(st[2]), np.empty(size, dtype=np.int64)
        assert lib.nf_host_choice_mt19937(key.ctypes.data, ctypes.addressof(pos), n, size, out.ctypes.data) == 0
        assert np.array_equal(out, want), (seed, n, size)
        r2 = np.random.RandomState()
        r2.set_state((st[0], key, pos.value) + tuple(st[3:]))
        assert np.array_equal(r.randint(0, 1 << 30, size=700), r2.randint(0, 1 << 30, size=700))
        assert r.standard_normal() == r2.standard_normal()
    # argument errors are reported, not executed
    key, pos, out = np.zeros(624, dtype=np.uint32), ctypes.c_int(0), np.empty(4, dtype=np.int64)
    assert lib.nf_host_choice_mt19937(key.ctypes.data, ctypes.addressof(pos), 3, 4, out.ctypes.data) != 0
    assert lib.nf_host_choice_mt19937(key.ctypes.data, ctypes.addressof(pos), 0, 0, out.ctypes.data) != 0


def test_pixel_sampler_uses_the_native_draw():
    from neurofluid_amd.train_step import PixelSampler
    import numpy as np
    s = PixelSampler(np.random.RandomState(5), 2, 32, lambda step: 1000, 0)
    assert s._native_state() is not None
    s.close()
    # a stream that is not the legacy MT19937 falls back to its own choice()
    class Other:
        def __init__(self): self.r = np.random.RandomState(1)
        def get_state(self): return ('PCG',) + self.r.get_state()[1:]
        def set_state(self, st): self.r.set_state(('MT19937',) + tuple(st[1:]))
        def choice(self, *a, **k): return self.r.choice(*a, **k)
    o = Other()
    s = PixelSampler(o, 1, 8, lambda step: 50, 0)
    got = s.next(0)
    s.close()
    assert np.array_equal(got[0], np.random.RandomState(1).choice(50, size=[8], replace=False))


# ------------------------------------------------------------------------------------------------
# round 3 (ADVICE): portable optimizer state, per-view pixel gather, sampler ownership of its random stream
# ------------------------------------------------------------------------------------------------
def test_portable_optimizer_state_round_trips_with_plain_adam():
    """A checkpoint written through portable_optimizer_state() is what a default torch.optim.Adam writes: CPU float `step`,
    no implementation switch frozen into the param groups — it loads into a plain Adam (the reference's), and a plain Adam's
    checkpoint loads back through load_optimizer_state() with the same next update."""
    import torch
    from neurofluid_amd.train_step import make_adam, portable_optimizer_state, load_optimizer_state
    torch.manual_seed(0)
    w0 = torch.randn(5, 3)
    g1, g2 = torch.randn(5, 3), torch.randn(5, 3)

    def run(opt, p, grads):
        for g in grads:
            p.grad = g.clone()
            opt.step()

    pa = torch.nn.Parameter(w0.clone())
    a = make_adam([pa], lr=1e-2)                       # CPU parameters: the default implementation
    run(a, pa, [g1])
    sd = portable_optimizer_state(a)
    assert all(g.get("fused") is None and g.get("foreach") is None for g in sd["param_groups"])
    assert all(st["step"].device.type == "cpu" and st["step"].dtype == torch.float32 for st in sd["state"].values())
    pb = torch.nn.Parameter(pa.detach().clone())
    b = torch.optim.Adam([pb], lr=1e-2)                # "the reference's optimizer"
    b.load_state_dict(sd)
    run(a, pa, [g2]); run(b, pb, [g2])
    assert torch.equal(pa, pb)
    pc = torch.nn.Parameter(pb.detach().clone())
    c = make_adam([pc], lr=1e-2)
    import copy
    load_optimizer_state(c, copy.deepcopy(b.state_dict()))      # and back: a plain Adam's checkpoint into ours
    g3 = torch.randn(5, 3)
    run(b, pb, [g3]); run(c, pc, [g3])
    assert torch.equal(pb, pc)


def test_gather_view_pixels_indexes_each_view_separately():
    """Same rows as indexing every view on its own (trainer/basetrainer.py:186-193), view-major, one camera position per ray."""
    import torch
    from neurofluid_amd.train_step import gather_view_pixels
    H, W, V, rc = 6, 5, 3, 7
    g = torch.Generator().manual_seed(1)
    rays = [torch.randn(H, W, 6, generator=g) for _ in range(V)]
    rgbs = [torch.rand(H * W, 3, generator=g) for _ in range(V)]
    cws = [torch.randn(3, 4, generator=g) for _ in range(V)]
    coords = torch.stack(torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij"), -1).reshape(-1, 2)
    sels = [np.random.RandomState(v).choice(H * W, rc, replace=False) for v in range(V)]
    r, c, ro = gather_view_pixels(rays, rgbs, cws, coords, sels, H, W)
    for v in range(V):
        yx = coords[sels[v]].long()
        assert torch.equal(r[v * rc:(v + 1) * rc], rays[v][yx[:, 0], yx[:, 1]])
        assert torch.equal(c[v * rc:(v + 1) * rc], rgbs[v][yx[:, 0] * W + yx[:, 1]])
        assert torch.equal(ro[v * rc:(v + 1) * rc], cws[v][:, 3].expand(rc, 3))
    r1, c1, _ = gather_view_pixels(rays[:1], rgbs[:1], cws[:1], coords, sels[:1], H, W)
    assert torch.equal(r1, r[:rc]) and torch.equal(c1, c[:rc])


def test_pixel_sampler_warns_about_foreign_draws_and_closes_twice():
    """Native mode owns a COPY of the stream: somebody else drawing from `rng` while the sampler is alive is detected at
    close() (warning; the stream ends at the sampler's own position).  close() is idempotent and usable as a context."""
    import warnings
    from neurofluid_amd.train_step import PixelSampler
    rng = np.random.RandomState(3)
    with PixelSampler(rng, 1, 8, lambda step: 50, 0) as s:
        native = s._native_state() is not None
        s.next(0)
        rng.rand(4)                                     # a foreign consumer
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            s.close()
    if native:
        assert any("another consumer" in str(x.message) for x in w)
    s.close()                                           # second close: no-op
    rng2 = np.random.RandomState(3)
    with warnings.catch_warnings(record=True) as w2:
        warnings.simplefilter("always")
        with PixelSampler(rng2, 1, 8, lambda step: 50, 0) as s2:
            s2.next(0)
    assert not any("another consumer" in str(x.message) for x in w2)


def test_lazy_results_widen_counts_on_first_read():
    """The renderer's result dict keeps num_nn_* as the kernels' int32 and widens to the reference's int64
    (models/renderer.py:138: nn_mask.sum(-1)) on the first read, whichever way the value is read."""
    import torch
    from neurofluid_amd.autograd import LazyResults
    def make():
        r = LazyResults({"rgb0": torch.zeros(2, 3)})
        r.set_lazy("num_nn_0", torch.arange(6, dtype=torch.int32), (2, 3, 1))
        return r
    r = make()
    assert list(r.keys()) == ["rgb0", "num_nn_0"] and "num_nn_0" in r and len(r) == 2
    assert r.raw_int32("num_nn_0")[0].dtype == torch.int32
    v = r["num_nn_0"]
    assert v.dtype == torch.int64 and v.shape == (2, 3, 1) and v.flatten().tolist() == list(range(6))
    assert r.raw_int32("num_nn_0") is None and r["num_nn_0"] is v
    for read in (lambda d: dict(d)["num_nn_0"], lambda d: {**d}["num_nn_0"], lambda d: d.get("num_nn_0"),
                 lambda d: dict(d.items())["num_nn_0"], lambda d: list(d.values())[1], lambda d: d.pop("num_nn_0"),
                 lambda d: d.copy()["num_nn_0"]):
        got = read(make())
        assert got is not None and got.dtype == torch.int64 and got.shape == (2, 3, 1)
    r = make(); r.discard("num_nn_0")
    assert list(r.keys()) == ["rgb0"]
    r = make(); r["num_nn_0"] = 5
    assert r["num_nn_0"] == 5


def test_bench_without_a_gpu_fails_with_a_message_not_an_assert():
    """bench.py --gpus N typed without a launcher spawns its own ranks; where no device is visible it must say so."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    for n in ("1", "8"):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", n, "--steps", "1", "--warmup", "0"],
                           capture_output=True, text=True, timeout=300, env=env, cwd=root)
        assert r.returncode != 0
        assert "MI355X" in r.stderr and "Traceback" not in r.stderr, r.stderr[-2000:]


# ------------------------------------------------------------------------------------------------
# the generator of the hand-scheduled MLP kernel (neurofluid_amd/csrc/gen_mlp_a.py): structure checks that need no GPU
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("qx,qd", [(25, 7), (25, 4), (24, 7), (17, 4), (8, 4), (9, 7)])
def test_mlp_asm_generator_structure(qx, qd, tmp_path):
    """gen_mlp_a.py for a feature-row shape: the instruction stream holds exactly the MFMAs of one tile (8 per 8-block K-step, 4 per view-branch
    half-step), a tile is a whole EVEN number of 8-slot chunks (ring phase = chunk parity; narrow rows end in padding slots), the LDS fits, every
    counted wait is within the 6-bit vmcnt, every accumulator range is one of the three sets, and one s_barrier per chunk + the prologue's."""
    import re
    import subprocess
    import sys as _sys
    gen = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "neurofluid_amd", "csrc", "gen_mlp_a.py")
    out = str(tmp_path / "body.inc")
    r = subprocess.run([_sys.executable, gen, out, str(qx), str(qd)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    text = open(out).read()
    lines = [ln[1:-3] for ln in text.splitlines() if ln.startswith('"')]
    slots = 8 * qx + 2 * qd + 1098
    padded = (slots + 15) // 16 * 16
    assert f"#define NF_A_SLOTS_{qx}_{qd} {padded}" in text
    lds = int(re.search(rf"#define NF_A_LDS_BYTES_{qx}_{qd} (\d+)", text).group(1))
    assert lds == 2 * 16384 + 4 * qx * 1024 + 2560 and lds <= 160 * 1024
    mf = [ln for ln in lines if ln.startswith("v_mfma_f32_32x32x2_f32")]
    assert len(mf) == 8 * (8 * qx + 9 + 8 * 128) + 4 * (4 * qd + 128 + 1)
    for ln in mf:          # destinations: accA v[16b:16b+15], accB a[16b:...], view accumulators a[128+16b:...]
        m = re.match(r"v_mfma_f32_32x32x2_f32 ([va])\[(\d+):(\d+)\], v(\d+), ([va])(\d+), (0|[va]\[\d+:\d+\])$", ln)
        assert m, ln
        lo, hi = int(m.group(2)), int(m.group(3))
        assert hi == lo + 15 and lo % 16 == 0 and lo < (128 if m.group(1) == "v" else 192)
        assert 128 <= int(m.group(4)) < 144                     # A operands: the two operand sets
    assert sum(ln == "s_barrier" for ln in lines) == padded // 8 + 1
    for ln in lines:
        m = re.match(r"s_waitcnt vmcnt\((\d+)\)", ln)
        if m:
            assert int(m.group(1)) < 64
    assert not any("None" in ln for ln in lines)
    # every global_load of the ring refill is followed, later in its chunk cycle, by exactly one publish of the same staging quad
    refill = [ln for ln in lines if ln.startswith("global_load_dwordx4") and ", v177, s[42:43]" in ln]
    publish = [ln for ln in lines if ln.startswith("ds_write_b128 v177")]
    assert len(publish) == 4 * (padded // 8) and len(refill) == len(publish) + 8      # + the prologue's chunks 0 and 1


def test_fp16_mlp_asm_generator_structure(tmp_path):
    """gen_mlp_ha.py: the instruction stream holds exactly the MFMAs of one pair of tiles (two per non-bias step of nf_mlp_h2.hip's stream: 1 414 - 78
    = 1 336 steps; the bias K-step of every output block is the C operand of the block's first MFMAs instead), the pair is a whole number of
    48-block rings (1 344 blocks = what nf_nerf_pack_ha writes), the LDS fits, every counted wait is inside its counter, accumulators are the four
    VGPR sets, A operands the four AGPR slots, B operands come from a bank / the stash slots / the direction registers, one rendezvous per
    16-step chunk + the prologue's, four LDS-DMA pieces per chunk + the prologue's two chunks, four bias-table loads per block + the prologue's,
    and every finished block is rounded exactly once (16 v_cvt_pk_f16_f32 per converted block)."""
    import re
    import subprocess
    import sys as _sys
    gen = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "neurofluid_amd", "csrc", "gen_mlp_ha.py")
    out = str(tmp_path / "body.inc")
    r = subprocess.run([_sys.executable, gen, out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    text = open(out).read()
    lines = [ln[1:-3] for ln in text.splitlines() if ln.startswith('"')]
    nblocks = 8 + 8 * 8 + 1 + 4 + 1
    steps = 8 * 14 + 3 * 8 * 17 + 8 * 30 + 4 * 8 * 17 + 17 + 4 * 21 + 9 - nblocks
    assert (nblocks, steps) == (78, 1336)
    assert "#define NF_HA_STEPS 1336" in text and "#define NF_HA_SLOTS 1344" in text
    lds = int(re.search(r"#define NF_HA_LDS_BYTES (\d+)", text).group(1))
    assert lds == 48 * 1024 + 4 * 2 * 13 * 1024 and lds <= 160 * 1024
    mf = [ln for ln in lines if ln.startswith("v_mfma_f32_32x32x16_f16")]
    assert len(mf) == 2 * steps
    first = 0
    for ln in mf:
        m = re.match(r"v_mfma_f32_32x32x16_f16 v\[(\d+):(\d+)\], a\[(\d+):(\d+)\], ([va])\[(\d+):(\d+)\], v\[(\d+):(\d+)\]$", ln)
        assert m, ln
        lo, hi = int(m.group(1)), int(m.group(2))
        assert hi == lo + 15 and lo in (0, 16, 32, 48)
        c = (int(m.group(8)), int(m.group(9)))
        assert c in ((lo, hi), (192, 207))                # accumulate, or start from the block's biases
        first += c == (192, 207)
        a0 = int(m.group(3))
        assert a0 in (224, 228, 232, 236) and int(m.group(4)) == a0 + 3
        b0, b1 = int(m.group(6)), int(m.group(7))
        assert b1 == b0 + 3 and b0 % 4 == 0
        if m.group(5) == "v":
            assert 64 <= b0 < 192                         # bank 0
        else:
            assert b0 < 128 or 160 <= b0 < 224            # bank 1 / direction operands / stash slots
    assert first == 2 * nblocks
    assert sum(ln == "s_barrier" for ln in lines) == 1344 // 16 + 1
    dma = [ln for ln in lines if ln.startswith("global_load_lds_dwordx4")]
    assert len(dma) == 4 * (1344 // 16) + 8
    bias = [ln for ln in lines if re.match(r"global_load_dwordx4 v\[(192|196|200|204):", ln)]
    assert len(bias) == 4 * nblocks + 4
    for ln in lines:
        m = re.match(r"s_waitcnt vmcnt\((\d+)\)", ln)
        if m:
            assert int(m.group(1)) < 64
        m = re.match(r"s_waitcnt lgkmcnt\((\d+)\)", ln)
        if m:
            assert int(m.group(1)) < 16
    assert sum(ln.startswith("v_cvt_pk_f16_f32") for ln in lines) == 16 * (nblocks - 2)     # 76 blocks convert their predecessor (all but layer 0 block 0 and view block 0)
    assert not any("None" in ln for ln in lines)


def test_reduced_precision_mlp_refuses_other_feature_rows_at_construction():
    """RENDERER.mlp_dtype fp16 / split have weight streams for the default 198 + 54 feature row only: a configuration with another encoding is refused
    when the module is built (CPU, no library needed), not inside its first frame; fp32 accepts every encoding (tests/golden/cfg_*.npz)."""
    import pytest
    from neurofluid_amd.renderer import RenderNet

    def renderer_cfg():
        return dict(use_mask=True, ray=dict(ray_chunk=1024, N_importance=128, N_samples=64),
                    NN_search=dict(fix_radius=True, particle_radius=0.025, search_raduis_scale=9.0, N_neighbor=20),
                    encoding=dict(density=True, var=True, smoothed_pos=True, smoothed_dir=True, exclude_ray=True, same_smooth_factor=False))
    for dt in ("fp16", "split"):
        cfg = renderer_cfg(); cfg["mlp_dtype"] = dt
        RenderNet(cfg, 9.0, 13.0)                                      # the default row: fine
        cfg["encoding"] = dict(cfg["encoding"], smoothed_dir=False)
        with pytest.raises(NotImplementedError, match="198 \\+ 54"):
            RenderNet(cfg, 9.0, 13.0)
    cfg = renderer_cfg(); cfg["encoding"] = dict(cfg["encoding"], smoothed_dir=False)
    assert RenderNet(cfg, 9.0, 13.0).in_channels_dir == 27
    with pytest.raises(ValueError):
        RenderNet(dict(renderer_cfg(), mlp_dtype="bf16"), 9.0, 13.0)
