"""Forward / backward orchestration of the fused renderer (RenderNet.forward,
/root/reference/models/renderer.py:211-270; autograd of the reference = SURVEY §8a row A12)."""
import torch

from . import ops


def _prep(x):
    return x.detach().contiguous().float()


def _run_passes(net, particles, ro, rays, white_bg, fine, save_acts, _retry=False, use_disp=False, noise_std=0.0, _noise=None,
                perturb=0.0):
    dev = rays.device
    z_table, u_table = net._tables(dev, use_disp)
    grid = net.grid_for(particles)
    pts = grid.points
    rays_c = _prep(rays)
    ro_c = _prep(ro)
    use_h = net.mlp_dtype in ("fp16", "split") and not save_acts
    if net.mlp_dtype != "fp32" and save_acts:
        raise RuntimeError(f"RENDERER.mlp_dtype={net.mlp_dtype} is an inference path; train with fp32")
    ws = None if save_acts else net.workspace()
    # learnt row capacities: the inference arena's, or the module's own table for training passes.  The passes run against
    # them without a host round trip; each pass's row count is copied to pinned memory right behind its search kernel
    # (ops.HostFetch) and verified when the whole call is enqueued — the wait is for the LAST SEARCH kernel, not for the MLPs
    # behind it.  Training: the host still has the loss and the backward to enqueue while the forward MLPs run.  Inference:
    # the caller gets its (still computing) result tensors back and enqueues the next frame's transition step and grid
    # build behind this frame's fine MLP — no launch gap between the frames of a rollout.  (Exact sizing — an `.item()` in
    # the middle of each pass — left the GPU idle for ~100 us twice per call; a blocking torch.cat(...).tolist() at the end
    # of the call waits for the whole frame: 0.13 ms of idle GPU per frame, and 9.3 vs 8.3 ms per training step in round 1.)
    caps = ws.row_cap if ws is not None else net.train_row_cap
    # Under HIP-graph capture of a whole training step (train_step.GraphedRendererStep sets net._capture) nothing may wait for the
    # device: the passes run against the learnt capacities, the counts stay on the device (the step's overflow bookkeeping reads them
    # there, nf_note_overflow) and every row-sized launch of forward AND backward is sized by the capacity.
    capture = getattr(net, "_capture", None)
    if capture is not None:
        if not save_acts or _retry:
            raise RuntimeError("graph capture is for the training forward")
        need = [(rays_c.shape[0], net.N_samples)] + ([(rays_c.shape[0], net.N_samples + net.N_importance)] if fine else [])
        if any(k not in caps for k in need):
            raise RuntimeError("graph capture of the training step needs learnt row capacities: run an eager step of this shape first")
        fetch = None
        after = lambda n_rows: capture["counts"].append(n_rows)
    else:
        fetch = ops.HostFetch(dev)
        fetch.add(grid.aabb_words())            # words 0..5: the cloud's bounds (the next grid's bbox hint), then one count per pass
        after = lambda n_rows: fetch.add(n_rows)
    opt = not _retry
    pre = getattr(net, "_prepacked", None) if capture is not None else None     # captured steps pack every blob up front, beside the front-end kernels
    pre0 = pre.get(id(net.nerf_coarse)) if pre else None
    pre1 = pre.get(id(net.nerf_fine)) if pre else None
    if save_acts:
        pk0, ws0, ph0 = (pre0["pk"] if pre0 else net.packed_weights(net.nerf_coarse)), None, None
    else:
        pk0, ws0, ph0 = net.packed_for_inference(net.nerf_coarse, use_h)
    # Random draws, all up front and in the reference's order, so that a seeded generator gives the reference's numbers: the
    # coarse jitter (perturb > 0: torch.rand(R, S0), utils/ray_utils.py:252), the coarse pass's sigma noise (noise_std > 0:
    # torch.randn(R, S0), models/renderer.py:193-196), the inverse-CDF draws (perturb > 0: torch.rand(R, N_imp),
    # utils/ray_utils.py:190), the fine pass's sigma noise.  A redo of the call (row capacities) reuses the first attempt's.
    R_ = rays_c.shape[0]
    if (noise_std or perturb > 0) and _noise is None:
        _noise = [net.draw_perturb((R_, net.N_samples), dev) if perturb > 0 else None,
                  net.draw_noise((R_, net.N_samples), dev) * noise_std if noise_std else None,
                  net.draw_perturb((R_, net.N_importance), dev) if (perturb > 0 and fine) else None,
                  net.draw_noise((R_, net.N_samples + net.N_importance), dev) * noise_std if (noise_std and fine) else None]
    pr0, nz0, pu1, nz1 = (_noise if _noise is not None else (None, None, None, None))
    z0 = ops.coarse_perturb(z_table, pr0, perturb) if pr0 is not None else None       # per-ray coarse depths
    p0 = ops.render_pass(grid, pts, rays_c, z0, None if z0 is not None else z_table, net.N_samples, net.raduis, net.num_neighbor, net.enc_flags,
                         net.use_mask, ro_c, pk0, net.in_channels_xyz, net.in_channels_dir, white_bg, save_acts,
                         packed_h=ph0, ws=ws, need_weights=fine, optimistic=opt, caps=caps, wstream=ws0, after_search=after,
                         noise=nz0, packed_n=pre0["pk_n"] if pre0 else None, pre_mlp=pre["join"] if pre else None)
    p0.packed = pk0
    p0.z = z0
    if capture is not None and capture.get("after_coarse") is not None:
        # the captured training step starts the coarse pass's backward HERE, on a side stream (its loss term needs nothing from the
        # fine pass): it runs under the fine pass's sampling / search / feature launches and beside its forward MLP
        p0.n_active, p0.graph_mode = p0.cap, True
        capture["after_coarse"](p0, rays_c, ro_c, z0, None if z0 is not None else z_table)
    p1 = None
    if fine:
        if pu1 is not None:
            z1 = ops.importance_sample_rays(z0, p0.weights, pu1, net.N_importance)
        else:
            z1 = ops.importance_sample(z_table, p0.weights, u_table, net.N_importance, net.zero_row(dev, use_disp))
        if save_acts:
            pk1, ws1, ph1 = (pre1["pk"] if pre1 else net.packed_weights(net.nerf_fine)), None, None
        else:
            pk1, ws1, ph1 = net.packed_for_inference(net.nerf_fine, use_h)
        p1 = ops.render_pass(grid, pts, rays_c, z1, None, net.N_samples + net.N_importance, net.raduis, net.num_neighbor,
                             net.enc_flags, net.use_mask, ro_c, pk1, net.in_channels_xyz, net.in_channels_dir, white_bg,
                             save_acts, packed_h=ph1, ws=ws, need_weights=False, optimistic=opt, caps=caps, wstream=ws1,
                             after_search=after, noise=nz1, packed_n=pre1["pk_n"] if pre1 else None)
        p1.z = z1
        p1.packed = pk1
    # Inference passes ran against learnt row capacities without a host round trip: verify ONCE, here, with the whole
    # call enqueued (a real rollout reads the image back anyway).  On overflow the capacities grow and the call is redone
    # with exact sizing; capacities also grow ahead of need when a count comes within 10 % of them.
    if capture is not None:
        for p in (p0, p1):
            if p is not None:
                p.n_active, p.graph_mode = p.cap, True          # launches sized by the capacity, counts read on the device
        capture["caps"] = [p.cap for p in (p0, p1) if p is not None]
        return p0, p1, rays_c, ro_c, grid
    cap_runs = [(p, p.cap) for p in (p0, p1) if p is not None and p.cap is not None]
    got = fetch.get()                       # waits for the LAST search kernel only; the MLPs behind it stay queued
    net.note_point_bounds(ops.decode_aabb(got[:6]))          # the next frame's grid bbox: no reduction + sync
    counts = [got[6 + k] for k, p in enumerate((p0, p1)) if p is not None and p.cap is not None]
    if cap_runs:
        overflow = False
        for (p, cap), n in zip(cap_runs, counts):
            key = (p.R, p.S)
            if n > cap * 0.9:
                caps[key] = ops._round_rows(n + n // 4 + 4096)
            overflow |= n > cap
            p.n_active = n
        if ops.PROFILE is not None:      # capacity runs booked their device-side count: the fetched value, no extra sync
            known = {id(p.n_rows): n for (p, _), n in zip(cap_runs, counts)}
            ops.PROFILE["rows"] = [(known[id(r)] if id(r) in known else int(r.item())) if torch.is_tensor(r) else r
                                   for r in ops.PROFILE["rows"]]
        if overflow:
            net.capacity_redos = getattr(net, "capacity_redos", 0) + 1        # (diagnostic: bench.py reports it for rollouts)
            return _run_passes(net, particles, ro, rays, white_bg, fine, save_acts, _retry=True, use_disp=use_disp, noise_std=noise_std,
                               _noise=_noise, perturb=perturb)
    return p0, p1, rays_c, ro_c, grid


class LazyResults(dict):
    """The renderer's result dict.  ``num_nn_*`` has the reference's dtype (int64: ``nn_mask.sum(-1)``,
    models/renderer.py:138) but the kernels count in int32, and the reference's callers read it only for TensorBoard
    histograms (trainer/trainer_renderer.py:138-141): the widening copy (0.14 ms per 400x400 frame, 2.4 % of an fp16 frame) is
    made when a key is first READ, from the int32 tensor the entry keeps alive.  Every way of reading a value goes through
    ``_force``; ``raw_int32`` hands the unconverted tensor to a caller that only forwards it (render_loop)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self._pending = {}

    def set_lazy(self, key, int32_tensor, shape):
        self._pending[key] = (int32_tensor, tuple(shape))
        dict.__setitem__(self, key, None)

    def raw_int32(self, key):
        """(int32 tensor, shape) of a count that has not been widened yet, else None."""
        return self._pending.get(key)

    def _force(self, key=None):
        for k in ([key] if key is not None else list(self._pending)):
            ent = self._pending.pop(k, None)
            if ent is not None:
                dict.__setitem__(self, k, ent[0].view(ent[1]).to(torch.int64))

    def __getitem__(self, key):
        self._force(key)
        return dict.__getitem__(self, key)

    def __setitem__(self, key, value):
        self._pending.pop(key, None)
        dict.__setitem__(self, key, value)

    def get(self, key, default=None):
        return self[key] if key in self else default

    def pop(self, key, *default):
        self._force(key)
        return dict.pop(self, key, *default)

    def __iter__(self):                     # (a Python-level __iter__ also keeps dict(res) / {**res} off the C fast path
        return dict.__iter__(self)          # that would copy the placeholder of a pending key)

    def discard(self, key):
        self._pending.pop(key, None)
        dict.pop(self, key, None)

    def items(self):
        self._force()
        return dict.items(self)

    def values(self):
        self._force()
        return dict.values(self)

    def copy(self):
        self._force()
        return dict(self)

    def __eq__(self, other):
        self._force()
        return dict.__eq__(self, other)

    __hash__ = None

    def __repr__(self):
        self._force()
        return dict.__repr__(self)


def _results(p0, p1):
    R = p0.R
    out = LazyResults({"rgb0": p0.rgb, "depth0": p0.depth, "opacity0": p0.opacity})
    out.set_lazy("num_nn_0", p0.num_nn, (R, p0.S, 1))
    out["mask_0"] = p0.mask_sum.view(R, 1)
    if p1 is not None:
        out.update({"rgb1": p1.rgb, "depth1": p1.depth, "opacity1": p1.opacity})
        out.set_lazy("num_nn_1", p1.num_nn, (R, p1.S, 1))
        out["mask_1"] = p1.mask_sum.view(R, 1)
    return out


def render_forward(net, particles, ro, rays, white_bg=True, fine=True, use_disp=False, noise_std=0.0, perturb=0.0):
    needs_grad = torch.is_grad_enabled() and (particles.requires_grad or any(p.requires_grad for p in net.parameters()))
    if needs_grad:
        from .autograd_bwd import render_with_grad
        return render_with_grad(net, particles, ro, rays, white_bg, fine, use_disp, noise_std, perturb)
    with torch.no_grad():
        p0, p1, _, _, _ = _run_passes(net, particles, ro, rays, white_bg, fine, save_acts=False, use_disp=use_disp,
                                      noise_std=noise_std, perturb=perturb)
        return _results(p0, p1)
