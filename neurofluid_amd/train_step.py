"""One optimiser step of the warm-up trainer (/root/reference/trainer/trainer_renderer.py:75-143):
for each of the 4 warm-up views sample `ray_chunk` random pixels of frame 0 (centre crop for the first
`precrop_iters` steps, trainer/basetrainer.py:171-193), render them from the GT particles, loss = sum over views
of MSE(rgb0) + MSE(rgb1); zero_grad / backward / Adam / ExponentialLR (utils/lr_schedulers.py:3-12).
Used by bench.py --workload train and by the Trainer in neurofluid_amd/trainers.py."""
import ctypes

import numpy as np
import torch

from . import dist as nfdist


_COORDS = {}


_STAGE = {}


def _upload(t, device):
    """Host -> device copy that neither synchronises the stream nor allocates: a pageable .to(device) waits for everything
    enqueued before it (i.e. for the previous optimiser step), and Tensor.pin_memory() page-locks a fresh buffer per call
    (0.9 ms of host time each in the trace).  A ring of 4 pinned staging buffers per (dtype, size class) is reused; a
    slot is rewritten only after the event recorded behind its last copy has completed."""
    if device.type != "cuda":
        return t.to(device)
    t = t.contiguous()
    n = t.numel()
    cap = 1 << max(10, (n - 1).bit_length())
    ring = _STAGE.get((t.dtype, cap))
    if ring is None:        # page-lock all four slots at once (each hipHostMalloc stalls the device: not one per early step)
        ring = _STAGE[(t.dtype, cap)] = {"i": 0, "slots": [[torch.empty(cap, dtype=t.dtype).pin_memory(), None] for _ in range(4)]}
    slot = ring["slots"][ring["i"] % 4]
    ring["i"] += 1
    if slot[1] is not None:
        slot[1].synchronize()
    buf = slot[0][:n].view(t.shape)
    buf.copy_(t)
    out = buf.to(device, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    slot[1] = ev
    return out


class HipAdam(torch.optim.Adam):
    """torch.optim.Adam whose step() is ONE HIP launch over the whole parameter list (nf_adam_step, csrc/nf_host.hip) instead of
    torch's multi_tensor_apply launches (2-3 x 44 us per training step for 2.0 M parameters).  Same state layout as a default
    (non-fused) torch.optim.Adam — `step` a CPU float tensor, `exp_avg`, `exp_avg_sq` — so state dicts interchange with it and
    with the reference's checkpoints; same arithmetic as its single-tensor path, operation by operation.  amsgrad / maximize /
    capturable / differentiable and non-fp32 or CPU parameters take the parent's step()."""

    def __init__(self, params, **kw):
        kw.pop("fused", None); kw.pop("foreach", None)
        super().__init__(params, foreach=False, fused=False, **kw)
        self._tables = {}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables = {}               # the state tensors were replaced: the pointer tables are rebuilt at the next step

    def _eligible(self):
        for group in self.param_groups:
            if group.get("amsgrad") or group.get("maximize") or group.get("capturable") or group.get("differentiable") or \
                    torch.is_tensor(group["lr"]):
                return False
            for p in group["params"]:
                if p.grad is not None and ((not p.is_cuda) or p.dtype is not torch.float32 or p.grad.is_sparse or not p.is_contiguous()):
                    return False
        return True

    def state_dict(self):
        """The parameters of a group share ONE `step` tensor object here; written out as is, torch.save keeps the aliasing and a
        plain torch.optim.Adam (the reference's trainer) that loads the file advances the shared counter once per PARAMETER.
        Every parameter gets its own copy in the dict, as torch.optim.Adam writes it."""
        sd = super().state_dict()
        for st in sd["state"].values():
            if torch.is_tensor(st.get("step")):
                st["step"] = st["step"].detach().clone()
        return sd

    @torch.no_grad()
    def step(self, closure=None):
        import ctypes
        from . import _lib
        if not self._eligible():
            return super().step(closure)
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            beta1, beta2 = group["betas"]
            lr, eps, wd = float(group["lr"]), float(group["eps"]), float(group["weight_decay"])
            n = len(ps)
            key = tuple(id(p) for p in ps)
            tab = self._tables.get(gi)
            if tab is None or tab["key"] != key or (tab["step"] is not None and self.state[ps[0]].get("step") is not tab["step"]):
                # (re)build the table: state tensors are created here as a default torch.optim.Adam creates them; the parameters of
                # a group that have taken the same number of steps SHARE one `step` tensor (one host increment per step instead
                # of one per tensor; state_dict() writes its value under every parameter, as torch does)
                PA, FA, LA = ctypes.c_void_p * n, ctypes.c_float * n, ctypes.c_int64 * n
                tab = self._tables[gi] = {"key": key, "p": PA(), "g": PA(), "m": PA(), "v": PA(), "sz": LA(), "ss": FA(), "bc": FA(),
                                          "step": None, "refs": []}
                steps = []
                for k, p in enumerate(ps):
                    st = self.state[p]
                    if len(st) == 0:
                        st["step"] = torch.tensor(0.0, dtype=torch.float32)
                        st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                        st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    if st["step"].is_cuda:                   # a state loaded from a fused optimizer's checkpoint
                        st["step"] = st["step"].detach().to("cpu", torch.float32)
                    steps.append(float(st["step"]))
                    tab["p"][k], tab["m"][k], tab["v"][k] = p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                    tab["sz"][k] = p.numel()
                    tab["refs"].append((st["exp_avg"], st["exp_avg_sq"]))
                if len(set(steps)) == 1:
                    shared = torch.tensor(steps[0], dtype=torch.float32)
                    for p in ps:
                        self.state[p]["step"] = shared
                    tab["step"] = shared
            if tab["step"] is not None:
                tab["step"] += 1
                t = float(tab["step"])
                ss, bc = lr / (1.0 - beta1 ** t), (1.0 - beta2 ** t) ** 0.5
                for k in range(n):
                    tab["ss"][k] = ss; tab["bc"][k] = bc
            else:
                for k, p in enumerate(ps):
                    stp = self.state[p]["step"]
                    stp += 1
                    t = float(stp)
                    tab["ss"][k] = lr / (1.0 - beta1 ** t); tab["bc"][k] = (1.0 - beta2 ** t) ** 0.5
            keep = []
            for k, p in enumerate(ps):
                g = p.grad
                if not g.is_contiguous() or g.dtype is not torch.float32:
                    g = g.contiguous().float(); keep.append(g)
                tab["p"][k] = p.data_ptr()                   # (a parameter's storage may have been replaced: load_state_dict, .to())
                tab["g"][k] = g.data_ptr()
            _lib.check(lib.nf_adam_step(n, tab["p"], tab["g"], tab["m"], tab["v"], tab["sz"], tab["ss"], tab["bc"], float(beta1), float(beta2),
                                        eps, wd, _lib.stream()), "nf_adam_step")
        return loss


def make_adam(params, **kw):
    """Adam for the trainers: HipAdam (one HIP launch per step; the state layout of a default torch.optim.Adam) for parameters on
    the GPU, plain torch.optim.Adam otherwise.  Checkpoints go through portable_optimizer_state(), which writes what a plain
    torch.optim.Adam writes (and what the reference's checkpoints hold)."""
    params = list(params)
    groups = params if params and isinstance(params[0], dict) else [{"params": params}]
    groups = [dict(g, params=list(g["params"])) for g in groups]
    on_gpu = all(p.is_cuda for g in groups for p in g["params"]) and any(len(g["params"]) for g in groups)
    return HipAdam(groups, **kw) if on_gpu else torch.optim.Adam(groups, **kw)


def portable_optimizer_state(optimizer):
    """optimizer.state_dict() normalised to what a default (non-fused) torch.optim.Adam saves: `step` as a CPU float tensor,
    no implementation switches (`fused`, `foreach`) frozen into the param groups — loadable on a CPU-only box, by the
    reference's optimizer, and by make_adam() alike (load_state_dict keeps the LOADING optimizer's own implementation
    switches when the saved groups do not name any)."""
    sd = optimizer.state_dict()
    out = {"state": {}, "param_groups": []}
    for k, st in sd["state"].items():
        st = {a: (b.detach().clone() if torch.is_tensor(b) else b) for a, b in st.items()}      # a snapshot, not an alias
        if torch.is_tensor(st.get("step")):
            st["step"] = st["step"].detach().to("cpu", torch.float32)
        out["state"][k] = st
    for g in sd["param_groups"]:
        g = dict(g)
        for key in ("fused", "foreach"):
            g[key] = None
        out["param_groups"].append(g)
    return out


def load_optimizer_state(optimizer, state):
    """load_state_dict that keeps THIS optimizer's implementation switches (a checkpoint written by the fused
    implementation of an older build, or by the reference's plain Adam, must not flip them)."""
    keep = [{k: g.get(k) for k in ("fused", "foreach")} for g in optimizer.param_groups]
    optimizer.load_state_dict(state)
    for g, k in zip(optimizer.param_groups, keep):
        g.update(k)
    if any(g.get("fused") for g in optimizer.param_groups):          # the fused kernel wants its step counters on the device
        for p, st in optimizer.state.items():
            if torch.is_tensor(st.get("step")) and p.is_cuda:
                st["step"] = st["step"].to(p.device, torch.float32)


def random_sample_coords(H, W, global_step, precrop_iters):
    """trainer/basetrainer.py:171-193 (CPU tensors, like the reference's device-less meshgrid).  The grid only
    has two variants (centre crop / full frame), so it is built once instead of once per view per step."""
    key = (H, W, global_step > precrop_iters)
    if key not in _COORDS:
        _COORDS[key] = _build_coords(H, W, global_step, precrop_iters)
    return _COORDS[key]


def _build_coords(H, W, global_step, precrop_iters):
    if global_step > precrop_iters:
        ys, xs = torch.linspace(0, H - 1, H), torch.linspace(0, W - 1, W)
    else:
        dH, dW = int(H // 2 * 0.5), int(W // 2 * 0.5)
        ys = torch.linspace(H // 2 - dH, H // 2 + dH - 1, 2 * dH)
        xs = torch.linspace(W // 2 - dW, W // 2 + dW - 1, 2 * dW)
    coords = torch.stack(torch.meshgrid(ys, xs, indexing="ij"), -1)
    return coords.reshape(-1, 2)


def choice_without_replacement(rng, n, size):
    """rng.choice(n, size=[size], replace=False) of a legacy numpy stream (np.random itself or a RandomState), drawn by
    nf_host_choice_mt19937 when the stream is MT19937 and the library is built (same indices, same state afterwards,
    about 2.5x faster than numpy's shuffle); numpy otherwise."""
    try:
        st = rng.get_state()
        native = st[0] == 'MT19937' and n >= size >= 0 and 1 <= n < 2 ** 31
        if native:
            from . import _lib
            lib = _lib.load()
    except Exception:
        native = False
    if not native:
        return rng.choice(n, size=[size], replace=False)
    import ctypes
    key, pos = np.array(st[1], dtype=np.uint32), ctypes.c_int(int(st[2]))
    out = np.empty(size, dtype=np.int64)
    _lib.check(lib.nf_host_choice_mt19937(key.ctypes.data, ctypes.addressof(pos), n, size, out.ctypes.data),
               "nf_host_choice_mt19937")
    rng.set_state(('MT19937', key, pos.value) + tuple(st[3:]))
    return out


def gather_view_pixels(rays_list, rgb_list, cw_list, coords, sels, H, W):
    """The pixel gathers of all views of one step with ONE upload and (GPU tensors) ONE launch (the reference indexes
    every view separately, trainer/basetrainer.py:186-193: 2 two-index gathers + 1 upload per view = 16 small dispatches
    and 1.4 ms of host time per step).  rays_list[v] (H, W, 6), rgb_list[v] (H*W, C), cw_list[v] (3, 4) on the device,
    coords (n, 2) on the host, sels[v] the selected rows of coords.  Returns rays (V*rc, 6), rgbs (V*rc, C), ro (V*rc, 3),
    view-major."""
    dev = rays_list[0].device
    V, rc = len(rays_list), len(sels[0])
    yx = torch.cat([coords[torch.as_tensor(s)] for s in sels]).long()
    flat_host = yx[:, 0] * W + yx[:, 1]                       # pixel index inside its own view
    if dev.type == "cuda" and 1 <= V <= 16 and all(t.dtype is torch.float32 for t in list(rays_list) + list(rgb_list) + list(cw_list)):
        # ONE upload + ONE launch (nf_gather_view_pixels) for all views and all three tensors
        if rc and (int(flat_host.min()) < 0 or int(flat_host.max()) >= H * W):
            raise IndexError("pixel selection outside the %d x %d image" % (H, W))
        from . import _lib
        lib = _lib.load()
        flat = _upload(flat_host, dev)
        rl = [r.reshape(H * W, -1).contiguous() for r in rays_list]
        gl = [g.reshape(H * W, -1).contiguous() for g in rgb_list]
        cl = [c.contiguous() for c in cw_list]
        C = gl[0].shape[1]
        if any(r.shape[1] != 6 for r in rl) or any(g.shape[1] != C for g in gl) or any(tuple(c.shape) != (3, 4) for c in cl):
            raise ValueError("gather_view_pixels: rays (H, W, 6), colours (H*W, C) and c2w (3, 4) expected for every view")
        rays = torch.empty(V * rc, 6, device=dev)
        rgbs = torch.empty(V * rc, C, device=dev)
        ro = torch.empty(V * rc, 3, device=dev)
        arr = ctypes.c_void_p * V
        _lib.check(lib.nf_gather_view_pixels(V, arr(*[t.data_ptr() for t in rl]), arr(*[t.data_ptr() for t in gl]), arr(*[t.data_ptr() for t in cl]),
                                             rc, C, H * W, flat.data_ptr(), rays.data_ptr(), rgbs.data_ptr(), ro.data_ptr(), _lib.stream()),
                   "nf_gather_view_pixels")
        return rays, rgbs, ro
    flat = _upload(flat_host, dev)          # ONE upload for all views
    # one index_select per view and tensor on that view's own storage (round 2 concatenated the whole images of all views
    # first: 23 MB of copies per step at 400^2, 92 MB at 800^2, to read 4 096 rows)
    rays = torch.cat([rays_list[v].reshape(H * W, -1).index_select(0, flat[v * rc:(v + 1) * rc]) for v in range(V)]) if V > 1 \
        else rays_list[0].reshape(H * W, -1).index_select(0, flat)
    rgbs = torch.cat([rgb_list[v].reshape(H * W, -1).index_select(0, flat[v * rc:(v + 1) * rc]) for v in range(V)]) if V > 1 \
        else rgb_list[0].reshape(H * W, -1).index_select(0, flat)
    ro = torch.stack([cw[:, 3] for cw in cw_list]).repeat_interleave(rc, dim=0)
    return rays, rgbs, ro


def summed_view_mse(out, rgbs, n_views, fine):
    """sum over views of MSE(rgb0_v) [+ MSE(rgb1_v)] (trainer_renderer.py:124-131) for equally sized views: every view's
    mean has the same denominator, so the sum is two global sums over one denominator (2 reductions instead of 2 per view
    plus the slice / add nodes of the autograd graph)."""
    denom = rgbs.numel() // n_views
    tot = torch.nn.functional.mse_loss(out["rgb0"], rgbs, reduction="sum")
    if fine:
        tot = tot + torch.nn.functional.mse_loss(out["rgb1"], rgbs, reduction="sum")
    return tot / denom


class _E2ELossFn(torch.autograd.Function):
    """loss = summed_view_mse(rgb0 [, rgb1]) + w_boundary * L1(pos, clamp(pos, lo, hi)) (trainer/trainer_e2e.py:264-280) as ONE launch
    (nf_e2e_loss, csrc/nf_host.hip) that also leaves the gradients for a unit upstream gradient; backward scales them.  The chain of
    torch ops it replaces — two mse_loss, the clamp / sub / abs / mean of the boundary term, the adds, and their ~20 autograd nodes —
    was ~35 launches of 3-9 us in a 3.2 ms step."""

    @staticmethod
    def forward(ctx, rgb0, rgb1, rgbs, pos, lo, hi, wb, denom):
        from . import _lib
        lib = _lib.load()
        rgb0c, rgbsc = rgb0.contiguous(), rgbs.contiguous()
        rgb1c = None if rgb1 is None else rgb1.contiguous()
        posc = None if pos is None else pos.contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=rgb0.device)
        g0 = torch.empty_like(rgb0c)
        g1 = None if rgb1c is None else torch.empty_like(rgb1c)
        gp = None if posc is None else torch.empty_like(posc)
        f3 = ctypes.c_float * 3
        ptr = lambda t: 0 if t is None else t.data_ptr()        # noqa: E731
        _lib.check(lib.nf_e2e_loss(ptr(rgb0c), ptr(rgb1c), ptr(rgbsc), rgb0c.numel(), int(denom), ptr(posc),
                                   0 if posc is None else posc.shape[0], f3(*[float(v) for v in lo]), f3(*[float(v) for v in hi]), float(wb),
                                   loss.data_ptr(), ptr(g0), ptr(g1), ptr(gp), _lib.stream()), "nf_e2e_loss")
        ctx.g = (g0, g1, gp)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        g0, g1, gp = ctx.g
        if g.is_cuda and g.dtype is torch.float32 and g.numel() == 1:
            # the three gradients times the upstream gradient in ONE launch (out of place: the saved unit-gradient buffers stay intact for a
            # second backward under retain_graph)
            from . import _lib
            gs = g.contiguous()
            o0, o1, op = torch.empty_like(g0), (None if g1 is None else torch.empty_like(g1)), (None if gp is None else torch.empty_like(gp))
            n = lambda t: 0 if t is None else t.numel()          # noqa: E731
            p = lambda t: 0 if t is None else t.data_ptr()       # noqa: E731
            _lib.check(_lib.load().nf_scale3(p(g0), n(g0), p(g1), n(g1), p(gp), n(gp), gs.data_ptr(), p(o0), p(o1), p(op), _lib.stream()), "nf_scale3")
            return o0, o1, None, op, None, None, None, None
        return g0 * g, (None if g1 is None else g1 * g), None, (None if gp is None else gp * g), None, None, None, None


def e2e_loss(out, rgbs, n_views, fine, pos=None, bounds=None, w_boundary=0.0):
    """The end-to-end step's loss: sum over views of MSE(rgb0_v) [+ MSE(rgb1_v)] + w_boundary * mean |pos - clamp(pos, bounds)|
    (equally sized views).  GPU tensors only (the HIP entry point); `bounds` = ((lo_x, lo_y, lo_z), (hi_x, hi_y, hi_z))."""
    use_pos = pos is not None and w_boundary != 0.0
    lo, hi = bounds if use_pos else ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0))
    return _E2ELossFn.apply(out["rgb0"], out["rgb1"] if fine else None, rgbs, pos if use_pos else None, lo, hi, w_boundary if use_pos else 0.0,
                            rgbs.numel() // n_views)


class PixelSampler:
    """Draws the per-view pixel selections `rng.choice(n, ray_chunk, replace=False)` (trainer_renderer.py:119) in the
    reference's order, one step ahead on a background thread: the draw is a full 160 000-element shuffle (1.4 ms per
    view on the host) and would otherwise sit between two GPU steps.  Same RNG stream, same indices.
    Every draw records the generator state it started from, and close() joins the worker and rewinds `rng` to the state
    before the first selection nobody consumed, so after train() the stream is exactly where a sampler without read-ahead
    would have left it — PROVIDED the sampler is the stream's only consumer while it is alive: in native mode the worker
    copies the MT19937 state once and never touches `rng` again, so a foreign draw in between (a dataset rotation, an eval
    hook, user code) would be silently rewound or overwritten by close().  close() therefore compares the stream with the
    state the sampler last left it in and warns when somebody else drew from it (the foreign draws are then lost: the
    stream is set to the sampler's own position, as documented).  Use it under try / finally (trainers do): a sampler that
    is never closed leaves a daemon thread polling.  An exception in the worker is re-raised by next()."""

    def __init__(self, rng, n_views, ray_chunk, n_pixels_of_step, first_step, depth=2):
        import queue
        import threading
        self.rng, self.n_views, self.ray_chunk, self.n_of = rng, n_views, ray_chunk, n_pixels_of_step
        self.q = queue.Queue(maxsize=depth)
        self.step = first_step
        self._stop = threading.Event()
        self._pending = None          # (step, sels, state_before) drawn but not yet queued when the stop flag was seen
        self._native_end = None
        try:
            self._state_at_start = rng.get_state()
        except Exception:
            self._state_at_start = None
        self.t = threading.Thread(target=self._run, daemon=True)
        self.t.start()

    def _native_state(self):
        """(key, pos, tail) when the stream is numpy's legacy MT19937 and the library is built: the draws then run in
        nf_host_choice_mt19937 (bit-identical indices and generator state, outside the GIL — a numpy draw holds the GIL
        for its whole 160 000-element shuffle, 4 x 1.4 ms per step, and starves the thread that feeds the GPU)."""
        try:
            st = self.rng.get_state()
            if st[0] != 'MT19937':
                return None
            from . import _lib
            _lib.load()
            return [np.array(st[1], dtype=np.uint32), int(st[2]), tuple(st[3:])]
        except Exception:       # no get_state (a Generator), library not built: numpy draws
            return None

    def _draw(self, nat, n):
        """One step's selections and the generator state they started from."""
        if nat is None:
            state = self.rng.get_state()
            return [self.rng.choice(n, size=[self.ray_chunk], replace=False) for _ in range(self.n_views)], state
        import ctypes
        from . import _lib
        lib = _lib.load()
        key, pos, tail = nat
        state = ('MT19937', key.copy(), pos) + tail
        if n < self.ray_chunk:
            raise ValueError("Cannot take a larger sample than population when 'replace=False'")
        cpos = ctypes.c_int(pos)
        sels = []
        for _ in range(self.n_views):
            out = np.empty(self.ray_chunk, dtype=np.int64)
            _lib.check(lib.nf_host_choice_mt19937(key.ctypes.data, ctypes.addressof(cpos), n, self.ray_chunk, out.ctypes.data),
                       "nf_host_choice_mt19937")
            sels.append(out)
        nat[1] = cpos.value
        return sels, state

    def _run(self):
        import queue
        step = self.step
        nat = self._native_state()
        try:
            while not self._stop.is_set():
                sels, state = self._draw(nat, self.n_of(step))
                item = (step, sels, state)
                while True:
                    if self._stop.is_set():
                        self._pending = item
                        return
                    try:
                        self.q.put(item, timeout=0.05)
                        break
                    except queue.Full:
                        continue
                step += 1
        except BaseException as e:          # surfaced by next(); never leaves the consumer blocked
            self.q.put(("error", e, None))
        finally:
            if nat is not None:             # native draws never touched self.rng: close() moves it
                self._native_end = ('MT19937', nat[0].copy(), nat[1]) + nat[2]

    def next(self, step):
        import queue
        while True:
            try:
                s, sels, _ = self.q.get(timeout=1.0)
                break
            except queue.Empty:
                if not self.t.is_alive() and self.q.empty():
                    raise RuntimeError("PixelSampler worker stopped without producing a selection")
        if s == "error":
            raise sels
        assert s == step, "PixelSampler is strictly sequential"
        return sels

    def _foreign_draws(self):
        """True when `rng` is not where this sampler left it (native mode: where it was when the sampler started; numpy mode:
        the state after the worker's last draw is unknown to us, so only native mode can tell)."""
        if self._state_at_start is None or getattr(self, "_native_end", None) is None:
            return False
        try:
            now = self.rng.get_state()
        except Exception:
            return False
        a = self._state_at_start
        return not (now[0] == a[0] and now[2] == a[2] and np.array_equal(now[1], a[1]))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def close(self):
        if self._stop.is_set() and not self.t.is_alive():
            return
        self._stop.set()
        self.t.join(timeout=10.0)
        if self._foreign_draws():
            import warnings
            warnings.warn("PixelSampler: another consumer drew from the sampler's random stream while the sampler was alive; "
                          "its draws are discarded (the stream is set to the sampler's own position)", RuntimeWarning)
        left = []
        try:
            while True:
                left.append(self.q.get_nowait())
        except Exception:
            pass
        if self._pending is not None:
            left.append(self._pending)
        states = [it[2] for it in left if it[0] != "error" and it[2] is not None]
        if states:                          # rewind to before the first unconsumed draw
            self.rng.set_state(states[0])
        elif getattr(self, "_native_end", None) is not None:
            self.rng.set_state(self._native_end)


class ExponentialLR(torch.optim.lr_scheduler.LambdaLR):
    """lr = base_lr * gamma ** (epoch / decay_epochs)  (utils/lr_schedulers.py:3-12)."""

    def __init__(self, optimizer, decay_epochs, gamma=0.1, last_epoch=-1):
        super().__init__(optimizer, lambda e: gamma ** (e / decay_epochs), last_epoch)


def renderer_train_step(renderer, optimizer, scheduler, particles, views, H, W, step_idx, ray_chunk=1024,
                        precrop_iters=500, rng=np.random, rank=0, world=1, sampler=None):
    """views: list of dicts {cw (3,4), rays (H,W,6), rgb (H*W,3)} on the GPU.  Returns the loss tensor.
    The reference renders the views one after the other; rays are independent, so the views are batched into ONE
    renderer call (per-ray camera position) and the per-view MSEs are taken on slices — same loss, 4x fewer launches.
    The pixel RNG is drawn per view in the reference's order (np.random.choice, trainer_renderer.py:119)."""
    coords = random_sample_coords(H, W, step_idx, precrop_iters)
    sels = sampler.next(step_idx) if sampler is not None else \
        [choice_without_replacement(rng, coords.shape[0], ray_chunk) for _ in views]
    rays, rgbs, ro = gather_view_pixels([v["rays"] for v in views], [v["rgb"] for v in views], [v["cw"] for v in views],
                                        coords, sels, H, W)
    out = renderer(particles, ro, rays, None, None)
    fine = renderer.N_importance > 0
    # (GPU: the views' MSE sums and their gradients in one launch, as RendererTrainer.train_step does)
    total = e2e_loss(out, rgbs, len(views), fine) if rgbs.is_cuda else summed_view_mse(out, rgbs, len(views), fine)
    optimizer.zero_grad()
    total.backward()
    nfdist.allreduce_grads(list(renderer.parameters()), world)
    optimizer.step()
    if scheduler is not None:
        scheduler.step()
    return total.detach()


def make_train_step(net, scene, dev, rank=0, world=1, lr=5e-4, decay_epochs=10000, seed=10):
    """Synthetic warm-up workload for bench.py: 4 views (the synthetic camera), random target colours."""
    H = W = 400
    g = torch.Generator().manual_seed(seed + rank)
    rays = scene["rays"].view(H, W, 6).to(dev)
    cw = scene["c2w"].to(dev)
    views = [dict(cw=cw, rays=rays, rgb=torch.rand(H * W, 3, generator=g).to(dev)) for _ in range(4)]
    P = scene["P"].to(dev)
    for p in net.parameters():
        p.requires_grad_(True)
    opt = make_adam(net.parameters(), lr=lr)
    sched = ExponentialLR(opt, decay_epochs=decay_epochs, gamma=0.1)
    rng = np.random.RandomState(seed + rank)
    state = {"step": 1000}   # past precrop_iters: full-frame sampling (steady state of the 100k-step schedule)
    sampler = PixelSampler(rng, len(views), 1024, lambda s: random_sample_coords(H, W, s, 500).shape[0], state["step"])

    def step():
        loss = renderer_train_step(net, opt, sched, P, views, H, W, state["step"], 1024, 500, rng, rank, world, sampler)
        state["step"] += 1
        return loss

    return step
