"""One optimiser step of the warm-up trainer (/root/reference/trainer/trainer_renderer.py:75-143):
for each of the 4 warm-up views sample `ray_chunk` random pixels of frame 0 (centre crop for the first
`precrop_iters` steps, trainer/basetrainer.py:171-193), render them from the GT particles, loss = sum over views
of MSE(rgb0) + MSE(rgb1); zero_grad / backward / Adam / ExponentialLR (utils/lr_schedulers.py:3-12).
Used by bench.py --workload train and by the Trainer in neurofluid_amd/trainers.py."""
import ctypes
import os

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

    def _table(self, gi, ps):
        """The pointer / size table of param group gi over the parameters ps (those with a gradient), (re)built when the list or the
        state tensors changed."""
        import ctypes
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
        return tab

    # ---- the step inside a replayed HIP graph (GraphedRendererStep): the per-step scalars travel through device memory
    def graph_scalars(self, gi=None):
        """(step_size, bc2_sqrt) of the NEXT step of param group gi; advances the group's shared step counter as step() does.  None when the
        group cannot run from a graph (per-tensor step counts, a parameter without a gradient, non-default switches).  gi=None: the
        optimizer's only group (None if it has several)."""
        if gi is None:
            if len(self.param_groups) != 1:
                return None
            gi = 0
        if not self._eligible():
            return None
        group = self.param_groups[gi]
        ps = [p for p in group["params"] if p.grad is not None]
        if not ps or len(ps) != len(group["params"]):
            return None
        tab = self._table(gi, ps)
        if tab["step"] is None:
            return None
        beta1, beta2 = group["betas"]
        tab["step"] += 1
        t = float(tab["step"])
        return float(group["lr"]) / (1.0 - beta1 ** t), (1.0 - beta2 ** t) ** 0.5

    def graph_rewind(self, steps, gi=0):
        """Take back `steps` counter advances (replayed steps that were skipped on the device and are redone)."""
        tab = self._tables.get(gi)
        if tab is not None and tab["step"] is not None:
            tab["step"] -= steps

    @torch.no_grad()
    def graph_enqueue(self, sched_dev, skip_dev, gi=0):
        """The step's launch of param group gi with its scalars in device memory (nf_adam_step_dev); called under capture, after backward."""
        from . import _lib
        group = self.param_groups[gi]
        ps = list(group["params"])
        tab = self._table(gi, ps)
        beta1, beta2 = group["betas"]
        for k, p in enumerate(ps):
            g = p.grad
            if g is None or not g.is_contiguous() or g.dtype is not torch.float32:
                raise RuntimeError("HipAdam.graph_enqueue: every parameter needs a contiguous fp32 gradient")
            tab["p"][k], tab["g"][k] = p.data_ptr(), g.data_ptr()
        _lib.check(_lib.load().nf_adam_step_dev(len(ps), tab["p"], tab["g"], tab["m"], tab["v"], tab["sz"], sched_dev.data_ptr(),
                                                None if skip_dev is None else skip_dev.data_ptr(), float(beta1), float(beta2),
                                                float(group["eps"]), float(group["weight_decay"]), _lib.stream()), "nf_adam_step_dev")

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
            tab = self._table(gi, ps)
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


# The captured step forks the coarse pass's backward behind the coarse forward (GraphedRendererStep._body); False = both passes' backward
# chains behind the fine forward, on two streams (the eager step's order).
EARLY_COARSE_BACKWARD = os.environ.get("NF_EARLY_COARSE_BACKWARD", "1") != "0"


# Captured steps pack every weight blob at the top of the graph on a side stream (autograd_bwd.prepack_for_capture); False = where the eager
# step packs them, between the dependent kernels.
PREPACK_IN_CAPTURE = os.environ.get("NF_PREPACK_IN_CAPTURE", "1") != "0"


class GraphedRendererStep:
    """The whole warm-up optimiser step (trainer/trainer_renderer.py:94-143: pixel gather -> coarse + fine forward -> loss -> backward ->
    Adam) captured ONCE as a HIP graph and replayed: per step the host uploads the pixel selection and the optimiser's two scalars
    (one pinned staging slot, stream-ordered copies) and launches one graph.  The eager step is a chain of ~40 launches with one host
    round trip in the middle (the passes' row counts, autograd._run_passes); it is GPU-bound on a quiet host, but every host stall lands
    on the GPU's critical path (round 5: 3.39 ms on one box, 4.17 ms on another).  Replayed, the host needs ~0.1 ms per 3 ms step and
    runs one step ahead of the GPU.

    What makes the step capturable:
      * row-sized launches of forward and backward run against the learnt row CAPACITIES of the two passes; the true counts stay on the
        device (kernels clamp to them; nf_nerf_wgrad_dev derives the eager call's row slicing from the device count: same sums);
      * Adam reads its step size / bias correction from device memory (nf_adam_step_dev);
      * a pass that meets more active rows than its capacity sets a sticky poison word on the device (nf_note_overflow) that turns this
        and every following optimiser launch into a no-op.  The host reads each step's record one step late (from mapped pinned memory,
        behind that step's event), raises the capacities, recaptures and REDOES the skipped steps from their kept selections: the
        parameter trajectory is the eager loop's.  (The loss tensor returned for a poisoned step first holds the truncated forward's
        value; the redo writes the true value into the SAME tensor.  verify() settles everything enqueued — the trainer calls it
        before it reads a loss on the host.)
    The same kernels with the same operands as the eager step: losses and parameters agree bit for bit
    (tests/test_gpu_render.py::test_graph_replayed_renderer_step_equals_eager).  Single process (world = 1) only: a data-parallel
    run keeps the eager step with its gradient all-reduce."""

    def __init__(self, renderer, optimizer, particles, views, H, W, ray_chunk):
        from . import _lib
        self.net, self.opt, self.P, self.views, self.H, self.W, self.rc = renderer, optimizer, particles, views, H, W, int(ray_chunk)
        self.dev = particles.device
        self.V = len(views)
        self.graph = None
        self.captures = 0           # diagnostics
        self.redone_steps = 0
        n = self.V * self.rc
        R = n
        self._cap_keys = [(R, renderer.N_samples)] + ([(R, renderer.N_samples + renderer.N_importance)] if renderer.N_importance > 0 else [])
        self.flat_dev = torch.zeros(n, dtype=torch.int64, device=self.dev)
        self.sched_dev = torch.zeros(2, dtype=torch.float32, device=self.dev)
        self.state_dev = torch.zeros(8, dtype=torch.int32, device=self.dev)       # nf_note_overflow's words
        self.ring_host = torch.zeros(64, dtype=torch.int32).pin_memory()          # 8 records, written by the device (mapped memory)
        self._ring_np = self.ring_host.numpy()
        self._ring_dev = _lib.load().nf_pinned_device_ptr(self.ring_host.data_ptr())
        if not self._ring_dev:
            raise RuntimeError("pinned host memory is not mapped into the device address space (nf_pinned_device_ptr)")
        self._stage = [[torch.empty(n, dtype=torch.int64).pin_memory(), torch.empty(2, dtype=torch.float32).pin_memory(), None] for _ in range(4)]
        self._si = 0
        self._launched = []         # [(device step index, flat_host, (ss, bc), event)] not yet checked, oldest first
        self._steps = 0             # replays since the last capture = the device counter
        self._rl = [v["rays"].reshape(H * W, -1).contiguous() for v in views]
        self._gl = [v["rgb"].reshape(H * W, -1).contiguous() for v in views]
        self._cl = [v["cw"].contiguous() for v in views]
        self._keep = None
        self._recapture = False
        self.rows_total = self.steps_total = 0      # active MLP rows / settled steps (bench.py's roofline)

    @staticmethod
    def eligible(renderer, optimizer, particles, world):
        return bool(world == 1 and particles.is_cuda and isinstance(optimizer, HipAdam) and len(optimizer.param_groups) == 1 and
                    not particles.requires_grad and getattr(renderer, "mlp_dtype", "fp32") == "fp32" and
                    all(p.requires_grad for p in renderer.parameters()))

    def ready(self):
        """True once an eager step of this shape has learnt the passes' row capacities (the capture needs them)."""
        return all(k in self.net.train_row_cap for k in self._cap_keys) and all(p.grad is not None for p in self.net.parameters())

    # ------------------------------------------------------------------
    def _body(self):
        """One step's launches (runs under capture)."""
        from . import _lib
        lib = _lib.load()
        V, rc, H, W = self.V, self.rc, self.H, self.W
        C = self._gl[0].shape[1]
        rays = torch.empty(V * rc, 6, device=self.dev)
        rgbs = torch.empty(V * rc, C, device=self.dev)
        ro = torch.empty(V * rc, 3, device=self.dev)
        arr = ctypes.c_void_p * V
        _lib.check(lib.nf_gather_view_pixels(V, arr(*[t.data_ptr() for t in self._rl]), arr(*[t.data_ptr() for t in self._gl]),
                                             arr(*[t.data_ptr() for t in self._cl]), rc, C, H * W, self.flat_dev.data_ptr(),
                                             rays.data_ptr(), rgbs.data_ptr(), ro.data_ptr(), _lib.stream()), "nf_gather_view_pixels")
        # forward, loss and backward WITHOUT the autograd engine: the step's graph is known (two render passes -> one loss), and
        # the engine brings its own stream bookkeeping (AccumulateGrad streams, leaf-stream joins) into the capture
        from .autograd import _run_passes
        from .autograd_bwd import render_backward, _nerf_params, _pass_backward, _side_stream
        net = self.net
        fine = net.N_importance > 0
        cap = {"counts": [], "caps": []}
        early = {}
        f3 = ctypes.c_float * 3
        if fine and EARLY_COARSE_BACKWARD:
            # The coarse pass's loss term and backward depend on nothing the fine pass produces (two networks, detached importance
            # samples; d loss / d rgb0 = 2 (rgb0 - gt) / N whatever rgb1 is).  Forked onto a side stream right behind the coarse
            # composite, its ~0.4 ms of MLP work fills the ~0.2 ms in which the fine pass's resampling / classify / search / feature
            # launches leave the chip nearly idle and then shares the CUs with the fine forward — instead of competing with the fine
            # pass's backward and weight gradients at the end of the step, where the critical path is.  Same kernels, same operands.
            # Measured (A/B on one box, alternating): 3.06-3.07 vs 3.08-3.09 ms per step — the fine pass's backward + weight gradients
            # shrink by 150 us, but its small front-end kernels wait for CU slots behind the co-running coarse kernels (k_importance_w
            # 62 -> 127 us, k_mlp_pack 8 -> 98 us) and the fine forward starts 170 us later; capturing on a high-priority stream changes
            # nothing (graph branches run on internal queues of equal priority).  The step is bound by the MLP kernels' throughput.
            def after_coarse(p0_, rays_c_, ro_c_, z0_, zt0_):
                cur = torch.cuda.current_stream(self.dev)
                side = _side_stream(self.dev)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    l0 = torch.empty(1, dtype=torch.float32, device=self.dev)
                    g0_ = torch.empty_like(p0_.rgb)
                    _lib.check(lib.nf_e2e_loss(p0_.rgb.data_ptr(), None, rgbs.data_ptr(), p0_.rgb.numel(), rgbs.numel() // V, None, 0,
                                               f3(0, 0, 0), f3(0, 0, 0), 0.0, l0.data_ptr(), g0_.data_ptr(), None, None, _lib.stream()),
                               "nf_e2e_loss")
                    gc_ = _pass_backward(net, net.nerf_coarse, p0_, rays_c_, z0_, zt0_, g0_, True)
                    for t in gc_ + [l0, g0_]:
                        t.record_stream(cur)
                early["gc"], early["side"], early["keep"] = gc_, side, (l0, g0_)
            cap["after_coarse"] = after_coarse
        net._capture = cap
        if PREPACK_IN_CAPTURE:
            from .autograd_bwd import prepack_for_capture
            prepack_for_capture(net, self.dev)
        try:
            p0, p1, rays_c, ro_c, grid = _run_passes(net, self.P, ro, rays, True, fine, save_acts=True)
        finally:
            net._capture = None
        if getattr(net, "_prepacked", None):
            net._prepacked["join"]()        # (idempotent: normally done in front of the coarse MLP launch; a forked stream must be joined inside the capture)
        loss = torch.empty(1, dtype=torch.float32, device=self.dev)
        g0 = torch.empty_like(p0.rgb)
        g1 = torch.empty_like(p1.rgb) if fine else None
        _lib.check(lib.nf_e2e_loss(p0.rgb.data_ptr(), p1.rgb.data_ptr() if fine else None, rgbs.data_ptr(), p0.rgb.numel(), rgbs.numel() // V,
                                   None, 0, f3(0, 0, 0), f3(0, 0, 0), 0.0, loss.data_ptr(), g0.data_ptr(), g1.data_ptr() if fine else None, None,
                                   _lib.stream()), "nf_e2e_loss")
        if "gc" in early:       # the coarse half is already running (or done) on the side stream: the fine half here, then join
            gf = _pass_backward(net, net.nerf_fine, p1, rays_c, p1.z, None, g1, True)
            torch.cuda.current_stream(self.dev).wait_stream(early["side"])
            gc = early["gc"]
        else:
            gc, gf = render_backward(net, p0, p1, rays_c, g0, g1, True)
        grads = list(gc) + (list(gf) if fine else [])
        params = _nerf_params(net)[:len(grads)]
        for p_, g_ in zip(params, grads):
            p_.grad = g_                      # views of the passes' gradient blobs (static storage of the graph's pool)
        c, k = cap["counts"], cap["caps"]
        _lib.check(lib.nf_note_overflow(c[0].data_ptr(), int(k[0]), c[1].data_ptr() if len(c) > 1 else None, int(k[1]) if len(k) > 1 else 0,
                                        self.state_dev.data_ptr(), self._ring_dev, _lib.stream()), "nf_note_overflow")
        self.opt.graph_enqueue(self.sched_dev, self.state_dev[0:1])
        out = {"rgb0": p0.rgb}
        if fine:
            out["rgb1"] = p1.rgb
        self._keep = (out, rgbs, c, p0, p1, grads, early.get("keep"), getattr(net, "_prepacked", None))
        net._prepacked = None
        self.caps = [int(v) for v in k]
        return loss[0]

    def _capture_graph(self):
        self.graph = None
        self._keep = None
        torch.cuda.synchronize(self.dev)
        self.state_dev.zero_()
        self._ring_np[:] = 0
        self._steps = 0
        self._recapture = False
        # gradients the eager steps left must not be what the captured backward accumulates into
        self.opt.zero_grad(set_to_none=True)
        from . import ops
        prof, ops.PROFILE = ops.PROFILE, None      # (timing events cannot be recorded into a capture)
        g = torch.cuda.CUDAGraph()
        try:
            with torch.no_grad(), torch.cuda.graph(g):
                self.loss_static = self._body()
        finally:
            ops.PROFILE = prof
            self.net._prepacked = None
        torch.cuda.synchronize(self.dev)
        self.graph = g
        self.captures += 1

    def _launch(self, flat_host, scal, loss_out=None):
        slot = self._stage[self._si % len(self._stage)]
        self._si += 1
        if slot[2] is not None:
            slot[2].synchronize()           # the copies that last read this staging slot are done
        slot[0].copy_(flat_host)
        slot[1][0], slot[1][1] = scal[0], scal[1]
        self.flat_dev.copy_(slot[0], non_blocking=True)
        self.sched_dev.copy_(slot[1], non_blocking=True)
        slot[2] = torch.cuda.Event()
        slot[2].record()
        self.graph.replay()
        if loss_out is None:
            loss_out = self.loss_static.clone()
        else:
            loss_out.copy_(self.loss_static)        # a redone step: the tensor the caller already holds gets the true value
        ev = torch.cuda.Event()
        ev.record()
        self._launched.append((self._steps, flat_host, scal, ev, loss_out))
        self._steps += 1
        return loss_out

    def _settle(self, keep_in_flight):
        """Reads the records of all launched steps but the newest `keep_in_flight` (waiting for them); on a poisoned step raises the
        capacities, recaptures and redoes every step from it on.  Returns the redone last step's loss, or None."""
        redo_loss = None
        while len(self._launched) > keep_in_flight:
            idx, _, _, ev, _ = self._launched[0]
            ev.synchronize()
            rec = self._ring_np[(idx & 7) * 8:(idx & 7) * 8 + 8]
            if int(rec[2]) != idx + 1:
                raise RuntimeError("GraphedRendererStep: step %d finished without its overflow record (found counter %d)" % (idx, int(rec[2])))
            poisoned, counts = int(rec[0]), [int(rec[3]), int(rec[4])]
            caps = self.net.train_row_cap
            if poisoned:
                todo = list(self._launched)             # this step and everything enqueued behind it were no-ops on the device
                torch.cuda.synchronize(self.dev)
                for key, n, cap in zip(self._cap_keys, counts, self.caps):
                    if n > cap:
                        caps[key] = max(caps.get(key, 0), ops_round_rows(n + n // 2 + 4096))
                self._launched = []
                self._capture_graph()
                for _, flat_host, scal, _, loss_t in todo:
                    redo_loss = self._launch(flat_host, scal, loss_t)
                    self.redone_steps += 1
                continue                                  # the redone steps are settled by the same loop
            self._launched.pop(0)
            self.rows_total += counts[0] + counts[1]
            self.steps_total += 1
            for key, n, cap in zip(self._cap_keys, counts, self.caps):
                if n > cap * 0.9 and caps.get(key, 0) <= cap:       # grow ahead of need (as the eager passes do)
                    caps[key] = ops_round_rows(n + n // 4 + 4096)
                    self._recapture = True                # taken up by the next step()
        return redo_loss

    def step(self, coords, sels):
        """coords (n, 2) host pixel grid, sels[v] = the selected rows for view v (PixelSampler / choice_without_replacement).
        Returns the step's loss (a fresh 0-dim tensor)."""
        yx = torch.cat([coords[torch.as_tensor(s)] for s in sels]).long()
        flat_host = yx[:, 0] * self.W + yx[:, 1]
        if flat_host.numel() != self.V * self.rc:
            raise ValueError("GraphedRendererStep: %d pixels per step expected" % (self.V * self.rc))
        if int(flat_host.min()) < 0 or int(flat_host.max()) >= self.H * self.W:
            raise IndexError("pixel selection outside the %d x %d image" % (self.H, self.W))
        if self.graph is None or self._recapture:
            redo = self._settle(0)          # (a redo in here recaptures with the raised capacities itself)
            if self.graph is None or (self._recapture and redo is None):
                self._capture_graph()
        scal = self.opt.graph_scalars()
        if scal is None:
            raise RuntimeError("GraphedRendererStep: the optimizer cannot step from a graph (see HipAdam.graph_scalars)")
        loss = self._launch(flat_host, scal)
        redo = self._settle(1)          # the previous step's record, with this step already queued behind it
        return redo if redo is not None else loss

    def verify(self):
        """Settle every enqueued step (redoing poisoned ones: their loss tensors are corrected in place)."""
        self._settle(0)

    def last_outputs(self):
        """(renderer result dict, target colours) of the LAST replayed step — static tensors, valid until the next step."""
        return self._keep[0], self._keep[1]


def ops_round_rows(n):
    from . import ops
    return ops._round_rows(n)


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

    use_graph = os.environ.get("NF_TRAIN_GRAPH", "1") != "0" and GraphedRendererStep.eligible(net, opt, P, world)
    gstep = GraphedRendererStep(net, opt, P, views, H, W, 1024) if use_graph else None

    def step():
        if gstep is not None and state["step"] >= 1003 and gstep.ready():       # three eager steps first (they learn the row capacities)
            coords = random_sample_coords(H, W, state["step"], 500)
            loss = gstep.step(coords, sampler.next(state["step"]))
            sched.step()
        else:
            loss = renderer_train_step(net, opt, sched, P, views, H, W, state["step"], 1024, 500, rng, rank, world, sampler)
        state["step"] += 1
        return loss

    step.graphed = gstep
    step.sampler = sampler          # (close() it when done: the read-ahead thread)
    return step
