"""Host-side mirror of /root/reference/models/transmodel.py (ParticleNet :14-163) and of the Open3D
``ml3d.layers.ContinuousConv`` objects it constructs (:79-98).

Parameter names/shapes follow Open3D so released checkpoints load: ``convX.kernel`` (4,4,4,Cin,Cout),
``convX.bias`` (Cout), ``convX.offset`` (3, buffer), ``denseX.weight/bias``, buffer ``gravity``.
The arithmetic runs in libneurofluid_hip (nf_cconv.hip); the fluid<->fluid neighbour search and the
per-pair interpolation data are computed ONCE per step and shared by all layers (the reference
rebuilds them in each of its five convs).
"""
import ctypes
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import check, ptr


class ContinuousConv(nn.Module):
    """Parameter container + standalone forward with the Open3D layer contract
    ``__call__(inp_features, inp_positions, out_positions, extents)``; exposes ``.nns`` afterwards."""

    def __init__(self, kernel_size, in_channels, filters, activation=None, interpolation='linear',
                 coordinate_mapping='ball_to_cube_volume_preserving', normalize=False, window_function=None,
                 radius_search_ignore_query_points=True, use_bias=True, align_corners=True, **kwargs):
        super().__init__()
        if list(kernel_size) != [4, 4, 4] or interpolation != 'linear' or normalize or not align_corners or \
                coordinate_mapping != 'ball_to_cube_volume_preserving' or activation is not None or not use_bias:
            raise NotImplementedError("only the configuration used by models/transmodel.py:86-95 is implemented")
        self.in_channels, self.filters = in_channels, filters
        # window_function(d^2 / radius^2) -> importance, as Open3D's layer calls it.  The poly6 window of
        # models/transmodel.py:73-77 is fused into nf_cconv_pairs; it is RECOGNISED by evaluating the callable on a
        # probe vector.  Any other callable is evaluated on the device for every pair (it is never silently replaced).
        self.window_function = window_function
        self.use_window = window_function is not None
        self.fused_window = self.use_window and _is_poly6(window_function)
        self.ignore_query_points = radius_search_ignore_query_points
        self.kernel = nn.Parameter(torch.empty(4, 4, 4, in_channels, filters).uniform_(-0.05, 0.05))
        self.bias = nn.Parameter(torch.zeros(filters))
        self.register_buffer('offset', torch.zeros(3))
        self.nns = None

    def forward(self, inp_features, inp_positions, out_positions, extents):
        extent = float(extents)
        radius = 0.5 * extent
        idx, rs, d2 = ops.fixed_radius_search(inp_positions, out_positions, radius, self.ignore_query_points)
        self.nns = SimpleNamespace(neighbors_index=idx, neighbors_row_splits=rs, neighbors_distance=d2)
        pw, pc = cconv_pairs(inp_positions, out_positions, rs, idx, d2, extent, self.use_window,
                             window_function=None if (self.fused_window or not self.use_window) else self.window_function)
        zero_w = torch.zeros(self.filters, self.in_channels, device=inp_features.device)
        zero_b = torch.zeros(self.filters, device=inp_features.device)
        return cconv_layer(inp_features, self.kernel, self.bias, zero_w, zero_b, rs, idx, pw, pc, relu=False)


# ------------------------------------------------------------------------------------------------
def _window_poly6(r_sqr):
    """models/transmodel.py:73-77."""
    return torch.clamp((1 - r_sqr) ** 3, 0, 1)


def _is_poly6(fn):
    """True when `fn` IS the poly6 window on a probe that covers the clamp on both sides."""
    probe = torch.tensor([-0.25, 0.0, 1e-3, 0.1, 0.25, 0.5, 0.75, 0.9, 0.999, 1.0, 1.5], dtype=torch.float32)
    try:
        got = fn(probe.clone())
    except Exception:
        return False
    return isinstance(got, torch.Tensor) and got.shape == probe.shape and torch.equal(got.float(), _window_poly6(probe))


def cconv_pairs(inp_pos, out_pos, row_splits, nbr, d2, extent, use_window=True, negate=False, window_function=None):
    """Per-pair interpolation data.  window_function=None with use_window: the fused poly6 window; a callable: the
    pairs are built unwindowed and the 8 corner weights of each pair are scaled by window_function(d^2 / radius^2)."""
    lib = _lib.load()
    nnz = nbr.shape[0]
    cap = ops.round_pairs(nnz)           # bucketed: see ops.round_pairs
    pw = torch.empty(cap * 8, dtype=torch.float32, device=inp_pos.device)[:max(nnz, 1) * 8]
    pc = torch.empty(cap * 8, dtype=torch.uint8, device=inp_pos.device)[:max(nnz, 1) * 8]
    fused = use_window and window_function is None
    check(lib.nf_cconv_pairs(ptr(inp_pos), ptr(out_pos), ptr(row_splits), ptr(nbr), ptr(d2), out_pos.shape[0],
                             float(extent), int(fused), int(negate), ptr(pw), ptr(pc), _lib.stream()), "nf_cconv_pairs")
    if use_window and window_function is not None and nnz > 0:
        radius = 0.5 * float(extent)
        imp = window_function(d2[:nnz] / (radius * radius)).to(torch.float32)
        pw[:nnz * 8].view(nnz, 8).mul_(imp.view(nnz, 1))
    return pw, pc


def cconv_layer(x, kernel, bias, dense_w, dense_b, row_splits, nbr, pw, pc, relu, residual=None):
    """y = cconv(act(x)) + Linear(act(x)) (+ residual): transform GEMM + gather."""
    lib = _lib.load()
    x = x.detach().contiguous().float()
    M, cin = x.shape
    cout = kernel.shape[-1]
    n_out = row_splits.shape[0] - 1
    G = torch.empty(M, 65 * cout, dtype=torch.float32, device=x.device)
    st = _lib.stream()
    check(lib.nf_cconv_transform(ptr(x), M, cin, cout, int(relu), ptr(kernel.detach().contiguous()),
                                 ptr(dense_w.detach().contiguous()), ptr(G), st), "nf_cconv_transform")
    y = torch.empty(n_out, cout, dtype=torch.float32, device=x.device)
    check(lib.nf_cconv_gather(ptr(G), cout, ptr(row_splits), 0, None, ptr(nbr), ptr(pw), ptr(pc), ptr(bias.detach().contiguous()),
                              ptr(dense_b.detach().contiguous()), ptr(residual), n_out, ptr(y), st), "nf_cconv_gather")
    return y


class _PitchedNeighbors:
    """`conv.nns` of the fused inference step: the neighbour rows live at a fixed pitch on the device; the CSR view the
    Open3D attribute names promise (neighbors_index / neighbors_row_splits / neighbors_distance) is built on first access."""

    def __init__(self, idx, d2, counts, pitch):
        self._idx, self._d2, self._counts, self._pitch, self._csr = idx, d2, counts, pitch, None

    def _build(self):
        if self._csr is None:
            c = self._counts.clamp(max=self._pitch).long()
            rs = torch.zeros(c.numel() + 1, dtype=torch.int64, device=c.device)
            rs[1:] = torch.cumsum(c, 0)
            n = c.numel()
            slot = torch.arange(self._pitch, device=c.device).unsqueeze(0).expand(n, -1)
            keep = (slot < c.unsqueeze(1)).reshape(-1)
            self._csr = (self._idx[:n * self._pitch][keep], rs, self._d2[:n * self._pitch][keep])
        return self._csr

    neighbors_index = property(lambda self: self._build()[0])
    neighbors_row_splits = property(lambda self: self._build()[1])
    neighbors_distance = property(lambda self: self._build()[2])


class ParticleNet(nn.Module):
    def __init__(self, kernel_size=[4, 4, 4], radius_scale=1.5, coordinate_mapping='ball_to_cube_volume_preserving',
                 interpolation='linear', use_window=True, particle_radius=0.025, timestep=1 / 50,
                 gravity=(0, -9.81, 0), other_feats_channels=0):
        super().__init__()
        if other_feats_channels != 0:
            raise NotImplementedError("other_feats_channels > 0 is never used by the reference callers")
        self.layer_channels = [32, 64, 64, 3]
        self.coordinate_mapping, self.interpolation, self.use_window = coordinate_mapping, interpolation, use_window
        self.kernel_size, self.radius_scale, self.particle_radius = kernel_size, radius_scale, particle_radius
        self.filter_extent = np.float32(6 * self.radius_scale * self.particle_radius)
        self.time_step = timestep
        self.register_buffer('gravity', torch.FloatTensor(gravity))
        window = self._window_poly6 if use_window else None      # models/transmodel.py:82-85

        def conv(cin, cout):
            return ContinuousConv(kernel_size=kernel_size, in_channels=cin, filters=cout, activation=None,
                                  interpolation=interpolation, coordinate_mapping=coordinate_mapping, normalize=False,
                                  window_function=window, radius_search_ignore_query_points=True)

        self.conv0_fluid = conv(4, 32)
        self.conv0_obstacle = conv(3, 32)
        self.dense0_fluid = nn.Linear(4, 32)
        torch.nn.init.xavier_uniform_(self.dense0_fluid.weight)
        torch.nn.init.zeros_(self.dense0_fluid.bias)
        self.convs, self.denses = [], []
        for i in range(1, 4):
            cin = self.layer_channels[i - 1] * (3 if i == 1 else 1)
            cout = self.layer_channels[i]
            setattr(self, f'dense{i}', nn.Linear(cin, cout))
            setattr(self, f'conv{i}', conv(cin, cout))
            self.denses.append(getattr(self, f'dense{i}'))
            self.convs.append(getattr(self, f'conv{i}'))
        self._box_cache = (None, None, None)
        self.num_fluid_neighbors = None
        self._graph_cfg, self._graph, self._nnz_seen = None, None, None
        # fused inference step (nf_trans.hip): CSR capacities in pairs per particle (dense SPH fluid at this radius has
        # ~40-50 fluid neighbours; the container contributes < 30), persistent buffers per particle count
        self.max_fluid_neighbors, self.max_box_neighbors = 128, 64
        self.fused_inference = True
        self._fused = None

    _window_poly6 = staticmethod(_window_poly6)

    @property
    def pos_correction(self):
        y3 = getattr(self, "_y3", None)
        return None if y3 is None else y3 * (1.0 / 128)

    # ------------------------------------------------------------------
    def integrate_pos_vel(self, pos, vel):
        lib = _lib.load()
        n = pos.shape[0]
        pos_new, vel_new = torch.empty_like(pos), torch.empty_like(vel)
        feats = torch.empty(n, 4, dtype=torch.float32, device=pos.device)
        g = (ctypes.c_float * 3)(*[float(v) for v in self._gravity_host()])
        check(lib.nf_trans_integrate(ptr(pos), ptr(vel), g, float(self.time_step), n, ptr(pos_new), ptr(vel_new),
                                     ptr(feats), _lib.stream()), "nf_trans_integrate")
        return pos_new, vel_new, feats

    def _gravity_host(self):
        if getattr(self, "_g_host", None) is None or self._g_version != self.gravity._version:
            self._g_host = self.gravity.detach().cpu().tolist()
            self._g_version = self.gravity._version
        return self._g_host

    def _box_grid(self, box):
        key = (box.data_ptr(), box._version, box.shape[0])
        if self._box_cache[0] != key:
            # the entry keeps `box` (an alias of its storage) alive, so the block cannot be freed and handed to a NEW
            # tensor with the same (ptr, version, N) while the entry exists
            self._box_cache = (key, ops.build_grid(box, 0.5 * float(self.filter_extent), firstk=False), box)
        return self._box_cache[1]

    def update_pos_vel(self, pos, pos_new, y3):
        lib = _lib.load()
        pc, vc = torch.empty_like(pos), torch.empty_like(pos)
        check(lib.nf_trans_update(ptr(pos), ptr(pos_new), ptr(y3), 1.0 / 128, float(self.time_step), pos.shape[0], ptr(pc),
                                  ptr(vc), _lib.stream()), "nf_trans_update")
        return pc, vc

    # ------------------------------------------------------------------
    def forward(self, pos, vel, box, box_feats, feats=None, fixed_radius_search_hash_table=None):
        if feats is not None:
            raise NotImplementedError("other feats are never passed by the reference callers")
        if torch.is_grad_enabled() and (pos.requires_grad or vel.requires_grad or
                                        any(p.requires_grad for p in self.parameters())):
            from .autograd_bwd import particle_net_with_grad
            return particle_net_with_grad(self, pos, vel, box, box_feats)
        with torch.no_grad():
            if self._graph_cfg is not None:
                return self._graph_step(pos, vel, box, box_feats)
            if self.fused_inference and self._fused_ok(pos, box):
                return self._forward_fused(pos, vel, box, box_feats)
            return self._forward_impl(pos, vel, box, box_feats)[:3]

    # ------------------------------------------------------------------
    # Fused inference step: 9 launches, no host round trip.  prepare (integrate + fluid grid, one
    # workgroup) -> search (fluid and box neighbours + pair interpolation data in one sweep) -> conv0 (obstacle + fluid +
    # dense) -> 3 x (transform GEMM, gather), the last gather with the position / velocity update fused.
    # Neighbour rows have a fixed pitch (max_fluid_neighbors / max_box_neighbors per particle); a particle with more
    # neighbours gets NaN outputs on the device and the next call that finds the (asynchronously copied) overflow record
    # raises.
    def _fused_ok(self, pos, box):
        lib = _lib.load()
        if getattr(self, "_fused_limits", None) is None:
            mp, mc = ctypes.c_int(), ctypes.c_int()
            lib.nf_trans_prepare_limits(ctypes.byref(mp), ctypes.byref(mc))
            self._fused_limits = (mp.value, mc.value)
        n = pos.shape[0]
        if n < 1 or n > self._fused_limits[0]:
            return False
        bbox = self._scene_bbox(box.detach())
        cell = 0.5 * float(self.filter_extent)
        cells = 1
        for d in range(3):
            cells *= int((bbox[3 + d] - bbox[d]) / cell) + 1
        return cells + n <= self._fused_limits[1]        # cell counters + scatter list share one workgroup's LDS

    def _fused_buffers(self, n, dev, bbox):
        st = self._fused
        key = (n, str(dev), bbox, self.max_fluid_neighbors, self.max_box_neighbors)
        if st is not None and st["key"] == key:
            return st
        lib = _lib.load()
        radius = 0.5 * float(self.filter_extent)
        bb = (ctypes.c_float * 6)(*[float(v) for v in bbox])
        pitch_f, pitch_b = int(self.max_fluid_neighbors), int(self.max_box_neighbors)
        cap_f, cap_b = n * pitch_f, n * pitch_b
        f32, i32, i64, u8 = torch.float32, torch.int32, torch.int64, torch.uint8
        E = lambda *shape, dtype=f32: torch.empty(*shape, dtype=dtype, device=dev)      # noqa: E731
        st = dict(key=key, bb=bb, pitch=(pitch_f, pitch_b),
                  grid_ws=E(lib.nf_grid_workspace_bytes(n, radius, bb), dtype=u8), pos_new=E(n, 3), vel_new=E(n, 3), feats=E(n, 4),
                  counts2=E(2 * n, dtype=i32), overflow=torch.zeros(2, dtype=i64, device=dev),
                  idx_f=E(cap_f, dtype=i32), d2_f=E(cap_f), pw_f=E(cap_f * 8), pc_f=E(cap_f * 8, dtype=u8),
                  idx_b=E(cap_b, dtype=i32), d2_b=E(cap_b), pw_b=E(cap_b * 8), pc_b=E(cap_b * 8, dtype=u8),
                  a0=E(n, 96), a1=E(n, 64), a2=E(n, 64), y3=E(n, 3), G=E(n * 65 * 64),
                  pending=[], slots=[torch.empty(2, dtype=i64).pin_memory() for _ in range(4)], step=0)
        self._fused = st
        return st

    def check_capacity(self, wait=False):
        """Raises if a particle of a finished fused step had more neighbours than its row pitch (its outputs were set to NaN
        on the device).  wait=True blocks until every launched step has reported."""
        st = self._fused
        if st is None:
            return
        keep = []
        for ev, slot in st["pending"]:
            if wait:
                ev.synchronize()
            if ev.query():
                f, b = slot.tolist()
                if f > st["pitch"][0] or b > st["pitch"][1]:
                    st["pending"] = []
                    st["overflow"].zero_()
                    raise RuntimeError(f"ParticleNet fused step: a particle with {f} fluid / {b} box neighbours (0 = within "
                                       f"bounds) exceeds the capacities {st['pitch']} per particle (its outputs are NaN); raise "
                                       "ParticleNet.max_fluid_neighbors / max_box_neighbors")
            else:
                keep.append((ev, slot))
        st["pending"] = keep

    def _forward_fused(self, pos, vel, box, box_feats):
        lib = _lib.load()
        stream = _lib.stream()
        pos = pos.detach().contiguous().float()
        vel = vel.detach().contiguous().float()
        box = box.detach().contiguous().float()
        box_feats = box_feats.detach().contiguous().float()
        n, dev = pos.shape[0], pos.device
        extent = float(self.filter_extent)
        radius = 0.5 * extent
        bbox = self._scene_bbox(box)
        st = self._fused_buffers(n, dev, bbox)
        self.check_capacity()
        if len(st["pending"]) >= len(st["slots"]):          # every report slot in flight: wait for the oldest
            st["pending"][0][0].synchronize()
            self.check_capacity()
        pitch_f, pitch_b = st["pitch"]
        bgrid = self._box_grid(box)
        g = (ctypes.c_float * 3)(*[float(v) for v in self._gravity_host()])
        check(lib.nf_trans_prepare(ptr(pos), ptr(vel), g, float(self.time_step), n, radius, st["bb"], ptr(st["grid_ws"]),
                                   st["grid_ws"].numel(), ptr(st["pos_new"]), ptr(st["vel_new"]), ptr(st["feats"]), stream),
              "nf_trans_prepare")
        nn = torch.empty(n, dtype=torch.float32, device=dev)
        check(lib.nf_trans_search(ptr(st["grid_ws"]), ptr(bgrid.ws), ptr(st["pos_new"]), n, radius, extent, int(self.use_window),
                                  pitch_f, pitch_b, ptr(st["counts2"]), ptr(nn), ptr(st["idx_f"]), ptr(st["d2_f"]), ptr(st["pw_f"]),
                                  ptr(st["pc_f"]), ptr(st["idx_b"]), ptr(st["d2_b"]), ptr(st["pw_b"]), ptr(st["pc_b"]), stream),
              "nf_trans_search")
        c0o, c0f, d0 = self.conv0_obstacle, self.conv0_fluid, self.dense0_fluid
        check(lib.nf_trans_conv0(ptr(box_feats), ptr(st["feats"]), ptr(st["counts2"]), pitch_f, pitch_b, n, ptr(st["idx_f"]),
                                 ptr(st["pw_f"]), ptr(st["pc_f"]), ptr(st["idx_b"]), ptr(st["pw_b"]), ptr(st["pc_b"]),
                                 ptr(c0o.kernel.detach()), ptr(c0o.bias.detach()), ptr(c0f.kernel.detach()), ptr(c0f.bias.detach()),
                                 ptr(d0.weight.detach()), ptr(d0.bias.detach()), ptr(st["a0"]), stream), "nf_trans_conv0")
        cnt_f, cnt_b = st["counts2"][:n], st["counts2"][n:]
        prev = st["a0"]
        outs = [st["a1"], st["a2"], st["y3"]]
        pos_c, vel_c = torch.empty_like(pos), torch.empty_like(pos)
        for li, (conv, dense) in enumerate(zip(self.convs, self.denses)):
            cin, cout = prev.shape[1], conv.kernel.shape[-1]
            check(lib.nf_cconv_transform(ptr(prev), n, cin, cout, 1, ptr(conv.kernel.detach()), ptr(dense.weight.detach()),
                                         ptr(st["G"]), stream), "nf_cconv_transform")
            y = outs[li]
            if li < 2:
                res = prev if dense.out_features == cin else None
                check(lib.nf_cconv_gather(ptr(st["G"]), cout, None, pitch_f, ptr(cnt_f), ptr(st["idx_f"]), ptr(st["pw_f"]),
                                          ptr(st["pc_f"]), ptr(conv.bias.detach()), ptr(dense.bias.detach()), ptr(res), n, ptr(y),
                                          stream), "nf_cconv_gather")
            else:
                check(lib.nf_cconv_gather_update(ptr(st["G"]), pitch_f, ptr(cnt_f), ptr(st["idx_f"]), ptr(st["pw_f"]), ptr(st["pc_f"]),
                                                 ptr(conv.bias.detach()), ptr(dense.bias.detach()), n, ptr(y), ptr(pos),
                                                 ptr(st["pos_new"]), 1.0 / 128, float(self.time_step), pitch_b, ptr(cnt_b),
                                                 ptr(st["overflow"]), ptr(pos_c), ptr(vel_c), stream), "nf_cconv_gather_update")
            prev = y
        # overflow record -> pinned host slot, checked by a later call (or check_capacity(wait=True))
        slot = st["slots"][st["step"] % len(st["slots"])]
        st["step"] += 1
        slot.copy_(st["overflow"], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        st["pending"].append((ev, slot))
        self.num_fluid_neighbors = nn
        self._y3 = st["y3"]
        self.conv0_fluid.nns = _PitchedNeighbors(st["idx_f"], st["d2_f"], cnt_f, pitch_f)
        return pos_c, vel_c, nn

    # ------------------------------------------------------------------
    # HIP-graph replay of the inference step.  The step is launch-bound (about 45 launches for 0.4 ms of GPU work at
    # 5 k particles) and its only data-dependent size is the pair count; with the CSR sized by a capacity
    # (max_neighbors per particle) there is no host round trip left, so the whole step is captured once per
    # (particle count, container) and replayed.  An overflow of the capacity cannot pass silently: the outputs are
    # poisoned with NaN inside the graph and the host raises at its next periodic check.
    def enable_step_graph(self, max_fluid_neighbors=128, max_box_neighbors=64, check_every=32):
        self._graph_cfg = dict(f=int(max_fluid_neighbors), b=int(max_box_neighbors), every=int(check_every))
        self._graph = None
        return self

    def disable_step_graph(self):
        self._graph_cfg, self._graph = None, None
        return self

    def _graph_step(self, pos, vel, box, box_feats):
        cfg = self._graph_cfg
        n = pos.shape[0]
        key = (n, box.data_ptr(), box._version, box_feats.data_ptr(), str(pos.device))
        if self._graph is None or self._graph["key"] != key:
            sp, sv = pos.detach().clone().float().contiguous(), vel.detach().clone().float().contiguous()
            cap = (n * cfg["f"], n * cfg["b"])

            def body():
                pc, vc, nn, _ = self._forward_impl(sp, sv, box, box_feats, nnz_cap=cap)
                ok = (self._nnz_seen[0] <= cap[0]) & (self._nnz_seen[1] <= cap[1])
                nan = torch.full((), float("nan"), device=pc.device)
                return torch.where(ok, pc, nan), torch.where(ok, vc, nan), nn, self._nnz_seen

            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):          # warm-up off the capture (caches: box grid, scene bbox, gravity)
                for _ in range(2):
                    body()
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                outs = body()
            self._graph = dict(key=key, g=g, sp=sp, sv=sv, outs=outs, cap=cap, count=0, refs=(box, box_feats))
        G = self._graph
        G["sp"].copy_(pos)
        G["sv"].copy_(vel)
        G["g"].replay()
        pc, vc, nn, seen = G["outs"]
        G["count"] += 1
        if G["count"] % cfg["every"] == 0:
            f, b = seen.tolist()
            if f > G["cap"][0] or b > G["cap"][1]:
                raise RuntimeError(f"ParticleNet step graph: {f} fluid / {b} box pairs exceed the capacity {G['cap']}; "
                                   "raise max_fluid_neighbors / max_box_neighbors in enable_step_graph()")
        self.num_fluid_neighbors = nn
        return pc.clone(), vc.clone(), nn.clone()

    def _forward_impl(self, pos, vel, box, box_feats, keep=False, nnz_cap=None):
        """nnz_cap = (fluid, box) pair capacities: no host sync (the CSR buffers are sized by the caller's bound
        instead of the exact count) — the form that can be captured in a HIP graph."""
        lib = _lib.load()
        st = _lib.stream()
        pos = pos.detach().contiguous().float()
        vel = vel.detach().contiguous().float()
        box = box.detach().contiguous().float()
        box_feats = box_feats.detach().contiguous().float()
        n = pos.shape[0]
        extent = float(self.filter_extent)
        radius = 0.5 * extent
        pos_new, vel_new, fluid_feats = self.integrate_pos_vel(pos, vel)
        # B2: one fluid->fluid and one box->fluid search per step (device-side scans), one sync for nnz
        bbox = self._scene_bbox(box)
        fgrid = ops.build_grid(pos_new, radius, bbox, firstk=False)
        bgrid = self._box_grid(box)
        f_rs = ops.radius_row_splits(fgrid, pos_new, radius, True)
        b_rs = ops.radius_row_splits(bgrid, pos_new, radius, True)
        if nnz_cap is None:
            nnz_f, nnz_b = torch.stack([f_rs[-1], b_rs[-1]]).tolist()
        else:
            nnz_f, nnz_b = int(nnz_cap[0]), int(nnz_cap[1])
            self._nnz_seen = torch.stack([f_rs[-1], b_rs[-1]])
        f_idx, f_d2 = ops.radius_fill(fgrid, pos_new, radius, f_rs, nnz_f, True)
        b_idx, b_d2 = ops.radius_fill(bgrid, pos_new, radius, b_rs, nnz_b, True)
        if nnz_cap is not None:      # consumers index the pair arrays through row_splits: never past the capacity
            f_rs, b_rs = f_rs.clamp(max=nnz_f), b_rs.clamp(max=nnz_b)
        f_pw, f_pc = cconv_pairs(pos_new, pos_new, f_rs, f_idx, f_d2, extent, self.use_window)
        b_pw, b_pc = cconv_pairs(box, pos_new, b_rs, b_idx, b_d2, extent, self.use_window)
        self.conv0_fluid.nns = SimpleNamespace(neighbors_index=f_idx[:nnz_f], neighbors_row_splits=f_rs,
                                               neighbors_distance=f_d2[:nnz_f])
        # layer 0: [obstacle | fluid | dense] -> (n, 96)   (:116-120)
        a0 = torch.empty(n, 96, dtype=torch.float32, device=pos.device)
        c0o, c0f, d0 = self.conv0_obstacle, self.conv0_fluid, self.dense0_fluid
        check(lib.nf_cconv_small(ptr(box_feats), 3, ptr(b_rs), ptr(b_idx), ptr(b_pw), ptr(b_pc), ptr(c0o.kernel.detach()),
                                 ptr(c0o.bias.detach()), n, ptr(a0), 96, 0, None, None, None, 0, st), "conv0_obstacle")
        check(lib.nf_cconv_small(ptr(fluid_feats), 4, ptr(f_rs), ptr(f_idx), ptr(f_pw), ptr(f_pc), ptr(c0f.kernel.detach()),
                                 ptr(c0f.bias.detach()), n, ptr(a0), 96, 32, ptr(fluid_feats), ptr(d0.weight.detach()),
                                 ptr(d0.bias.detach()), 64, st), "conv0_fluid")
        ans = [a0]
        for conv, dense in zip(self.convs, self.denses):
            prev = ans[-1]
            res = prev if dense.out_features == prev.shape[-1] else None
            ans.append(cconv_layer(prev, conv.kernel, conv.bias, dense.weight, dense.bias, f_rs, f_idx, f_pw, f_pc,
                                   relu=True, residual=res))
        self.num_fluid_neighbors = (f_rs[1:] - f_rs[:-1]).to(torch.float32)   # reduce_subarrays_sum(ones) (:135-138)
        self._y3 = ans[-1]                       # pos_correction (models/transmodel.py:147) is derived on access
        pos_c, vel_c = self.update_pos_vel(pos, pos_new, ans[-1])
        aux = dict(ans=ans, f=(f_rs, f_idx, f_pw, f_pc), f_d2=f_d2, b=(b_rs, b_idx, b_pw, b_pc), pos_new=pos_new,
                   vel_new=vel_new, fluid_feats=fluid_feats) if keep else None
        return pos_c, vel_c, self.num_fluid_neighbors, aux

    def _scene_bbox(self, box):
        """Static grid bounds from the container (cached: no per-step sync); particles that leave it are
        clamped into border cells, which keeps the search exact (include/neurofluid_hip.h)."""
        key = (box.data_ptr(), box._version, box.shape[0])
        if getattr(self, "_bbox_key", None) != key:          # _bbox_ref pins the storage (see _box_grid)
            lo, hi = torch.aminmax(box, dim=0)
            self._bbox = tuple((lo - 0.5).tolist()) + tuple((hi + 0.5).tolist())
            self._bbox_key, self._bbox_ref = key, box
        return self._bbox
