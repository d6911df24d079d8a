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


import os as _os
import time as _time


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
    check(lib.nf_cconv_gather(ptr(G), cout, ptr(row_splits), ptr(nbr), ptr(pw), ptr(pc), ptr(bias.detach().contiguous()),
                              ptr(dense_b.detach().contiguous()), ptr(residual), n_out, ptr(y), st), "nf_cconv_gather")
    return y


class PairCapacityExceeded(RuntimeError):
    """Raised by the graph-replayed training step (ParticleNet.training_graph) when a step's neighbour pairs did not fit the
    capacities its graphs were captured with: the capacities have been raised, the caller redoes the step (E2ETrainer does)."""


class _LazyCSR:
    """`conv.nns` of a step that ran against pair capacities whose true totals are still on their way to the host: the arrays are
    cut to the true count on first access (which waits for the step's event)."""

    def __init__(self, idx, rs, d2, total_fn):
        self._idx, self.neighbors_row_splits, self._d2, self._total = idx, rs, d2, total_fn

    neighbors_index = property(lambda self: self._idx[:self._total()])
    neighbors_distance = property(lambda self: self._d2[:self._total()])


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


class AsyncStep:
    """A ParticleNet inference step in flight (ParticleNet.step_async)."""

    def __init__(self, pn, handle, out, stream, event):
        self.pn, self.handle, self.out, self.stream, self.event = pn, handle, out, stream, event

    def result(self):
        """(pos, vel, num_fluid_neighbors) of the step, usable on the caller's CURRENT stream."""
        pn = self.pn
        if pn._async_pending is not self:
            raise RuntimeError("AsyncStep.result: this step was already consumed")
        pn._async_pending = None
        if self.out is None:
            with torch.no_grad(), torch.cuda.stream(self.stream):
                self.out = pn._fused_finish(self.handle)       # (an overflowing step is redone on the exact path, on the step's stream)
                self.event = torch.cuda.Event()
                self.event.record(self.stream)
            self.handle = None
        cur = torch.cuda.current_stream(self.out[0].device)
        cur.wait_event(self.event)
        for t in self.out:
            if torch.is_tensor(t):
                t.record_stream(cur)            # allocated on the step's stream, consumed on the caller's
        return self.out


class ParticleNet(nn.Module):
    def __init__(self, kernel_size=[4, 4, 4], radius_scale=1.5, coordinate_mapping='ball_to_cube_volume_preserving',
                 interpolation='linear', use_window=True, particle_radius=0.025, timestep=1 / 50,
                 gravity=(0, -9.81, 0), other_feats_channels=0):
        super().__init__()
        # other_feats_channels > 0 (models/transmodel.py:24,43,50,111-114): extra per-particle input features next to [1, v].  No
        # reference caller uses them; they are served on the exact multi-launch path (the fused step's front kernel is built
        # for the 4-channel input).
        self.other_feats_channels = int(other_feats_channels)
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

        self.conv0_fluid = conv(4 + self.other_feats_channels, 32)
        self.conv0_obstacle = conv(3, 32)
        self.dense0_fluid = nn.Linear(4 + self.other_feats_channels, 32)
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
        # fused inference step (nf_trans.hip): CSR capacities in pairs per particle (dense SPH fluid at this radius has
        # ~40-50 fluid neighbours; the container contributes < 30), persistent buffers per particle count
        self.max_fluid_neighbors, self.max_box_neighbors = 128, 64
        self.fused_inference = True
        self.fused_grow_pitch = True        # on overflow: redo the step exactly AND grow the pitch (False: only redo)
        self.optimistic_pair_capacity = True    # exact path: pair arrays sized by learnt capacities, no mid-step host round trip
        # Opt-in (E2ETrainer sets it): the training step's forward and backward launch sequences (~30 and ~60 small kernels) are
        # captured into HIP graphs per (cloud size, pair capacities, scene) and REPLAYED — the end-to-end step is bound by the
        # host's launch rate, not by the GPU.  One forward may be outstanding per backward (truncated BPTT of length 1); a
        # step whose pairs exceed the captured capacities raises PairCapacityExceeded from backward() (autograd_bwd.py).
        self.training_graph = False
        # build-only switch, arithmetic of the conv1 / conv2 contractions of the fused inference step: "fp32" (default: fp32
        # MFMA, the reference's arithmetic) or "split" (hi + lo fp16 operands, three fp16 MFMAs per product block, fp32
        # accumulate: fp32-LEVEL accuracy — 22-bit products — on the fp16 matrix pipe, which overlaps with the gather)
        self.conv_arith = "fp32"
        # fixed-radius search of the fused step: "auto" (all-pairs up to nf_trans_all_pairs_max_points() particles — for clouds of a
        # few thousand particles n^2 distance tests are cheaper than any grid's chain of dependent round trips —, the cell grid
        # beyond), "grid", "all_pairs".  Same neighbour sets and counts; the order inside a row differs (cell order / index order)
        self.fused_search = "auto"
        self._fused, self._fused_skip = None, 0
        self._async_pending = None          # the AsyncStep in flight (step_async), if any
        self._lib_cached = None

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
        nf = 0 if feats is None else int(feats.shape[-1])
        if nf != self.other_feats_channels:
            raise ValueError(f"feats has {nf} channels, the model was built with other_feats_channels={self.other_feats_channels}")
        if torch.is_grad_enabled() and (pos.requires_grad or vel.requires_grad or (feats is not None and feats.requires_grad) or
                                        any(p.requires_grad for p in self.parameters())):
            from .autograd_bwd import particle_net_with_grad, particle_net_graphed
            if self.training_graph and feats is None and not pos.requires_grad and not vel.requires_grad and self.optimistic_pair_capacity:
                out = particle_net_graphed(self, pos, vel, box, box_feats)
                if out is not None:
                    return out
            return particle_net_with_grad(self, pos, vel, box, box_feats, feats)
        if self._async_pending is not None and self._async_pending.handle is not None:
            raise RuntimeError("ParticleNet.forward: an AsyncStep is in flight (its scratch is the module's): consume it with result() first")
        with torch.no_grad():
            if feats is None and self.fused_inference and self._fused_ok(pos, box):
                return self._forward_fused(pos, vel, box, box_feats)
            return self._forward_impl(pos, vel, box, box_feats, other=feats)[:3]

    def verify_training_step(self):
        """training_graph: compare the LAST graph-replayed forward's true pair totals with the capacities its graphs were captured for — now,
        instead of at the start of backward() — and raise PairCapacityExceeded on overflow (capacities already raised).  For a caller that is
        about to USE the step's outputs for something backward() cannot undo (a logged metric); waits for the forward replay's event."""
        tg = self.__dict__.get("_tgraphs")
        if self.training_graph and tg is not None and tg.key is not None and not getattr(tg, "checked", True):
            from .autograd_bwd import _tg_check
            _tg_check(self, tg)

    # ------------------------------------------------------------------
    # Fused inference step, round 3: ONE C call = prepare (integrate + fluid grid, one workgroup) ->
    # front (search of both clouds, row-entry lists, layer 0) -> three G-free continuous convolutions (gather a patch per
    # point in LDS, contract with the filter on the fp32 matrix pipe; nothing of size n x 64 x C is materialised), the last
    # one with the position / velocity update.  Neighbour rows keep a fixed pitch (max_fluid_neighbors / max_box_neighbors per
    # particle, <= nf_trans_front_max_pitch()).  The reference's search has no cap, so a count above the pitch must not
    # change results: the overflow record leaves the device right behind the front kernel (pinned memory + event, while
    # the convolutions run), the host reads it before forward() returns, and on overflow THIS step is redone on the exact
    # CSR path (_forward_impl) and the pitch grows for the next ones — nothing is poisoned, nothing surfaces later.
    def _fused_ok(self, pos, box):
        lib = self._lib_cached
        if lib is None:
            lib = self._lib_cached = _lib.load()
        if getattr(self, "_fused_limits", None) is None:
            mp, mc = ctypes.c_int(), ctypes.c_int()
            lib.nf_trans_prepare_limits(ctypes.byref(mp), ctypes.byref(mc))
            self._fused_limits = (mp.value, mc.value, lib.nf_trans_front_max_pitch(), lib.nf_trans_all_pairs_max_points())
        n = pos.shape[0]
        if self.fused_search not in ("auto", "grid", "all_pairs"):
            raise ValueError("ParticleNet.fused_search must be 'auto', 'grid' or 'all_pairs'")
        if n < 1 or n > self._fused_limits[0] or self._fused_skip > 0:
            self._fused_skip = max(self._fused_skip - 1, 0)
            return False
        if max(int(self.max_fluid_neighbors), int(self.max_box_neighbors)) > self._fused_limits[2]:
            return False
        if self.fused_search != "grid" and n <= self._fused_limits[3]:
            return True                         # all-pairs search: no grid, no bound on the scene
        if self.fused_search == "all_pairs":
            return False
        # the single-workgroup grid build holds the cell counters and the scatter list in LDS; ask the library for the
        # cell count of THIS bbox (its header code: float32 floor + clamping) instead of re-deriving it here
        bbox = self._scene_bbox(box.detach())
        key = (n, bbox)
        if getattr(self, "_fused_fit", (None, None))[0] != key:
            bb = (ctypes.c_float * 6)(*[float(v) for v in bbox])
            cells = lib.nf_grid_cells(n, 0.5 * float(self.filter_extent), bb)
            self._fused_fit = (key, cells > 0 and cells + n <= self._fused_limits[1])
        return self._fused_fit[1]

    def _fused_buffers(self, n, dev, bbox):
        st = self._fused
        key = (n, str(dev), bbox, int(self.max_fluid_neighbors), int(self.max_box_neighbors), self.conv_arith)
        if st is not None and st["key"] == key:
            return st
        lib = _lib.load()
        radius = 0.5 * float(self.filter_extent)
        bb = (ctypes.c_float * 6)(*[float(v) for v in bbox])
        pitch_f, pitch_b = int(self.max_fluid_neighbors), int(self.max_box_neighbors)
        f32, i32, i64, u8, i16 = torch.float32, torch.int32, torch.int64, torch.uint8, torch.int16
        E = lambda *shape, dtype=f32: torch.empty(*shape, dtype=dtype, device=dev)      # noqa: E731
        max_wg = torch.cuda.get_device_properties(dev).multi_processor_count
        sf = ctypes.c_size_t()
        check(lib.nf_cconv_gf_plan(n, 64, max_wg, None, None, None, ctypes.byref(sf)), "nf_cconv_gf_plan")
        st = dict(key=key, pitch=(pitch_f, pitch_b), max_wg=max_wg,
                  # (zeroed: the step's grid build keeps its cell counters and ticket in here and expects them at rest)
                  grid_ws=torch.zeros(lib.nf_grid_workspace_bytes(n, radius, bb), dtype=u8, device=dev), pos_new=E(n, 3), vel_new=E(n, 3), feats=E(n, 4),
                  counts2=E(2 * n, dtype=i32), idx_f=E(n * pitch_f, dtype=i32), d2_f=E(n * pitch_f),
                  roff=torch.zeros(n * 20, dtype=i16, device=dev), ent=E(n * 4 * pitch_f * 3, dtype=i32),
                  a0=E(n, 96), a1=E(n, 64), a1r=E(n, 64), g3=E(lib.nf_cconv3_workspace_floats(n)), y3=E(n, 3), scratch=E(sf.value),
                  # overflow record: the device keeps the largest count above its pitch (atomicMax) and raises two pinned,
                  # device-visible host words — both written ONLY when a row overflows, re-zeroed by the host after the redo
                  ovf=torch.zeros(2, dtype=i64, device=dev), flag_host=torch.zeros(4, dtype=i32).pin_memory(),
                  done=torch.zeros(1, dtype=i32, device=dev), step=0, wsig=None, packed=None, skey=None)
        st["flag_np"] = st["flag_host"].numpy()
        st["flag_dev"] = lib.nf_pinned_device_ptr(st["flag_host"].data_ptr())
        if not st["flag_dev"]:
            raise RuntimeError("pinned host memory is not mapped into the device address space (nf_pinned_device_ptr)")
        S = _lib.TransStep()
        S.grid_ws, S.grid_ws_bytes = st["grid_ws"].data_ptr(), st["grid_ws"].numel()
        for k in ("pos_new", "vel_new", "feats", "counts2", "idx_f", "d2_f", "roff", "ent", "a0", "a1", "a1r", "g3", "y3", "scratch"):
            setattr(S, k, st[k].data_ptr())
        S.overflow2, S.done_counter = st["ovf"].data_ptr(), st["done"].data_ptr()
        S.n, S.pitch_f, S.pitch_b, S.use_window, S.max_wg = n, pitch_f, pitch_b, int(self.use_window), max_wg
        S.radius, S.extent, S.dt, S.scale = radius, float(self.filter_extent), float(self.time_step), 1.0 / 128
        for d in range(6):
            S.bbox[d] = float(bbox[d])
        st["S"], st["Sref"] = S, ctypes.byref(S)
        st["nns"] = _PitchedNeighbors(st["idx_f"], st["d2_f"], st["counts2"][:n], pitch_f)
        self._fused = st
        return st

    def _fused_weights(self, st, dev):
        """Packed filters of conv1..3 (+ their Linear branches) and the pointers of layer 0 / the biases in the step struct;
        re-packed only when a parameter's storage or in-place version changed."""
        lib = _lib.load()
        c0f, c0o, d0 = self.conv0_fluid, self.conv0_obstacle, self.dense0_fluid
        tensors = [c0f.kernel, c0f.bias, c0o.kernel, c0o.bias, d0.weight, d0.bias]
        for conv, dense in zip(self.convs, self.denses):
            tensors += [conv.kernel, conv.bias, dense.weight, dense.bias]
        if self.conv_arith not in ("fp32", "split"):
            raise ValueError("ParticleNet.conv_arith must be 'fp32' or 'split'")
        split = self.conv_arith == "split"
        sig = tuple([t._version for t in tensors] + [t.data_ptr() for t in tensors] + [split])
        if st["wsig"] == sig:
            return
        S = st["S"]
        keep = [t.detach().contiguous().float() for t in tensors]
        S.k_fluid, S.b_fluid, S.k_obst, S.b_obst, S.dense0_w, S.dense0_b = [t.data_ptr() for t in keep[:6]]
        packed = []
        for li, (conv, dense) in enumerate(zip(self.convs, self.denses)):
            k, bc, w, bd = keep[6 + 4 * li:10 + 4 * li]
            cin, cout = k.shape[-2], k.shape[-1]
            setattr(S, f"bc{li + 1}", bc.data_ptr())
            setattr(S, f"bd{li + 1}", bd.data_ptr())
            if li == 2:                 # the 3-channel layer: transform + gather (nf_cconv3_layer), its own packing
                wp = torch.empty(lib.nf_cconv3_packed_floats(), dtype=torch.float32, device=dev)
                check(lib.nf_cconv3_pack(ptr(k), ptr(w), ptr(wp), _lib.stream()), "nf_cconv3_pack")
                packed.append(wp)
                S.wp3 = wp.data_ptr()
                continue
            if split:
                wp = torch.empty(lib.nf_cconv_gf_packed_split_bytes(cin, cout), dtype=torch.uint8, device=dev)
                check(lib.nf_cconv_gf_pack_split(ptr(k), ptr(w), cin, cout, ptr(wp), _lib.stream()), "nf_cconv_gf_pack_split")
            else:
                wp = torch.empty(lib.nf_cconv_gf_packed_floats(cin, cout), dtype=torch.float32, device=dev)
                check(lib.nf_cconv_gf_pack(ptr(k), ptr(w), cin, cout, ptr(wp), _lib.stream()), "nf_cconv_gf_pack")
            packed.append(wp)
            setattr(S, f"wp{li + 1}", wp.data_ptr())
        S.split = int(split)
        st["wsig"], st["packed"], st["keep"] = sig, packed, keep

    def check_capacity(self, wait=False):
        """Kept for callers of round 2's API: the fused step now verifies its row capacities before forward() returns and
        redoes an overflowing step on the exact path, so there is never anything pending."""
        return None

    def _forward_fused(self, pos, vel, box, box_feats):
        return self._fused_finish(self._fused_enqueue(pos, vel, box, box_feats))

    def step_async(self, pos, vel, box, box_feats, stream=None, wait_current=True):
        """One inference step ENQUEUED on `stream` (default: the current one) without waiting for its completion word: returns an
        AsyncStep whose result() does the wait (and the exact-path redo of an overflowing step) and hands (pos, vel, num_neighbors) to the
        caller's current stream.  Round 5: a rollout enqueues the step of frame t + 1 on a side stream while frame t renders — the step
        depends on the previous state only (eval_e2e.py:58-134), and its 0.18 ms then run in the idle tails of the renderer's persistent
        MLP launches instead of in front of the next frame (neurofluid_amd/rollout.py).  ONE step may be pending per module (the fused
        step's scratch and flag words are the module's).  wait_current=False: the inputs are already complete for `stream` (they were produced on
        it, or long ago) — the step then does NOT queue behind what the caller's stream still has in flight, which is the point of a lookahead."""
        if self._async_pending is not None:
            raise RuntimeError("ParticleNet.step_async: the previous AsyncStep has not been consumed (call its result())")
        cur = torch.cuda.current_stream(pos.device)
        stream = stream or cur
        ev_in = None
        if wait_current and stream is not cur:
            ev_in = torch.cuda.Event()
            ev_in.record(cur)                   # the inputs are complete on the caller's stream here
        with torch.no_grad(), torch.cuda.stream(stream):
            if ev_in is not None:
                stream.wait_event(ev_in)
            if self.fused_inference and self._fused_ok(pos, box):
                handle, out = self._fused_enqueue(pos, vel, box, box_feats), None
            else:                               # clouds / settings the fused step does not serve: the ordinary forward, on that stream
                handle, out = None, self.forward(pos, vel, box, box_feats)
            ev = torch.cuda.Event()
            ev.record(stream)
        step = AsyncStep(self, handle, out, stream, ev)
        self._async_pending = step
        return step

    def _fused_enqueue(self, pos, vel, box, box_feats):
        # The host side of a step is on the critical path of a rollout (it must fit behind the ~100 us of GPU work that follow
        # the front kernel): nothing here allocates or converts unless an input really needs it.
        lib = self._lib_cached
        f32 = torch.float32
        if pos.dtype is not f32 or not pos.is_contiguous():
            pos = pos.detach().contiguous().float()
        if vel.dtype is not f32 or not vel.is_contiguous():
            vel = vel.detach().contiguous().float()
        if box.dtype is not f32 or not box.is_contiguous():
            box = box.detach().contiguous().float()
        if box_feats.dtype is not f32 or not box_feats.is_contiguous():
            box_feats = box_feats.detach().contiguous().float()
        n, dev = pos.shape[0], pos.device
        st = self._fused_buffers(n, dev, self._scene_bbox(box))
        self._fused_weights(st, dev)
        S = st["S"]
        # (a box updated IN PLACE — a moving obstacle — keeps its pointer: the versions and the row count are part of the key,
        # as they are of _box_grid's own)
        skey = (box.data_ptr(), box._version, box.shape[0], box_feats.data_ptr(), box_feats._version, self.gravity._version)
        if st["skey"] != skey:                 # scene pointers / gravity in the step struct
            bgrid = self._box_grid(box)
            S.box_grid, S.box_feats = bgrid.ws.data_ptr(), box_feats.data_ptr()
            g = self._gravity_host()
            for d in range(3):
                S.gravity[d] = float(g[d])
            st["skey"], st["scene_refs"] = skey, (bgrid, box, box_feats)
        sid = st["step"] = (st["step"] + 1) & 0x3fffffff or 1
        st["S"].search = {"auto": 0, "grid": 1, "all_pairs": 2}[self.fused_search]
        nn = torch.empty(n, dtype=f32, device=dev)
        pos_c, vel_c = torch.empty_like(pos), torch.empty_like(pos)
        rc = lib.nf_trans_step(st["Sref"], pos.data_ptr(), vel.data_ptr(), nn.data_ptr(), pos_c.data_ptr(), vel_c.data_ptr(),
                               st["flag_dev"], sid, torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            check(rc, "nf_trans_step")
        return st, sid, pos, vel, box, box_feats, nn, pos_c, vel_c

    def _fused_finish(self, handle, _again=False):
        """Second half of a fused step: wait for its completion word, redo it on the exact path if a neighbour row overflowed."""
        st, sid, pos, vel, box, box_feats, nn, pos_c, vel_c = handle
        lib = self._lib_cached
        # The front kernel's last workgroup writes `sid` into a pinned word; the host spins on that word (plain memory reads,
        # no runtime call): the overflow words are final then, and the three convolutions are still to run — the wait costs
        # no GPU time.  (A HIP event recorded between the launches of one batch completes with the batch.)
        flag = st["flag_np"]
        # (round 5: the spin runs in C, nf_host_wait_word — ctypes releases the interpreter lock for the call; a Python loop held it)
        if lib.nf_host_wait_word(flag.ctypes.data + 8, sid, 2.0) != 0:      # 2 s without the word: surface whatever went wrong on the device
            torch.cuda.synchronize()
            if flag[2] != sid:
                raise RuntimeError("nf_trans_step: the front kernel never reported completion")
        if flag[0] or flag[1]:
            return self._fused_overflow(st, pos, vel, box, box_feats, _again)
        self.num_fluid_neighbors = nn
        self._y3 = st["y3"]
        nns = st["nns"]
        nns._csr = None                        # lazily rebuilt view of THIS step's rows
        self.conv0_fluid.nns = nns
        return pos_c, vel_c, nn

    def _fused_overflow(self, st, pos, vel, box, box_feats, _again=False):
        """A particle had more neighbours than its row pitch: the step is redone before its results are handed out, and the pitch
        grows for the following steps (up to what the front kernel stages).  Round 6: when the grown pitch holds the step's longest
        rows, the redo is the FUSED step again — its sums run over a row's entries in pair order, whatever the pitch — so a rollout's
        bits no longer depend on the pitch history (tests: a model that starts at pitch 8 / 4 equals one that starts at 64 / 64, bit for
        bit).  A row longer than the front kernel stages at all (or `fused_grow_pitch = False`) is redone on the exact CSR path — the
        reference's uncapped search; same neighbour sets and counts, another summation order (both within 2e-7 of the oracle per step)
        — and the exact path serves the next steps while the clump lasts."""
        of, ob = st["ovf"].tolist()            # (syncs: the exact maxima, for the pitch growth)
        st["ovf"].zero_()
        st["flag_np"][:2] = 0
        self.fused_overflows = getattr(self, "fused_overflows", 0) + 1
        cap = self._fused_limits[2]
        if self.fused_grow_pitch:
            if of:
                self.max_fluid_neighbors = min(cap, max(int(self.max_fluid_neighbors), of + of // 4 + 8))
            if ob:
                self.max_box_neighbors = min(cap, max(int(self.max_box_neighbors), ob + ob // 4 + 8))
        if of > cap or ob > cap:
            self._fused_skip = 16           # a clump denser than the front kernel stages: exact path for a while
        elif self.fused_grow_pitch and of <= int(self.max_fluid_neighbors) and ob <= int(self.max_box_neighbors) and not _again:
            handle = self._fused_enqueue(pos, vel, box, box_feats)          # (new buffers for the new pitch; the same stream)
            return self._fused_finish(handle, _again=True)
        return self._forward_impl(pos, vel, box, box_feats)[:3]

    def _forward_impl(self, pos, vel, box, box_feats, keep=False, other=None, _exact=False, _capture=None):
        """The exact multi-launch path (CSR neighbour lists sized by one host round trip): training (keep=True saves what the
        backward needs), clouds beyond the fused step's limits, and the redo of a fused step whose row pitch overflowed."""
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
        # Pair arrays are sized by the count.  Exact sizing reads it back (a host round trip in the middle of the step, with
        # the GPU idle behind it: ~0.2 ms of a 4.4 ms end-to-end training step).  With capacities learnt from earlier calls of
        # this cloud size the step runs WITHOUT the round trip: the true totals travel to pinned memory behind the count
        # kernels, the row splits are clamped to the capacities (nf_csr_clamp: every consumer stays inside the arrays), and
        # the totals are compared with the capacities when the whole step is enqueued — they arrived long before; on
        # overflow the capacities grow and THIS step is redone with exact sizes (same results as sizing exactly at once).
        caps = self.__dict__.setdefault("_pair_caps", {})
        cap = caps.get(n) if (self.optimistic_pair_capacity and not _exact) else None
        fetch = None
        if cap is not None:
            nnz_f, nnz_b = cap
            tot = torch.empty(2, dtype=torch.int32, device=pos.device)
            f_idx, f_d2 = ops.radius_fill(fgrid, pos_new, radius, f_rs, nnz_f, True)
            b_idx, b_d2 = ops.radius_fill(bgrid, pos_new, radius, b_rs, nnz_b, True)
            check(lib.nf_csr_clamp(ptr(f_rs), n, nnz_f, tot.data_ptr(), st), "nf_csr_clamp")
            check(lib.nf_csr_clamp(ptr(b_rs), n, nnz_b, tot.data_ptr() + 4, st), "nf_csr_clamp")
            if _capture is not None:        # under graph capture: the totals go to the caller's pinned words (a copy node), the
                _capture["tot_dev"] = tot   # caller compares them after the replay's event — or on the device (e2e_graph: nf_note_overflow4)
                if _capture.get("tot_pinned") is not None:
                    _capture["tot_pinned"].copy_(tot, non_blocking=True)
            else:
                fetch = ops.HostFetch(pos.device)
                fetch.add(tot)
        else:
            nnz_f, nnz_b = torch.stack([f_rs[-1], b_rs[-1]]).tolist()
            caps[n] = (ops.round_pairs(nnz_f + nnz_f // 8 + 4096), ops.round_pairs(nnz_b + nnz_b // 4 + 4096))
            f_idx, f_d2 = ops.radius_fill(fgrid, pos_new, radius, f_rs, nnz_f, True)
            b_idx, b_d2 = ops.radius_fill(bgrid, pos_new, radius, b_rs, nnz_b, True)
        f_pw, f_pc = cconv_pairs(pos_new, pos_new, f_rs, f_idx, f_d2, extent, self.use_window)
        b_pw, b_pc = cconv_pairs(box, pos_new, b_rs, b_idx, b_d2, extent, self.use_window)
        # layer 0: [obstacle | fluid | dense] -> (n, 96)   (:116-120)
        a0 = torch.empty(n, 96, dtype=torch.float32, device=pos.device)
        c0o, c0f, d0 = self.conv0_obstacle, self.conv0_fluid, self.dense0_fluid
        check(lib.nf_cconv_small(ptr(box_feats), 3, ptr(b_rs), ptr(b_idx), ptr(b_pw), ptr(b_pc), ptr(c0o.kernel.detach()),
                                 ptr(c0o.bias.detach()), n, ptr(a0), 96, 0, None, None, None, 0, st), "conv0_obstacle")
        if other is None:
            check(lib.nf_cconv_small(ptr(fluid_feats), 4, ptr(f_rs), ptr(f_idx), ptr(f_pw), ptr(f_pc), ptr(c0f.kernel.detach()),
                                     ptr(c0f.bias.detach()), n, ptr(a0), 96, 32, ptr(fluid_feats), ptr(d0.weight.detach()),
                                     ptr(d0.bias.detach()), 64, st), "conv0_fluid")
        else:       # 4 + F input channels: the general layer kernels (transform GEMM + gather) and a plain GEMM for dense0
            fluid_feats = torch.cat([fluid_feats, other.detach().float().to(fluid_feats.device)], 1).contiguous()
            zw = torch.zeros(32, fluid_feats.shape[1], dtype=torch.float32, device=pos.device)
            zb = torch.zeros(32, dtype=torch.float32, device=pos.device)
            a0[:, 32:64] = cconv_layer(fluid_feats, c0f.kernel, c0f.bias, zw, zb, f_rs, f_idx, f_pw, f_pc, relu=False)
            a0[:, 64:96] = ops.gemm(fluid_feats, d0.weight.detach().t()) + d0.bias.detach()
        ans = [a0]
        for conv, dense in zip(self.convs, self.denses):
            prev = ans[-1]
            res = prev if dense.out_features == prev.shape[-1] else None
            ans.append(cconv_layer(prev, conv.kernel, conv.bias, dense.weight, dense.bias, f_rs, f_idx, f_pw, f_pc,
                                   relu=True, residual=res))
        self.num_fluid_neighbors = (f_rs[1:] - f_rs[:-1]).to(torch.float32)   # reduce_subarrays_sum(ones) (:135-138)
        self._y3 = ans[-1]                       # pos_correction (models/transmodel.py:147) is derived on access
        pos_c, vel_c = self.update_pos_vel(pos, pos_new, ans[-1])
        if fetch is not None:
            got_f, got_b = fetch.get()          # (the count kernels finished long ago: no stall, no GPU bubble)
            if got_f > nnz_f or got_b > nnz_b:
                caps[n] = (max(nnz_f, ops.round_pairs(got_f + got_f // 8 + 4096)), max(nnz_b, ops.round_pairs(got_b + got_b // 4 + 4096)))
                self.pair_capacity_redos = getattr(self, "pair_capacity_redos", 0) + 1
                return self._forward_impl(pos, vel, box, box_feats, keep=keep, other=other, _exact=True)
            nnz_f, nnz_b = got_f, got_b
        if _capture is not None:
            self.conv0_fluid.nns = _LazyCSR(f_idx, f_rs, f_d2, _capture["total_fluid"])
        else:
            self.conv0_fluid.nns = SimpleNamespace(neighbors_index=f_idx[:nnz_f], neighbors_row_splits=f_rs,
                                                   neighbors_distance=f_d2[:nnz_f])
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
