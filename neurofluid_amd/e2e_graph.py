"""The end-to-end optimiser step (/root/reference/trainer/trainer_e2e.py:189-302: transition forward -> render of the predicted particles
from the step's views -> rgb + boundary loss -> backward through BOTH models -> Adam) captured ONCE as a HIP graph and replayed.

The eager step is a chain of ~125 launches with two host round trips (the transition step's pair totals, the render passes' row counts)
driven by Python and the autograd engine: GPU-busy 0.93 on a quiet host, and every host stall lands on the GPU's critical path (blocks of
20 steps: 2.73-3.16 ms on one box).  Replayed, the host uploads the step's pixel selection, the optimisers' scalars and a 3-pointer table
per view (the frame the step draws from changes every step: the gather reads its views' addresses from device memory,
nf_gather_view_pixels_tab) and launches one graph.

What makes the step capturable — the same devices as train_step.GraphedRendererStep, plus the carried state:
  * neighbour-pair arrays and render rows run against learnt CAPACITIES, the true counts stay on the device; nf_note_overflow4 compares
    all four with their capacities and sets a sticky poison word that turns this and every following optimiser launch into a no-op;
    the host reads each step's record one step late, raises the capacities, recaptures and REDOES the poisoned steps;
  * truncated BPTT of length 1 (trainer_e2e.py:196-198): the predicted state is copied into the graph's own input buffers at the end of
    the step; the state every step STARTED from is kept in a small ring, so a redone step starts where the first attempt did;
  * the two Adams (or the two parameter groups of one) read their scalars from device memory (nf_adam_step_dev).
Same kernels on the same operands in the same order as the eager step: losses, states and parameters agree bit for bit
(tests/test_gpu_trainers.py::test_graph_replayed_e2e_step_equals_eager).  Single process only: a data-parallel run keeps the eager step
(its dL/dpos and gradient all-reduces sit between the launches)."""
import ctypes

import torch

from . import _lib
from .train_step import HipAdam, ops_round_rows


class GraphedE2EStep:
    def __init__(self, trainer, H, W):
        tr = trainer
        self.tr, self.net, self.pn = tr, tr.renderer, tr.transition_model
        self.dev = tr.device
        self.H, self.W = int(H), int(W)
        self.V, self.rc = len(tr.train_view_names), int(tr.options.RENDERER.ray.ray_chunk)
        # (optimizer, group index) of every parameter group, renderer first: one pair of scalars each
        self.groups = [(tr.optimizer, gi) for gi in range(len(tr.optimizer.param_groups))]
        if tr.separate:
            self.groups += [(tr.transition_optimizer, gi) for gi in range(len(tr.transition_optimizer.param_groups))]
        G, n = len(self.groups), self.V * self.rc
        self.flat_dev = torch.zeros(n, dtype=torch.int64, device=self.dev)
        self.sched_dev = torch.zeros(2 * G, dtype=torch.float32, device=self.dev)
        self.table_dev = torch.zeros(3 * self.V, dtype=torch.int64, device=self.dev)
        self.state_dev = torch.zeros(8, dtype=torch.int32, device=self.dev)
        self.ring_host = torch.zeros(64, dtype=torch.int32).pin_memory()
        self._ring_np = self.ring_host.numpy()
        self._ring_dev = _lib.load().nf_pinned_device_ptr(self.ring_host.data_ptr())
        if not self._ring_dev:
            raise RuntimeError("pinned host memory is not mapped into the device address space (nf_pinned_device_ptr)")
        self._stage = [[torch.empty(n, dtype=torch.int64).pin_memory(), torch.empty(2 * G, dtype=torch.float32).pin_memory(),
                        torch.empty(3 * self.V, dtype=torch.int64).pin_memory(), None] for _ in range(4)]
        self._si = 0
        self.graph = None
        self.pos_s = self.vel_s = None
        self._snap = None                   # ring of 4 (pos, vel): the state each launched step started from
        self._launched = []                 # records of steps not yet checked, oldest first
        self._steps = 0                     # replays since the last capture = the device counter
        self._have_state = False            # pos_s / vel_s hold the carried state (False after an eager step or a fresh capture)
        self._recapture = False
        self._keep = None
        self.captures = self.redone_steps = 0
        self.rows_total = self.steps_total = 0
        self._box = None

    # ------------------------------------------------------------------
    @staticmethod
    def eligible(tr):
        o = tr.options
        opts = [tr.optimizer] + ([tr.transition_optimizer] if tr.separate else [])
        std = type(tr.rgb_criterion) is torch.nn.MSELoss and tr.rgb_criterion.reduction == 'mean' and \
            type(tr.L1_criterion) is torch.nn.L1Loss and tr.L1_criterion.reduction == 'mean'
        return bool(tr.world == 1 and std and o.TRAIN.grad_clip_value == 0 and all(isinstance(op, HipAdam) for op in opts) and
                    getattr(tr.renderer, "mlp_dtype", "fp32") == "fp32" and tr.transition_model.optimistic_pair_capacity and
                    all(p.requires_grad for m in (tr.renderer, tr.transition_model) for p in m.parameters()))

    def _cap_keys(self):
        R = self.V * self.rc
        net = self.net
        return [(R, net.N_samples)] + ([(R, net.N_samples + net.N_importance)] if net.N_importance > 0 else [])

    def ready(self, data):
        """True once eager steps of this shape have learnt every capacity the capture needs and every parameter has a gradient."""
        n = data['particles_pos'].shape[0]
        box, bn = data['box'], data['box_normals']
        ok = self.pn.__dict__.get("_pair_caps", {}).get(n) is not None and all(k in self.net.train_row_cap for k in self._cap_keys())
        ok = ok and all(p.grad is not None for op, gi in self.groups for p in op.param_groups[gi]["params"])
        return bool(ok and box.is_contiguous() and bn.is_contiguous() and box.dtype == torch.float32 and bn.dtype == torch.float32)

    def invalidate_state(self):
        """The trainer ran a step outside the graph: the carried state is tr.pos_for_next_step / vel_for_next_step again."""
        self._have_state = False

    # ------------------------------------------------------------------
    def _key(self, data):
        from .autograd_bwd import _tg_key
        pn, box, bn = self.pn, data['box'], data['box_normals']
        n = data['particles_pos'].shape[0]
        return _tg_key(pn, n, box, bn, pn._box_grid(box), pn._scene_bbox(box)) + \
            tuple(self.net.train_row_cap.get(k) for k in self._cap_keys()) + \
            tuple(p.data_ptr() for p in self.net.parameters()) + (data['rgb_1'][0].reshape(self.H * self.W, -1).shape[1],)

    def _body(self):
        """One step's launches (runs under capture)."""
        from .autograd import _run_passes
        from .autograd_bwd import render_backward, _nerf_params, _trans_backward, _pn_params
        lib = _lib.load()
        tr, net, pn = self.tr, self.net, self.pn
        V, rc, H, W, C = self.V, self.rc, self.H, self.W, self._C
        box, bn = self._box
        from .train_step import PREPACK_IN_CAPTURE
        if PREPACK_IN_CAPTURE:          # the renderer's weight blobs, packed on a side stream while the transition forward runs
            from .autograd_bwd import prepack_for_capture
            prepack_for_capture(net, self.dev)
        cap_state = {"tot_pinned": None, "total_fluid": lambda: self._total_fluid()}
        pos_c, vel_c, nn, aux = pn._forward_impl(self.pos_s, self.vel_s, box, bn, keep=True, _capture=cap_state)
        tot = cap_state["tot_dev"]
        rays = torch.empty(V * rc, 6, device=self.dev)
        rgbs = torch.empty(V * rc, C, device=self.dev)
        ro = torch.empty(V * rc, 3, device=self.dev)
        _lib.check(lib.nf_gather_view_pixels_tab(V, self.table_dev.data_ptr(), rc, C, H * W, self.flat_dev.data_ptr(), rays.data_ptr(),
                                                 rgbs.data_ptr(), ro.data_ptr(), _lib.stream()), "nf_gather_view_pixels_tab")
        fine = net.N_importance > 0
        cap = {"counts": [], "caps": []}
        net.invalidate_grid()               # the particle grid is rebuilt INSIDE the graph (the cloud moves every step) ...
        hint = net._bbox_hint
        net._bbox_hint = tuple(pn._scene_bbox(box))         # ... over the scene's static bounds (any bbox gives the same bits)
        net._capture = cap
        try:
            p0, p1, rays_c, ro_c, grid = _run_passes(net, pos_c, ro, rays, True, fine, save_acts=True)
        finally:
            net._capture = None
            net._bbox_hint = hint           # (the eager callers' hint — the previous cloud's tight bounds — is theirs again)
        if getattr(net, "_prepacked", None):
            net._prepacked["join"]()        # (idempotent: normally done in front of the coarse MLP launch; a forked stream must be joined inside the capture)
        wb = float(tr.options.TRAIN.loss_weight['boundary_loss'])
        use_pos = wb != 0.0
        loss = torch.empty(1, dtype=torch.float32, device=self.dev)
        g0 = torch.empty_like(p0.rgb)
        g1 = torch.empty_like(p1.rgb) if fine else None
        g_pos = torch.empty_like(pos_c) if use_pos else None
        f3 = ctypes.c_float * 3
        lo = (tr.x_bound[1], tr.y_bound[1], tr.z_bound[1]) if use_pos else (0.0, 0.0, 0.0)        # (the argument order of E2ETrainer.train_step)
        hi = (tr.x_bound[0], tr.y_bound[0], tr.z_bound[0]) if use_pos else (0.0, 0.0, 0.0)
        _lib.check(lib.nf_e2e_loss(p0.rgb.data_ptr(), p1.rgb.data_ptr() if fine else None, rgbs.data_ptr(), p0.rgb.numel(), rgbs.numel() // V,
                                   pos_c.data_ptr() if use_pos else None, pos_c.shape[0] if use_pos else 0, f3(*[float(v) for v in lo]),
                                   f3(*[float(v) for v in hi]), wb if use_pos else 0.0, loss.data_ptr(), g0.data_ptr(),
                                   g1.data_ptr() if fine else None, g_pos.data_ptr() if use_pos else None, _lib.stream()), "nf_e2e_loss")
        dpart = torch.zeros_like(grid.points)
        gc, gf = render_backward(net, p0, p1, rays_c, g0, g1, True, particles=grid.points, ro_c=ro_c, dparticles=dpart)
        g_tot = (g_pos + dpart) if use_pos else dpart       # autograd's accumulation order: the loss node's term first
        tgrads = _trans_backward(pn, aux, bn, (False, False, False), g_tot, None)[3]
        rgrads = list(gc) + (list(gf) if fine else [])
        for p_, g_ in zip(_nerf_params(net)[:len(rgrads)], rgrads):
            p_.grad = g_
        for p_, g_ in zip(_pn_params(pn), tgrads):
            p_.grad = g_
        c, k = cap["counts"], cap["caps"]
        pc = pn._pair_caps[pos_c.shape[0]]
        P4, I4 = ctypes.c_void_p * 4, ctypes.c_int32 * 4
        counts = P4(c[0].data_ptr(), c[1].data_ptr() if len(c) > 1 else None, tot.data_ptr(), tot.data_ptr() + 4)
        caps4 = I4(int(k[0]), int(k[1]) if len(k) > 1 else 0, int(pc[0]), int(pc[1]))
        _lib.check(lib.nf_note_overflow4(counts, caps4, self.state_dev.data_ptr(), self._ring_dev, _lib.stream()), "nf_note_overflow4")
        # truncated BPTT of length 1: the next step starts from this step's (detached) prediction
        self.pos_s.copy_(pos_c)
        self.vel_s.copy_(vel_c)
        for g, (op, gi) in enumerate(self.groups):
            op.graph_enqueue(self.sched_dev[2 * g:2 * g + 2], self.state_dev[0:1], gi)
        out = {"rgb0": p0.rgb}
        if fine:
            out["rgb1"] = p1.rgb
        # (pos_c — the step's prediction — is a buffer of the graph's own pool: valid until the next replay; pred_pos() hands it out)
        self._keep = (out, rgbs, pos_c, nn, c, tot, p0, p1, aux, rgrads, tgrads, g_tot, getattr(net, "_prepacked", None))
        net._prepacked = None
        self.caps = [int(v) for v in k] + [int(pc[0]), int(pc[1])]
        return loss[0]

    def _total_fluid(self):
        self._settle(0)
        return int(self._last_counts[2]) if getattr(self, "_last_counts", None) else 0

    def _capture_graph(self, data):
        from . import ops
        self.graph = None
        self._keep = None
        torch.cuda.synchronize(self.dev)
        pn = self.pn
        box, bn = data['box'], data['box_normals']
        pn._scene_bbox(box); pn._box_grid(box)                 # caches filled OUTSIDE the capture (they may sync)
        n = data['particles_pos'].shape[0]
        keep_state = None
        if self.pos_s is not None and self.pos_s.shape[0] == n and self._have_state:
            keep_state = (self.pos_s.clone(), self.vel_s.clone())
        self.pos_s, self.vel_s = torch.zeros(n, 3, device=self.dev), torch.zeros(n, 3, device=self.dev)
        self._snap = [(torch.zeros(n, 3, device=self.dev), torch.zeros(n, 3, device=self.dev)) for _ in range(4)]
        # warm values for the capture run (its kernels do execute nothing, but sizes and pointers are read)
        src = keep_state if keep_state is not None else (data['particles_pos'], data['particles_vel'])
        self.pos_s.copy_(src[0]); self.vel_s.copy_(src[1])
        self._box = (box, bn)
        self._C = data['rgb_1'][0].reshape(self.H * self.W, -1).shape[1]
        self.state_dev.zero_()
        self._ring_np[:] = 0
        self._steps = 0
        self._recapture = False
        for op, _ in self.groups:
            op.zero_grad(set_to_none=True)
        prof, ops.PROFILE = ops.PROFILE, None
        torch.cuda.synchronize(self.dev)
        g = torch.cuda.CUDAGraph()
        try:
            with torch.no_grad(), torch.cuda.graph(g):
                self.loss_static = self._body()
        finally:
            ops.PROFILE = prof
            self.net._prepacked = None
        torch.cuda.synchronize(self.dev)
        # (the capture does not execute: the state buffers still hold what was written above)
        self.graph = g
        self._graph_key = self._key(data)
        self.captures += 1

    # ------------------------------------------------------------------
    def _launch(self, rec):
        """rec = dict(flat, scal, table, reset, refs, loss): enqueue one replay (and what it needs in front of it)."""
        slot = self._stage[self._si % len(self._stage)]
        self._si += 1
        if slot[3] is not None:
            slot[3].synchronize()
        slot[0].copy_(rec["flat"])
        for g, (ss, bc) in enumerate(rec["scal"]):
            slot[1][2 * g], slot[1][2 * g + 1] = ss, bc
        slot[2].copy_(rec["table"])
        self.flat_dev.copy_(slot[0], non_blocking=True)
        self.sched_dev.copy_(slot[1], non_blocking=True)
        self.table_dev.copy_(slot[2], non_blocking=True)
        slot[3] = torch.cuda.Event()
        slot[3].record()
        if rec["reset"] is not None:                  # frame 0 of a sequence (or a state handed over by an eager step)
            self.pos_s.copy_(rec["reset"][0]); self.vel_s.copy_(rec["reset"][1])
        sp, sv = self._snap[self._steps & 3]
        sp.copy_(self.pos_s); sv.copy_(self.vel_s)
        self.graph.replay()
        if rec["loss"] is None:
            rec["loss"] = self.loss_static.clone()
        else:
            rec["loss"].copy_(self.loss_static)
        ev = torch.cuda.Event()
        ev.record()
        rec["idx"], rec["ev"] = self._steps, ev
        self._launched.append(rec)
        self._steps += 1
        self._have_state = True
        return rec["loss"]

    def _settle(self, keep_in_flight):
        redo_loss = None
        while len(self._launched) > keep_in_flight:
            rec = self._launched[0]
            idx = rec["idx"]
            rec["ev"].synchronize()
            r = self._ring_np[(idx & 7) * 8:(idx & 7) * 8 + 8]
            if int(r[2]) != idx + 1:
                raise RuntimeError("GraphedE2EStep: step %d finished without its overflow record (found counter %d)" % (idx, int(r[2])))
            poisoned, counts = int(r[0]), [int(r[3]), int(r[4]), int(r[5]), int(r[6])]
            rcaps, pn = self.net.train_row_cap, self.pn
            n = self.pos_s.shape[0]
            if poisoned:
                todo = list(self._launched)
                torch.cuda.synchronize(self.dev)
                for key, cnt, cap in zip(self._cap_keys(), counts[:2], self.caps[:2]):
                    if cnt > cap:
                        rcaps[key] = max(rcaps.get(key, 0), ops_round_rows(cnt + cnt // 2 + 4096))
                from . import ops
                pf, pb = pn._pair_caps[n]
                if counts[2] > pf or counts[3] > pb:
                    pn._pair_caps[n] = (max(pf, ops.round_pairs(counts[2] + counts[2] // 8 + 4096)),
                                        max(pb, ops.round_pairs(counts[3] + counts[3] // 4 + 4096)))
                    pn.pair_capacity_redos = getattr(pn, "pair_capacity_redos", 0) + 1
                # the state the FIRST poisoned step started from (its snapshot), then the steps again, in order
                start = tuple(t.clone() for t in self._snap[idx & 3])
                self._launched = []
                self._have_state = False
                self._capture_graph(rec["data"])
                first = True
                for t in todo:
                    if first and t["reset"] is None:
                        t["reset"] = start
                    first = False
                    redo_loss = self._launch(t)
                    self.redone_steps += 1
                continue
            self._launched.pop(0)
            self._last_counts = counts
            self.rows_total += counts[0] + counts[1]
            self.steps_total += 1
            for key, cnt, cap in zip(self._cap_keys(), counts[:2], self.caps[:2]):
                if cnt > cap * 0.9 and rcaps.get(key, 0) <= cap:
                    rcaps[key] = ops_round_rows(cnt + cnt // 4 + 4096)
                    self._recapture = True
        return redo_loss

    def step(self, data, data_idx, coords, sels):
        """One optimiser step on frame `data` (E2ETrainer's device-resident item).  coords (n, 2) host pixel grid, sels[v] the selected
        rows of view v.  Returns the step's loss (a 0-dim tensor; corrected in place if the step has to be redone)."""
        tr = self.tr
        yx = torch.cat([coords[torch.as_tensor(s)] for s in sels]).long()
        flat = yx[:, 0] * self.W + yx[:, 1]
        if flat.numel() != self.V * self.rc:
            raise ValueError("GraphedE2EStep: %d pixels per step expected" % (self.V * self.rc))
        if int(flat.min()) < 0 or int(flat.max()) >= self.H * self.W:
            raise IndexError("pixel selection outside the %d x %d image" % (self.H, self.W))
        # the carried state: frame 0 restarts the sequence; after an eager step (or a fresh capture) it comes from the trainer
        reset = None
        if data_idx == 0:
            reset = (data['particles_pos'], data['particles_vel'])
        elif not self._have_state:
            reset = (tr.pos_for_next_step, tr.vel_for_next_step)
        if self.graph is None or self._recapture or self._graph_key != self._key(data):
            redo = self._settle(0)
            if self.graph is None or self._graph_key != self._key(data) or (self._recapture and redo is None):
                if reset is None:                       # the capture replaces the state buffers: carry their contents over
                    reset = (self.pos_s.clone(), self.vel_s.clone())
                self._capture_graph(data)
        scal = [op.graph_scalars(gi) for op, gi in self.groups]
        if any(s is None for s in scal):
            raise RuntimeError("GraphedE2EStep: an optimizer cannot step from a graph (see HipAdam.graph_scalars)")
        HW = self.H * self.W
        refs = [(data['rays_1'][v].reshape(HW, -1).contiguous(), data['rgb_1'][v].reshape(HW, -1).contiguous(), data['cw_1'][v].contiguous())
                for v in range(self.V)]
        if any(t.dtype is not torch.float32 for r in refs for t in r) or any(r[0].shape[1] != 6 or r[1].shape[1] != self._C for r in refs):
            raise ValueError("GraphedE2EStep: fp32 rays (H*W, 6) and colours (H*W, %d) expected" % self._C)
        table = torch.tensor([t.data_ptr() for r in refs for t in r], dtype=torch.int64)
        rec = dict(flat=flat, scal=scal, table=table, reset=reset, refs=refs, loss=None, data=data)
        loss = self._launch(rec)
        redo = self._settle(1)
        # what the trainer's own attributes say about the carried state: the graph's input buffers hold it (valid until the next replay)
        tr.pos_for_next_step, tr.vel_for_next_step = self.pos_s, self.vel_s
        return redo if redo is not None else loss

    def verify(self):
        """Settle every enqueued step (redoing poisoned ones: their loss tensors are corrected in place)."""
        self._settle(0)

    def pred_pos(self):
        """The LAST replayed step's predicted positions (static; valid until the next step) — after verify() for a value that is logged."""
        return self._keep[2]

    def last_outputs(self):
        return self._keep[0], self._keep[1]
