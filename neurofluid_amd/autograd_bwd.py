"""Autograd of the fused renderer and of the transition model (row B8).

Renderer backward = composite backward (HIP) -> MLP data-gradient on fp32 MFMA (HIP, nf_nerf_mlp_bwd) ->
weight gradients dW_l = dpre_l^T . input_l over all active rows (one batched fp32-MFMA launch, nf_nerf_wgrad) ->
optional gradient w.r.t. the particle positions through the local-geometry features (e2e training,
trainer/trainer_e2e.py:219-236).  Every dense product runs on the fp32 matrix pipe in hand-written HIP: the plain
GEMMs of both backward passes (dX = dpre W; dB = relu(x)^T dG, dx = dG B^T of the continuous convolutions) go through
ops.gemm (nf_gemm.hip) — no vendor BLAS on the path.
"""
import ctypes

import torch

from . import _lib, ops
from ._lib import check, ptr
from .autograd import _run_passes, _results, LazyResults

ACT, DPRE = 2432, 2436


def _nerf_params(net):
    ps = []
    for n in (net.nerf_coarse, net.nerf_fine):
        layers = n.linear_layers()
        ps += [l.weight for l in layers] + [l.bias for l in layers]
    return ps


def _pack_bwd(nerf, cx, cd, dev, n_layout=True):
    lib = _lib.load()
    layers = nerf.linear_layers()
    P = _lib.NerfParams()
    keep = []
    for i, l in enumerate(layers):
        w, b = l.weight.detach().contiguous(), l.bias.detach().contiguous()
        keep += [w, b]
        P.w[i], P.b[i] = w.data_ptr(), b.data_ptr()
    out = torch.empty(lib.nf_nerf_packed_bwd_floats(), dtype=torch.float32, device=dev)
    check(lib.nf_nerf_pack_bwd(ctypes.byref(P), cx, cd, ptr(out), _lib.stream()), "nf_nerf_pack_bwd")
    if not n_layout:
        return out
    out_n = torch.empty_like(out)           # nf_nerf_mlp_bwd_n's arrangement: two K-steps of a wave's blocks per 16-B load
    check(lib.nf_nerf_pack_bwd_n(ptr(out), ptr(out_n), _lib.stream()), "nf_nerf_pack_bwd_n")
    return out_n


def _pass_backward(net, nerf, pb, rays_c, z, z_table, g_rgb, white_bg, particles=None, ro_c=None, dparticles=None, wgrad_on=None):
    """Returns the 24 parameter gradients (12 weights, 12 biases) of one NeRF for one render pass; when
    `dparticles` is given also accumulates dL/d(particle positions) into it (e2e training).
    wgrad_on: a stream for the weight-gradient launch (it forks behind the data-gradient kernel; the CALLER joins it) — nothing on the
    way to dL/d particles waits for the weight gradients, so the end-to-end step's transition backward can start while they run."""
    lib = _lib.load()
    st = _lib.stream()
    dev = rays_c.device
    cx, cd = net.in_channels_xyz, net.in_channels_dir
    layers = nerf.linear_layers()
    n = pb.n_active
    if n == 0:
        return [torch.zeros_like(l.weight) for l in layers] + [torch.zeros_like(l.bias) for l in layers]
    R, S = pb.R, pb.S
    scratch = torch.empty(R * S, dtype=torch.float32, device=dev)
    d_rs = torch.empty(R * S, 4, dtype=torch.float32, device=dev)
    g = g_rgb.detach().contiguous().float()
    if getattr(pb, "noise", None) is not None:
        check(lib.nf_composite_bwd_noise(ptr(pb.rgbsigma), ptr(z), ptr(z_table), ptr(rays_c), ptr(g), None, pb.gate, R, S,
                                         int(white_bg), ptr(pb.noise), ptr(scratch), ptr(d_rs), ptr(pb.num_nn), pb.K, st),
              "nf_composite_bwd_noise")
    else:
        check(lib.nf_composite_bwd(ptr(pb.rgbsigma), ptr(z), ptr(z_table), ptr(rays_c), ptr(g), None, pb.gate, R, S,
                                   int(white_bg), ptr(scratch), ptr(d_rs), ptr(pb.num_nn), pb.K, st), "nf_composite_bwd")
    pre = (getattr(net, "_prepacked", None) or {}).get(id(nerf))
    packed_t = pre["bwd_n"] if pre else _pack_bwd(nerf, cx, cd, dev)
    # every row-sized temporary is allocated at the pass's bucketed capacity (ops._round_rows): the active-row count
    # changes from step to step, and exact sizes make the caching allocator grow by a fresh block per step (and give
    # the GEMM library a new problem size per step); a handful of capacity buckets are re-used instead
    cap = ops._round_rows(n)
    dpre_full = torch.empty(cap, DPRE, dtype=torch.float32, device=dev)
    dpre = dpre_full[:n]
    dX = None
    if getattr(pb, "amask", None) is not None and dparticles is not None:
        # end-to-end step: dL/dX comes out of the same launch (three more K loops over images the kernel holds anyway)
        dX = torch.empty(cap, cx + cd, dtype=torch.float32, device=dev)
        check(lib.nf_nerf_mlp_bwd_n3(ptr(pb.packed), ptr(packed_t), cx, cd, ptr(pb.amask), ptr(pb.n_rows), n, ptr(pb.row_sample),
                                     ptr(pb.rgbsigma), ptr(d_rs), ptr(dpre_full), ptr(dX), st), "nf_nerf_mlp_bwd_n3")
    elif getattr(pb, "amask", None) is not None:      # the forward left the ReLU masks as bits: no read of the saved activations here
        check(lib.nf_nerf_mlp_bwd_n2(ptr(pb.packed), ptr(packed_t), cx, cd, ptr(pb.amask), ptr(pb.n_rows), n, ptr(pb.row_sample),
                                     ptr(pb.rgbsigma), ptr(d_rs), ptr(dpre_full), st), "nf_nerf_mlp_bwd_n2")
    else:
        check(lib.nf_nerf_mlp_bwd_n(ptr(pb.packed), ptr(packed_t), cx, cd, ptr(pb.acts), ptr(pb.n_rows), n, ptr(pb.row_sample),
                                    ptr(pb.rgbsigma), ptr(d_rs), ptr(dpre_full), st), "nf_nerf_mlp_bwd_n")
    # weight gradients: one batched fp32-MFMA launch for all 15 GEMMs of the net (nf_nerf_wgrad)
    nsl = 21          # 12 jobs (workgroups of up to four 128 x 128 tiles that share operand blocks) x 21 row slices = 252 workgroups, one per CU

    def launch_wgrad():
        st_ = _lib.stream()
        blob = torch.empty(lib.nf_nerf_wgrad_floats(cx, cd), dtype=torch.float32, device=dev)
        wsp = torch.empty(lib.nf_nerf_wgrad_workspace_floats(cx, cd, nsl), dtype=torch.float32, device=dev)
        colsum = torch.empty(DPRE, dtype=torch.float32, device=dev)
        if getattr(pb, "graph_mode", False):        # a captured step: n is the CAPACITY, the true count is read on the device (same slicing, same sums)
            check(lib.nf_nerf_wgrad_dev(ptr(dpre_full), ptr(pb.acts), ptr(pb.X), cx, cd, ptr(pb.n_rows), n, nsl, ptr(wsp), ptr(blob), ptr(colsum), st_),
                  "nf_nerf_wgrad_dev")
        else:
            check(lib.nf_nerf_wgrad(ptr(dpre_full), ptr(pb.acts), ptr(pb.X), cx, cd, n, nsl, ptr(wsp), ptr(blob), ptr(colsum), st_),
                  "nf_nerf_wgrad")
        return blob, colsum

    if wgrad_on is not None:
        here = torch.cuda.current_stream(dev)
        wgrad_on.wait_stream(here)                  # dpre is complete behind everything enqueued here so far
        with torch.cuda.stream(wgrad_on):
            blob, colsum = launch_wgrad()
        for t_ in (dpre_full, pb.acts, pb.X):       # read by the fork: their blocks must not be handed out again before it is done
            if t_ is not None:
                t_.record_stream(wgrad_on)
        blob.record_stream(here); colsum.record_stream(here)
    else:
        blob, colsum = launch_wgrad()
    gw, o = [], 0
    for l in layers:
        k = l.weight.numel()
        gw.append(blob[o:o + k].view_as(l.weight))
        o += k
    # bias gradients = column sums of dpre, summed by the weight-gradient launch from the slabs it stages
    gb = [colsum[k * 256:(k + 1) * 256] for k in range(8)]
    gb += [colsum[8 * 256:9 * 256], colsum[9 * 256:9 * 256 + 128], colsum[2435:2436], colsum[2432:2435]]
    if dparticles is not None:
        # dL/dX = dpre W for the three layers that read the feature matrix (fp32-MFMA GEMMs on column slices of dpre, the
        # skip layer's term accumulated in place), then HIP scatter
        if dX is None:          # (the tile-per-wave forward of a non-default feature row: no mask words, no fused dX)
            W1, W5, Wd = layers[0].weight.detach(), layers[4].weight.detach(), layers[9].weight.detach()
            dX = torch.empty(cap, cx + cd, dtype=torch.float32, device=dev)
            ops.gemm(dpre[:, 0:256], W1, out=dX[:n, :cx])
            ops.gemm(dpre[:, 4 * 256:5 * 256], W5[:, :cx], out=dX[:n, :cx], accumulate=True)
            ops.gemm(dpre[:, 9 * 256:9 * 256 + 128], Wd[:, 256:], out=dX[:n, cx:])
        check(lib.nf_render_features_bwd(ptr(particles), ptr(rays_c), ptr(z), ptr(z_table), R, S, float(net.raduis),
                                         net.num_neighbor, net.enc_flags, ptr(ro_c), int(ro_c.dim() == 2),
                                         ptr(pb.row_sample), ptr(pb.row_nbr), ptr(pb.n_rows), n, ptr(dX), ptr(dparticles),
                                         st), "nf_render_features_bwd")
    return gw + gb


def prepack_for_capture(net, dev):
    """Every weight blob a captured training step needs — forward blob, its tile-per-workgroup arrangement and the backward blob of both
    NeRFs: 8 small launches that otherwise sit between dependent kernels on the step's critical path (pack -> classify, pack_n -> forward
    MLP, pack_bwd -> pack_n -> backward MLP, per pass) — enqueued on a side stream at the top of the capture, beside the pixel gather and
    the coarse pass's classify / search / feature kernels.  Sets net._prepacked (consulted by autograd._run_passes, ops.render_pass and
    _pass_backward while net._capture is set); "join" makes the calling stream wait for them once, in front of the first MLP launch.
    The caller clears net._prepacked when the capture body is done."""
    lib = _lib.load()
    cur = torch.cuda.current_stream(dev)
    side = _side_stream(dev, 2)
    side.wait_stream(cur)               # the parameters are final on the capturing stream (the previous step's Adam)
    cx, cd = net.in_channels_xyz, net.in_channels_dir
    pre = {}
    with torch.cuda.stream(side):
        for nerf in (net.nerf_coarse, net.nerf_fine):
            pk = net.packed_weights(nerf)
            pk_n = None
            if ((cx + 7) // 8, (cd + 7) // 8) == (25, 7):
                pk_n = torch.empty_like(pk)
                check(lib.nf_nerf_pack_n(ptr(pk), cx, cd, ptr(pk_n), _lib.stream()), "nf_nerf_pack_n")
            bwd_n = _pack_bwd(nerf, cx, cd, dev)
            for t in (pk, pk_n, bwd_n):
                if t is not None:
                    t.record_stream(cur)
            pre[id(nerf)] = {"pk": pk, "pk_n": pk_n, "bwd_n": bwd_n}
    state = {"joined": False}

    def join():
        if not state["joined"]:
            torch.cuda.current_stream(dev).wait_stream(side)
            state["joined"] = True
    pre["join"] = join
    net._prepacked = pre
    return pre


TWO_STREAM_BACKWARD = True
# End-to-end step (particles need a gradient): the two passes' weight-gradient launches run on a third stream that is joined when the
# WHOLE backward is over (an engine callback), so the transition model's backward — which needs dL/d particles only — starts ~0.3 ms
# earlier instead of behind them.  Only when no parameter holds a .grad yet (an accumulation into an existing .grad would be enqueued
# on the backward's own stream, ahead of the join).
WGRAD_SIDE_STREAM = True
_SIDE = {}


def _side_stream(device, which=0):
    key = (device.type, device.index, which)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


class _RenderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, particles, ro, rays, white_bg, fine, *params):
        ctx.set_materialize_grads(False)      # backward reads rgb0 / rgb1 only: no zero tensors for the 8 other outputs
        opts = fine if isinstance(fine, tuple) else (fine, False, 0.0, 0.0)      # (fine, use_disp, noise_std, perturb) ride in one non-tensor argument
        fine, use_disp, noise_std, perturb = opts
        ctx.use_disp = use_disp
        p0, p1, rays_c, ro_c, grid = _run_passes(net, particles, ro, rays, white_bg, fine, save_acts=True, use_disp=use_disp,
                                                 noise_std=noise_std, perturb=perturb)
        ctx.net, ctx.p0, ctx.p1, ctx.rays_c, ctx.white_bg, ctx.fine = net, p0, p1, rays_c, white_bg, fine
        ctx.particles_need_grad = particles.requires_grad
        ctx.ro_c, ctx.pts = ro_c, grid.points
        res = _results(p0, p1)
        keys = ["rgb0", "depth0", "opacity0", "num_nn_0", "mask_0"]
        if fine:
            keys += ["rgb1", "depth1", "opacity1", "num_nn_1", "mask_1"]
        ctx.keys = keys
        # the neighbour counts leave the Function as the kernels' int32 (widened lazily by render_with_grad's dict)
        outs = tuple(res.raw_int32(k)[0].view(res.raw_int32(k)[1]) if res.raw_int32(k) is not None else res[k] for k in keys)
        ctx.mark_non_differentiable(*[o for k, o in zip(keys, outs) if not k.startswith("rgb")])
        # The pass buffers live on ctx for backward.  They must NOT keep the tensors that are returned: an output holds
        # its grad_fn (this ctx) through a C++ edge the Python GC cannot see, so ctx -> buffers -> output -> ctx would
        # be an uncollectable cycle and every training step would leak its activations (~0.4 GB at 1024 rays).
        for pb in (p0, p1):
            if pb is not None:
                pb.rgb = pb.depth = pb.opacity = pb.mask_sum = pb.weights = None
        return outs

    @staticmethod
    def backward(ctx, *grads):
        g = dict(zip(ctx.keys, grads))
        dpart = torch.zeros_like(ctx.pts) if ctx.particles_need_grad else None
        wside = None
        if ctx.particles_need_grad and WGRAD_SIDE_STREAM and all(p.grad is None for p in _nerf_params(ctx.net)):
            wside = _side_stream(ctx.pts.device, 1)
        gc, gf = render_backward(ctx.net, ctx.p0, ctx.p1 if ctx.fine else None, ctx.rays_c, g.get("rgb0"), g.get("rgb1") if ctx.fine else None,
                                 ctx.white_bg, ctx.use_disp, particles=ctx.pts, ro_c=ctx.ro_c, dparticles=dpart, wgrad_on=wside)
        if wside is not None:
            here = torch.cuda.current_stream(ctx.pts.device)
            torch.autograd.Variable._execution_engine.queue_callback(lambda: here.wait_stream(wside))
        return (None, dpart, None, None, None, None) + tuple(gc) + tuple(gf)


def render_backward(net, p0, p1, rays_c, g_rgb0, g_rgb1, white_bg, use_disp=False, particles=None, ro_c=None, dparticles=None, wgrad_on=None):
    """The renderer's backward from the pass buffers of a training forward (_run_passes(save_acts=True)): the 24 parameter gradients of
    each NeRF (None where a pass received no gradient), dL/d particles accumulated into `dparticles` when given.  Used by the autograd
    Function above and, directly, by the captured training step (train_step.GraphedRendererStep: no autograd engine inside the graph)."""
    z_table, _ = net._tables(rays_c.device, use_disp)
    z0 = getattr(p0, "z", None)                  # perturb > 0: the coarse pass ran on per-ray depths
    zt0 = None if z0 is not None else z_table
    extra = dict(particles=particles, ro_c=ro_c, dparticles=dparticles, wgrad_on=wgrad_on)
    fine = p1 is not None
    both = g_rgb0 is not None and fine and g_rgb1 is not None and TWO_STREAM_BACKWARD
    if both:
        # The two passes' backward chains are independent (two networks; the importance samples are detached), and the
        # MLP kernels quantise badly on their own: one wave per 32-row tile for 0.35 ms, 1 024 waves per round, so the
        # coarse pass (~600 tiles) leaves 40 % of the chip idle for a whole round and the fine pass (~2 100 tiles) pays
        # a third round for 2.04 rounds of work.  On two streams the dispatcher fills the CUs from both launches.
        cur = torch.cuda.current_stream(rays_c.device)
        side = _side_stream(rays_c.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            gc = _pass_backward(net, net.nerf_coarse, p0, rays_c, z0, zt0, g_rgb0, white_bg, **extra)
            for t in gc:
                if t is not None:
                    t.record_stream(cur)        # allocated on the side stream, consumed on the current one
        gf = _pass_backward(net, net.nerf_fine, p1, rays_c, p1.z, None, g_rgb1, white_bg, **extra)
        cur.wait_stream(side)
        return gc, gf
    gc = _pass_backward(net, net.nerf_coarse, p0, rays_c, z0, zt0, g_rgb0, white_bg, **extra) if g_rgb0 is not None else [None] * 24
    gf = _pass_backward(net, net.nerf_fine, p1, rays_c, p1.z, None, g_rgb1, white_bg, **extra) if (fine and g_rgb1 is not None) else [None] * 24
    return gc, gf


def render_with_grad(net, particles, ro, rays, white_bg, fine, use_disp=False, noise_std=0.0, perturb=0.0):
    opts = (fine, bool(use_disp), float(noise_std), float(perturb)) if (use_disp or noise_std or perturb) else fine
    outs = _RenderFn.apply(net, particles, ro, rays, white_bg, opts, *_nerf_params(net))
    keys = ["rgb0", "depth0", "opacity0", "num_nn_0", "mask_0"] + (
        ["rgb1", "depth1", "opacity1", "num_nn_1", "mask_1"] if fine else [])
    out = LazyResults()
    for k, v in zip(keys, outs):
        if k.startswith("num_nn") and v.dtype == torch.int32:
            out.set_lazy(k, v, v.shape)
        else:
            out[k] = v
    return out


# ================================================================================================
# transition model (B8)
# ================================================================================================
def _pn_params(pn):
    ps = [pn.conv0_fluid.kernel, pn.conv0_fluid.bias, pn.conv0_obstacle.kernel, pn.conv0_obstacle.bias,
          pn.dense0_fluid.weight, pn.dense0_fluid.bias]
    for conv, dense in zip(pn.convs, pn.denses):
        ps += [conv.kernel, conv.bias, dense.weight, dense.bias]
    return ps


def _virtual_b(kernel, dense_w):
    """[filter as (Cin x 64*Cout) | dense_w^T] — the B operand of nf_cconv_transform, materialised for the GEMMs."""
    cin, cout = kernel.shape[-2], kernel.shape[-1]
    kf = kernel.detach().reshape(64, cin, cout).permute(1, 0, 2).reshape(cin, 64 * cout)
    return torch.cat([kf, dense_w.detach().t()], dim=1)


class _ParticleNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pn, pos, vel, box, box_feats, feats, *params):
        ctx.set_materialize_grads(False)      # g_pos / g_vel arrive as None when unused (backward handles both)
        pos_c, vel_c, nn, aux = pn._forward_impl(pos, vel, box, box_feats, keep=True, other=feats)
        ctx.pn, ctx.aux = pn, aux
        ctx.box, ctx.box_feats = box.detach().contiguous().float(), box_feats.detach().contiguous().float()
        ctx.in_grad = (pos.requires_grad, vel.requires_grad, feats is not None and feats.requires_grad)
        ctx.mark_non_differentiable(nn)
        return pos_c, vel_c, nn

    @staticmethod
    def backward(ctx, g_pos, g_vel, _g_nn):
        g_in_pos, g_in_vel, g_in_feats, grads = _trans_backward(ctx.pn, ctx.aux, ctx.box_feats, ctx.in_grad, g_pos, g_vel)
        return (None, g_in_pos, g_in_vel, None, None, g_in_feats) + grads


def _trans_backward(pn, aux, box_feats_c, in_grad, g_pos, g_vel):
    """Backward of one ParticleNet step (B8): (g_in_pos, g_in_vel, g_in_feats, parameter gradients in _pn_params order)."""
    if True:
        from .transmodel import cconv_pairs
        lib = _lib.load()
        st = _lib.stream()
        dt = float(pn.time_step)
        ans = aux["ans"]
        f_rs, f_idx, f_pw, f_pc = aux["f"]
        b_rs, b_idx, b_pw, b_pc = aux["b"]
        n = ans[0].shape[0]
        dev = ans[0].device
        d_pos_c = torch.zeros(n, 3, device=dev) if g_pos is None else g_pos.detach().float()      # read only below
        if g_vel is not None:
            d_pos_c = d_pos_c + g_vel.detach().float() / dt          # vel_c = (pos_c - pos) / dt
        extent = float(pn.filter_extent)
        f_d2 = aux["f_d2"]          # this CALL's pair distances (the module's .nns is overwritten by every forward)
        # the fluid<->fluid pairs as seen from the neighbour (transposed operator), once for all layers
        t_pw, t_pc = cconv_pairs(aux["pos_new"], aux["pos_new"], f_rs, f_idx, f_d2, extent, pn.use_window, negate=True)
        dy = (d_pos_c * (1.0 / 128)).contiguous()                    # pos_correction = y3 / 128
        grads = {}
        for li in (3, 2, 1):
            conv, dense = pn.convs[li - 1], pn.denses[li - 1]
            prev = ans[li - 1]
            cout = conv.kernel.shape[-1]
            dG = torch.empty(n, 65 * cout, dtype=torch.float32, device=dev)
            check(lib.nf_cconv_gather_bwd(ptr(dy), cout, ptr(f_rs), ptr(f_idx), ptr(t_pw), ptr(t_pc), n, ptr(dG), st),
                  "nf_cconv_gather_bwd")
            dB = ops.gemm(prev.t(), dG, relu_a=True)                 # relu(prev)^T dG: (Cin, 65*Cout), split-K over the particles
            cin = prev.shape[1]
            # round 5: the glue between the launches as HIP kernels (nf_host.hip) instead of ~8 ATen launches per layer
            gK, gW = torch.empty_like(conv.kernel), torch.empty_like(dense.weight)
            check(lib.nf_cconv_split_db(ptr(dB), cin, cout, ptr(gK), ptr(gW), st), "nf_cconv_split_db")
            gb, gb2 = torch.empty(cout, dtype=torch.float32, device=dev), torch.empty(cout, dtype=torch.float32, device=dev)
            check(lib.nf_colsum(ptr(dy), n, cout, cout, ptr(gb), ptr(gb2), st), "nf_colsum")
            grads[conv.kernel], grads[dense.weight], grads[conv.bias], grads[dense.bias] = gK, gW, gb, gb2
            dx = ops.gemm(dG, _virtual_b(conv.kernel, dense.weight).t())      # (n, Cin)
            dprev = torch.empty_like(dx)        # dx where prev > 0, + dy on the residual branch (transmodel.py:127-128)
            check(lib.nf_relu_bwd_add(ptr(dx), ptr(prev), ptr(dy) if dense.out_features == prev.shape[-1] else None, ptr(dprev),
                                      dx.numel(), st), "nf_relu_bwd_add")
            dy = dprev
        # layer 0: dy is d[obstacle(32) | fluid(32) | dense0(32)]
        ff = aux["fluid_feats"]
        c0f, c0o, d0 = pn.conv0_fluid, pn.conv0_obstacle, pn.dense0_fluid
        dKo = torch.zeros_like(c0o.kernel)
        dKf = torch.zeros_like(c0f.kernel)
        wsf = torch.empty(lib.nf_cconv_small_bwd_filter_workspace_floats(4, n), dtype=torch.float32, device=dev)
        check(lib.nf_cconv_small_bwd_filter(ptr(box_feats_c), 3, ptr(b_rs), ptr(b_idx), ptr(b_pw), ptr(b_pc), ptr(dy), 96, 0,
                                            n, ptr(wsf), ptr(dKo), st), "conv0_obstacle filter grad")
        cin0 = ff.shape[1]
        dG0 = None
        if cin0 == 4:
            check(lib.nf_cconv_small_bwd_filter(ptr(ff), 4, ptr(f_rs), ptr(f_idx), ptr(f_pw), ptr(f_pc), ptr(dy), 96, 32, n,
                                                ptr(wsf), ptr(dKf), st), "conv0_fluid filter grad")
        else:       # other_feats_channels > 0: the general layer's backward (transposed gather + GEMM), no activation on the input
            dyf = dy[:, 32:64].contiguous()
            dG0 = torch.empty(n, 65 * 32, dtype=torch.float32, device=dev)
            check(lib.nf_cconv_gather_bwd(ptr(dyf), 32, ptr(f_rs), ptr(f_idx), ptr(t_pw), ptr(t_pc), n, ptr(dG0), st),
                  "nf_cconv_gather_bwd")
            dB0 = ops.gemm(ff.t(), dG0[:, :64 * 32])
            dKf = dB0.reshape(cin0, 64, 32).permute(1, 0, 2).reshape(c0f.kernel.shape).contiguous()
        grads[c0o.kernel], grads[c0f.kernel] = dKo, dKf
        gb0 = torch.empty(96, dtype=torch.float32, device=dev)
        check(lib.nf_colsum(ptr(dy), n, 96, 96, ptr(gb0), None, st), "nf_colsum")
        grads[c0o.bias], grads[c0f.bias] = gb0[:32], gb0[32:64]
        grads[d0.weight], grads[d0.bias] = ops.gemm(dy[:, 64:].t(), ff), gb0[64:]
        # input gradients (2-step unrolls, trainer_transmodel.py): through integrate/update and the velocity features
        g_in_pos = g_in_vel = None
        if in_grad[0]:
            g_in_pos = d_pos_c - (g_vel.detach().float() / dt if g_vel is not None else 0.)
        g_in_feats = None
        if in_grad[1] or in_grad[2]:
            if cin0 == 4:
                dfeat = torch.empty(n, 4, dtype=torch.float32, device=dev)
                check(lib.nf_cconv_small_bwd_feat(ptr(c0f.kernel.detach().contiguous()), 4, ptr(f_rs), ptr(f_idx), ptr(t_pw),
                                                  ptr(t_pc), ptr(dy), 96, 32, n, ptr(dfeat), st), "conv0_fluid feature grad")
            else:
                zw = torch.zeros(32, cin0, dtype=torch.float32, device=dev)
                dfeat = ops.gemm(dG0, _virtual_b(c0f.kernel, zw).t())          # (n, 4 + F)
            ops.gemm(dy[:, 64:], d0.weight.detach(), out=dfeat, accumulate=True)
            if in_grad[1]:
                g_in_vel = d_pos_c * dt + dfeat[:, 1:4]             # pos_new = pos + vel dt + g dt^2/2 ; feats = [1, vel + g dt]
            if in_grad[2]:
                g_in_feats = dfeat[:, 4:].contiguous()
        return g_in_pos, g_in_vel, g_in_feats, tuple(grads[p] for p in _pn_params(pn))



def particle_net_with_grad(pn, pos, vel, box, box_feats, feats=None):
    return _ParticleNetFn.apply(pn, pos, vel, box, box_feats, feats, *_pn_params(pn))


# ================================================================================================
# Graph-replayed training step of the transition model (ParticleNet.training_graph; E2ETrainer sets it)
# ================================================================================================
class _TransGraphs:
    """The captured forward / backward launch sequences of one ParticleNet for one (cloud size, pair capacities, scene,
    parameter storage) key, with their static input / output buffers."""

    def __init__(self):
        self.key = None
        self.serial = 0


def _tg_key(pn, n, box, box_feats, bgrid, bbox):
    """Everything a captured launch sequence has baked in: sizes, capacities, every storage pointer it reads — the container
    grid's WORKSPACE included: ParticleNet._box_cache holds one entry, a call on another container replaces it, and a graph
    captured against the old workspace would be replayed on freed memory when training resumes with the first container —
    and the host scalars that travel as kernel arguments (time step, filter extent, the scene box)."""
    caps = pn.__dict__.get("_pair_caps", {}).get(n)
    return (n, caps, box.data_ptr(), box._version, box.shape[0], box_feats.data_ptr(), box_feats._version, bool(pn.use_window),
            tuple(p.data_ptr() for p in _pn_params(pn)), pn.gravity._version, bgrid.ws.data_ptr(),
            float(pn.time_step), float(pn.filter_extent), tuple(float(v) for v in bbox))


def particle_net_graphed(pn, pos, vel, box, box_feats):
    """ParticleNet.forward under autograd through HIP-graph replay, or None when this call cannot take that route (the pair
    capacities of this cloud size are not learnt yet: the eager path learns them and warms every kernel up; or a capture failed:
    the module then stays on the eager path)."""
    n = pos.shape[0]
    if pn.__dict__.get("_pair_caps", {}).get(n) is None or n == 0:
        return None
    if not (box.is_contiguous() and box_feats.is_contiguous() and box.dtype == torch.float32 and box_feats.dtype == torch.float32):
        return None
    tg = _tg_prepare(pn, pos, vel, box, box_feats)
    if tg is None:
        return None
    return _GraphedParticleNetFn.apply(pn, tg, pos, vel, *_pn_params(pn))


def _tg_prepare(pn, pos, vel, box, box_feats):
    """The graphs of this (cloud size, capacities, scene, parameter storage), captured on first use."""
    tg = pn.__dict__.setdefault("_tgraphs", _TransGraphs())
    n, dev = pos.shape[0], pos.device
    bbox, bgrid = pn._scene_bbox(box), pn._box_grid(box)         # caches filled OUTSIDE any capture (they may sync)
    key = _tg_key(pn, n, box, box_feats, bgrid, bbox)
    if tg.key == key:
        return tg
    tg.key, tg.bwd = None, {}
    tg.caps = key[1]
    tg.pos_s, tg.vel_s = torch.empty(n, 3, device=dev), torch.empty(n, 3, device=dev)
    tg.tot_pinned = torch.zeros(2, dtype=torch.int32).pin_memory()
    tg.ev = torch.cuda.Event()
    tg.box, tg.box_feats, tg.bgrid = box, box_feats, bgrid       # pins the storages the graphs read (incl. the grid workspace)
    tg.total_fluid = lambda: _tg_totals(tg)[0]
    cap_state = {"tot_pinned": tg.tot_pinned, "total_fluid": lambda: tg.total_fluid()}
    tg.pos_s.copy_(pos.detach()); tg.vel_s.copy_(vel.detach())
    torch.cuda.synchronize()
    try:
        tg.fwd = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(tg.fwd):
            tg.outs = pn._forward_impl(tg.pos_s, tg.vel_s, box, box_feats, keep=True, _capture=cap_state)
    except Exception as ex:          # noqa: BLE001  a stack that cannot capture this sequence: stay eager, say so once
        import warnings
        warnings.warn("ParticleNet.training_graph: graph capture of the training step failed (%r); using the eager path" % (ex,))
        pn.training_graph = False
        torch.cuda.synchronize()
        return None
    tg.key = key
    return tg


# The graph-replayed backward hands its gradient buffers to the parameters directly (see _GraphedParticleNetFn.backward); False = through
# autograd's AccumulateGrad (a clone per parameter).  Parameters with tensor hooks always take the autograd route.
GRAPH_GRADS_DIRECT = True


class _GraphedParticleNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pn, tg, pos, vel, *params):
        ctx.set_materialize_grads(False)
        tg.pos_s.copy_(pos.detach()); tg.vel_s.copy_(vel.detach())
        tg.fwd.replay()
        tg.ev.record()
        tg.serial += 1
        tg.checked = False
        ctx.pn, ctx.tg, ctx.serial = pn, tg, tg.serial
        pos_c, vel_c, nn, _aux = tg.outs
        out = (pos_c.clone(), vel_c.clone(), nn.clone())          # (the graph's own outputs are rewritten by the next replay)
        pn.num_fluid_neighbors, pn._y3 = out[2], _aux["ans"][-1]
        ctx.mark_non_differentiable(out[2])
        return out

    @staticmethod
    def backward(ctx, g_pos, g_vel, _g_nn):
        pn, tg = ctx.pn, ctx.tg
        if ctx.serial != tg.serial:
            raise RuntimeError("ParticleNet.training_graph: a second forward ran before this step's backward (the graphs hold ONE "
                               "step's activations: truncated BPTT of length 1); use training_graph = False for unrolled steps")
        _tg_check(pn, tg)
        dev = tg.pos_s.device
        n = tg.pos_s.shape[0]
        has_vel = g_vel is not None
        b = tg.bwd.get(has_vel)
        gp = torch.zeros(n, 3, device=dev) if g_pos is None else g_pos.detach().float()
        gv = g_vel.detach().float() if has_vel else None
        aux = tg.outs[3]
        if b is None:
            b = {"g_pos": torch.empty(n, 3, device=dev), "g_vel": torch.empty(n, 3, device=dev) if has_vel else None, "graph": None}
            b["g_pos"].copy_(gp)
            if has_vel:
                b["g_vel"].copy_(gv)
            torch.cuda.synchronize()
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.no_grad(), torch.cuda.graph(graph):
                    b["out"] = _trans_backward(pn, aux, tg.box_feats, (False, False, False), b["g_pos"], b["g_vel"])
                b["graph"] = graph
            except Exception as ex:          # noqa: BLE001  (the forward graph's activations are ordinary tensors: the eager backward reads them)
                import warnings
                warnings.warn("ParticleNet.training_graph: graph capture of the backward failed (%r); it runs eagerly" % (ex,))
                torch.cuda.synchronize()
            tg.bwd[has_vel] = b
        else:
            b["g_pos"].copy_(gp)
            if has_vel:
                b["g_vel"].copy_(gv)
        if b["graph"] is None:
            with torch.no_grad():
                return (None, None, None, None) + _trans_backward(pn, aux, tg.box_feats, (False, False, False), gp, gv)[3]
        params = _pn_params(pn)
        direct = GRAPH_GRADS_DIRECT and all(p.is_leaf and not p._backward_hooks and not getattr(p, "_post_accumulate_grad_hooks", None)
                                            for p in params)
        if direct:
            # A gradient that autograd accumulates into a leaf is CLONED when somebody else holds the tensor (the graph does): 17 copy
            # launches per step.  The replayed backward's output buffers are handed to the parameters as their .grad instead — they
            # stay valid until the next replay, i.e. through clip + optimiser step of this iteration.  A .grad left over from an
            # earlier backward (no zero_grad in between) is accumulated into; if it IS one of the graph's buffers it is moved out
            # first, so the replay cannot overwrite what was accumulated.
            for p_, g_ in zip(params, b["out"][3]):
                if p_.grad is not None and p_.grad.data_ptr() == g_.data_ptr():
                    p_.grad = p_.grad.clone()
        b["graph"].replay()
        if direct:
            for p_, g_ in zip(params, b["out"][3]):
                if p_.grad is None:
                    p_.grad = g_
                else:
                    p_.grad.add_(g_)
            return (None, None, None, None) + (None,) * len(params)
        return (None, None, None, None) + b["out"][3]


def _tg_totals(tg):
    tg.ev.synchronize()
    return int(tg.tot_pinned[0]), int(tg.tot_pinned[1])


def _tg_check(pn, tg):
    """Compare the step's true pair totals (they travelled to pinned memory inside the forward graph) with the capacities the
    graphs were captured with; on overflow raise the capacities (the next forward recaptures) and hand the step back."""
    if tg.checked:
        return
    from .transmodel import PairCapacityExceeded
    got_f, got_b = _tg_totals(tg)
    cap_f, cap_b = tg.caps
    if got_f > cap_f or got_b > cap_b:
        n = tg.pos_s.shape[0]
        pn._pair_caps[n] = (max(cap_f, ops.round_pairs(got_f + got_f // 8 + 4096)), max(cap_b, ops.round_pairs(got_b + got_b // 4 + 4096)))
        pn.pair_capacity_redos = getattr(pn, "pair_capacity_redos", 0) + 1
        raise PairCapacityExceeded("%d / %d neighbour pairs exceed the captured capacities %d / %d" % (got_f, got_b, cap_f, cap_b))
    tg.checked = True
