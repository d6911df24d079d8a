"""GPU tests of the stand-alone module forwards (models/nerf.py: Embedding :21-38, NeRF :83-124) and of the plain fp32-MFMA
GEMM that replaced the vendor BLAS calls of the training path.  A caller that keeps the reference's own
models/renderer.py (INTEGRATION.md level 2) drives exactly these."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


def T(a, dev=None):
    t = torch.from_numpy(np.asarray(a))
    return t.to(dev) if dev is not None else t


def _nerf(dev, prefix="nerf_coarse"):
    from neurofluid_amd.nerf import NeRF
    from oracle import render_oracle as ro
    st = ro.deterministic_nerf_state()
    net = NeRF(in_channels_xyz=198, in_channels_dir=54)
    net.load_state_dict({k[len(prefix) + 1:]: v for k, v in st.items() if k.startswith(prefix + ".")}, strict=True)
    return net.to(dev), st


def test_embedding_forward_vs_golden(dev):
    """A5 (models/nerf.py:21-38): Embedding.forward on the device vs what the reference's own module returned
    (tests/golden/a5_embed.npz).  Stated tolerance: 6e-8 absolute for |2^k x| <= ~1e3 — the kernel emits the correctly
    rounded sin / cos of the exact fp32 argument (double-precision sincos + angle doubling); torch's fp32 sin / cos on the
    CPU is itself up to 3.5e-8 from that value.  The pass-through columns are exact."""
    from neurofluid_amd.nerf import Embedding
    g = load_golden("a5_embed")
    for xk, ek, c, nf in (("x3", "e3_10", 3, 10), ("x3", "e3_4", 3, 4), ("x1", "e1_4", 1, 4)):
        emb = Embedding(c, nf)
        x = T(g[xk], dev)
        out = emb(x)
        ref = T(g[ek])
        assert out.shape == ref.shape == (x.shape[0], emb.out_channels)
        assert torch.equal(out[:, :c].cpu(), ref[:, :c])
        err = float((out.cpu() - ref).abs().max())
        assert err <= 6e-8 * 4, err          # x1 reaches 2^3 * 20 = 160 rad: fp32 torch.sin there is ~1e-7 from the true value
    # ragged / larger batch and leading batch dims, against the float64 definition
    gen = torch.Generator().manual_seed(5)
    x = (torch.rand(3, 1001, 3, generator=gen) * 4 - 2)
    out = Embedding(3, 10)(x.to(dev)).cpu()
    xd = x.double()
    cols = [xd] + [f(xd * 2.0 ** k) for k in range(10) for f in (torch.sin, torch.cos)]
    ref = torch.cat(cols, -1)
    assert out.shape == (3, 1001, 63)
    assert float((out.double() - ref).abs().max()) <= 6e-8


def test_embedding_backward_vs_autograd(dev):
    """d/dx of the encoding vs torch autograd through the oracle's definition in float64."""
    from neurofluid_amd.nerf import Embedding
    gen = torch.Generator().manual_seed(6)
    x = (torch.rand(257, 3, generator=gen) * 2 - 1)
    gout = torch.randn(257, 27, generator=gen)
    xg = x.to(dev).requires_grad_(True)
    Embedding(3, 4)(xg).backward(gout.to(dev))
    xd = x.double().requires_grad_(True)
    cols = [xd] + [f(xd * 2.0 ** k) for k in range(4) for f in (torch.sin, torch.cos)]
    torch.cat(cols, -1).backward(gout.double())
    torch.testing.assert_close(xg.grad.cpu().double(), xd.grad, rtol=1e-5, atol=1e-5)


def test_nerf_forward_vs_golden(dev):
    """A6 (models/nerf.py:83-124): NeRF.forward(x) and NeRF.forward(x[:, :cx], sigma_only=True) on the HIP path vs the rows
    the reference's own module returned (tests/golden/a6_nerf.npz); same tolerance as the fused path's MLP test."""
    g = load_golden("a6_nerf")
    net, _ = _nerf(dev)
    x = T(g["x"], dev)
    with torch.no_grad():
        out = net(x)
        sig = net(x[:, :198].contiguous(), sigma_only=True)
    assert out.shape == (40, 4) and sig.shape == (40, 1)
    torch.testing.assert_close(out.cpu(), T(g["out"]), rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(sig.cpu(), T(g["sigma_only"]), rtol=1e-4, atol=2e-5)
    assert torch.equal(sig, out[:, 3:4])          # sigma does not depend on the view branch
    with pytest.raises(ValueError):
        net(x[:, :100])
    with pytest.raises(RuntimeError):
        net(x.cpu())                               # no CPU fallback


def test_nerf_forward_autograd_vs_oracle(dev):
    """Parameter and INPUT gradients of the stand-alone NeRF.forward vs torch autograd through the oracle's MLP
    (what a caller that keeps the reference's renderer and trains end to end needs), incl. the sigma_only form."""
    from oracle import render_oracle as ro
    net, st = _nerf(dev, "nerf_fine")
    gen = torch.Generator().manual_seed(9)
    n = 300
    x = (torch.rand(n, 252, generator=gen) * 2 - 1)
    gout = torch.randn(n, 4, generator=gen)
    xg = x.to(dev).requires_grad_(True)
    out = net(xg)
    out.backward(gout.to(dev))
    stc = {k: v.clone().requires_grad_(k.startswith("nerf_fine")) for k, v in st.items()}
    xc = x.clone().requires_grad_(True)
    ref = ro.nerf_forward(stc, "nerf_fine", xc, 198, 54)
    ref.backward(gout)
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=2e-5)

    def rows_agree(got, want, what):
        # Row by row: the two sides run DIFFERENT forwards (GPU MFMA order vs CPU), so a hidden unit whose pre-activation sits
        # within ~1e-6 of zero takes different sides of the ReLU kink — that row's gradient then differs by O(1/16) while every
        # other row agrees to fp32 rounding (~1 such unit among 300 rows x 2 432 units; it is what the 3e-3 overall figure of
        # the first run of this test was).  Bar: median row 1e-5, at least 97 % of the rows within 1e-4, overall 2e-2.
        rr = (got - want).norm(dim=1) / (want.norm(dim=1) + 1e-30)
        assert float(rr.median()) < 1e-5 and float((rr < 1e-4).float().mean()) >= 0.97, (what, float(rr.median()), float((rr < 1e-4).float().mean()))
        assert float((got - want).norm() / want.norm()) < 2e-2, what

    rows_agree(xg.grad.cpu(), xc.grad, "dx")
    worst = 0.0
    for name, p in net.named_parameters():
        r = stc["nerf_fine." + name].grad
        rel = float((p.grad.cpu() - r).norm() / (r.norm() + 1e-30))
        worst = max(worst, rel)
        assert rel < 5e-2, (name, rel)          # the same kink rows enter the weight sums (observed 1.6e-2 on one layer; the
                                                # kernels are exact for their operands: test_backward_kernels_exact_for_their_operands)
    # sigma_only: gradient reaches x[:, :cx] and the eight trunk layers + sigma head only
    for p in net.parameters():
        p.grad = None
    xs = x[:, :198].to(dev).requires_grad_(True)
    s = net(xs, sigma_only=True)
    s.backward(gout[:, 3:4].to(dev))
    xc2 = x[:, :198].clone().requires_grad_(True)
    stc2 = {k: v.clone().requires_grad_(k.startswith("nerf_fine")) for k, v in st.items()}
    r = ro.nerf_forward(stc2, "nerf_fine", xc2, 198, 54, sigma_only=True)
    r.backward(gout[:, 3:4])
    assert xs.grad.shape == (n, 198)
    rows_agree(xs.grad.cpu(), xc2.grad, "dx (sigma_only)")
    gw = dict(net.named_parameters())["xyz_encoding_3.0.weight"].grad.cpu()
    rw = stc2["nerf_fine.xyz_encoding_3.0.weight"].grad
    assert float((gw - rw).norm() / rw.norm()) < 5e-2
    # the view branch is not evaluated in this form (models/nerf.py:100-113): like the reference, no gradient at all — not zeros
    for name, p in net.named_parameters():
        unused = name.startswith(("xyz_encoding_final", "dir_encoding", "rgb"))
        assert (p.grad is None) == unused, name
        assert (stc2["nerf_fine." + name].grad is None) == unused, name


def test_backward_kernels_exact_for_their_operands(dev):
    """The comparison with the oracle's autograd above cannot be tight: two different forwards put a hidden unit whose
    pre-activation is ~1e-7 on different sides of the ReLU kink.  This test removes that freedom: nf_nerf_wgrad (15 weight
    GEMMs + bias sums) and the dX GEMMs (nf_gemm_f32) against the float64 product of THEIR OWN operands (the dpre and the
    activations the HIP forward / backward produced), for row counts from a fraction of a slice to many slices: <= 2e-6."""
    import ctypes
    from neurofluid_amd import _lib, ops
    from neurofluid_amd._lib import check, ptr
    from neurofluid_amd.nerf import _nerf_param_struct
    from neurofluid_amd.autograd_bwd import _pack_bwd, DPRE
    lib = _lib.load()
    net, _ = _nerf(dev, "nerf_fine")
    layers = net.linear_layers()
    cx, cd = 198, 54
    for n in (7, 300, 4096, 20000):
        g = torch.Generator().manual_seed(n)
        x = (torch.rand(n, 252, generator=g) * 2 - 1).to(dev)
        gout = torch.randn(n, 4, generator=g).to(dev)
        P, keep = _nerf_param_struct(layers)
        packed = torch.empty(lib.nf_nerf_packed_floats(cx, cd), device=dev)
        check(lib.nf_nerf_pack(ctypes.byref(P), cx, cd, ptr(packed), _lib.stream()))
        X = ops.rows_to_tiles(x, cx, cd)
        n_rows = torch.tensor([n], dtype=torch.int32, device=dev)
        row_sample = torch.arange(n, dtype=torch.int32, device=dev)
        out = torch.zeros(n, 4, device=dev)
        acts = torch.empty(n * 2432, device=dev)
        check(lib.nf_nerf_mlp_fwd(ptr(packed), cx, cd, ptr(X), ptr(n_rows), n, ptr(row_sample), ptr(out), ptr(acts), _lib.stream()))
        packed_t, packed_tn = _pack_bwd(net, cx, cd, dev, n_layout=False), _pack_bwd(net, cx, cd, dev)
        dpre = torch.empty(n, DPRE, device=dev)
        check(lib.nf_nerf_mlp_bwd(ptr(packed), ptr(packed_t), cx, cd, ptr(acts), ptr(n_rows), n, ptr(row_sample), ptr(out), ptr(gout),
                                  ptr(dpre), _lib.stream()))
        # the tile-per-workgroup kernel of the training steps: the same sums in the same order, bit for bit (also on a ragged tile)
        dpre_n = torch.full((n, DPRE), float("nan"), device=dev)
        check(lib.nf_nerf_mlp_bwd_n(ptr(packed), ptr(packed_tn), cx, cd, ptr(acts), ptr(n_rows), n, ptr(row_sample), ptr(out), ptr(gout),
                                    ptr(dpre_n), _lib.stream()))
        assert torch.equal(dpre_n, dpre)
        blob = torch.empty(lib.nf_nerf_wgrad_floats(cx, cd), device=dev)
        wsp = torch.empty(lib.nf_nerf_wgrad_workspace_floats(cx, cd, 22), device=dev)
        colsum = torch.empty(DPRE, device=dev)
        check(lib.nf_nerf_wgrad(ptr(dpre), ptr(acts), ptr(X), cx, cd, n, 22, ptr(wsp), ptr(blob), ptr(colsum), _lib.stream()))
        A, D, xd = acts.view(n, 2432).double(), dpre.double(), x.double()
        o = 0
        for li, l in enumerate(layers):
            k = l.weight.numel()
            got = blob[o:o + k].view_as(l.weight).double()
            o += k
            if li == 0: ref = D[:, 0:256].t() @ xd[:, :cx]
            elif li == 4: ref = D[:, 1024:1280].t() @ torch.cat([xd[:, :cx], A[:, 768:1024]], 1)
            elif li < 8: ref = D[:, 256 * li:256 * (li + 1)].t() @ A[:, 256 * (li - 1):256 * li]
            elif li == 8: ref = D[:, 2048:2304].t() @ A[:, 1792:2048]
            elif li == 9: ref = D[:, 2304:2432].t() @ torch.cat([A[:, 2048:2304], xd[:, cx:]], 1)
            elif li == 10: ref = D[:, 2435:2436].t() @ A[:, 1792:2048]
            else: ref = D[:, 2432:2435].t() @ A[:, 2304:2432]
            assert float((got - ref).norm() / (ref.norm() + 1e-300)) <= 2e-6, (n, li)
        bias_ref = D.sum(0)
        assert float((colsum.double() - bias_ref).norm() / bias_ref.norm()) <= 2e-6
        W1, W5, Wd = layers[0].weight.detach(), layers[4].weight.detach(), layers[9].weight.detach()
        dx = torch.empty(n, cx + cd, device=dev)
        ops.gemm(dpre[:, 0:256], W1, out=dx[:, :cx])
        ops.gemm(dpre[:, 1024:1280], W5[:, :cx], out=dx[:, :cx], accumulate=True)
        ops.gemm(dpre[:, 2304:2432], Wd[:, 256:], out=dx[:, cx:])
        ref = torch.cat([D[:, 0:256] @ W1.double() + D[:, 1024:1280] @ W5[:, :cx].double(), D[:, 2304:2432] @ Wd[:, 256:].double()], 1)
        assert float((dx.double() - ref).norm() / ref.norm()) <= 2e-6, n


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (37, 5, 3), (128, 128, 32), (129, 131, 33), (300, 198, 256), (96, 4160, 4913),
                                   (4913, 96, 4160), (32, 4, 4913), (2000, 54, 128)])
def test_gemm_f32_vs_float64(dev, M, N, K):
    """nf_gemm_f32 in its four operand orders (A contiguous along k or m, B along n or k), with relu-on-load, accumulation
    and split-K, on ragged sizes, strided column slices and unaligned bases; vs the float64 product.  Tolerance: fp32
    summation of K products, 4e-7 * sqrt(K) relative to the row/column norms (observed ~1e-7 * sqrt(K))."""
    from neurofluid_amd import ops
    gen = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K + 3, generator=gen).to(dev)
    B = torch.randn(K, N + 5, generator=gen).to(dev)
    Av, Bv = A[:, 3:], B[:, 1:N + 1]            # unaligned bases, non-trivial row strides
    ref = (Av.double() @ Bv.double())
    scale = Av.double().norm(dim=1, keepdim=True) * Bv.double().norm(dim=0, keepdim=True) + 1e-30
    tol = 4e-7 * max(K, 1) ** 0.5

    def ok(got, want=ref):
        assert float(((got.double() - want).abs() / scale).max()) <= tol

    ok(ops.gemm(Av, Bv))                                                   # A along k, B along n
    ok(ops.gemm(Av.t().contiguous().t(), Bv))                              # A along m
    ok(ops.gemm(Av, Bv.t().contiguous().t()))                              # B along k
    ok(ops.gemm(Av.t().contiguous().t(), Bv.t().contiguous().t(), splits=3))
    ok(ops.gemm(Av, Bv, relu_a=True), torch.relu(Av).double() @ Bv.double())
    C = torch.randn(M, N + 2, generator=gen).to(dev)
    C0 = C.clone()
    ops.gemm(Av, Bv, out=C[:, 1:N + 1], accumulate=True, splits=2)
    ok(C[:, 1:N + 1] - C0[:, 1:N + 1])
    assert torch.equal(C[:, 0], C0[:, 0]) and torch.equal(C[:, N + 1], C0[:, N + 1])       # neighbours of the slice untouched
