"""C2 on the GPU, numerically: the coupled gradient of end-to-end training (trainer/trainer_e2e.py:189-261)
    rgb loss -> RenderNet (both passes) -> dL/d(pred_pos) -> ParticleNet parameters
HIP (both autograd Functions chained by torch) vs torch autograd through BOTH oracles."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_coupled_e2e_gradients_vs_oracle_autograd(dev):
    from test_gpu_render import make_net, _fluid_rays, _oracle_render_diff
    from test_gpu_trans import make_pn, _oracle_state_with_grad
    from neurofluid_amd.autograd import _run_passes
    from oracle import render_oracle as ro, trans_oracle as to
    net = make_net(dev)
    pn, _ = make_pn(dev)
    P = ro.watercube_particles()
    V = torch.zeros_like(P); V[:, 2] = -0.4
    box, bn = to.watercube_box()
    rays, roc = _fluid_rays(100)
    tgt = torch.rand(rays.shape[0], 3, generator=torch.Generator().manual_seed(9))
    lo_b, hi_b = torch.tensor([-0.975, -0.975, -0.975]), torch.tensor([0.975, 0.975, 2.4302])

    def boundary(pos, lo, hi):          # trainer/basetrainer.py:58-70,141-143: L1 to the clipped positions
        return torch.nn.functional.l1_loss(pos, torch.max(torch.min(pos, hi), lo))

    # ---- HIP
    pred, _, _ = pn(P.to(dev), V.to(dev), box.to(dev), bn.to(dev))
    with torch.no_grad():
        _, p1, _, _, _ = _run_passes(net, pred.detach(), roc.to(dev), rays.to(dev), True, True, save_acts=True)
    z1 = p1.z.cpu()
    out = net(pred, roc.to(dev), rays.to(dev), None, None)
    loss = torch.nn.functional.mse_loss(out["rgb0"], tgt.to(dev)) + torch.nn.functional.mse_loss(out["rgb1"], tgt.to(dev)) \
        + boundary(pred, lo_b.to(dev), hi_b.to(dev))
    loss.backward()
    # ---- oracles: the transition oracle's own predicted positions carry the graph; their VALUES are replaced by the
    # HIP path's (they agree to ~1e-7, but the first-K neighbour sets of the renderer are discontinuous in them)
    st_t = _oracle_state_with_grad()
    pred_o, _, _ = to.particle_net_forward(st_t, P, V, box, bn)
    assert float((pred_o.detach() - pred.detach().cpu()).norm(dim=-1).mean()) < 1e-6
    pos_o = pred_o + (pred.detach().cpu() - pred_o.detach())
    st_r = {k: v.clone().requires_grad_(True) for k, v in ro.deterministic_nerf_state().items()}
    lo = _oracle_render_diff(st_r, pos_o, roc, rays, z1, tgt) + boundary(pos_o, lo_b, hi_b)
    lo.backward()
    assert abs(float(loss.detach()) - float(lo.detach())) <= 1e-5
    worst_t, worst_r = 0.0, 0.0
    for name, prm in pn.named_parameters():
        ref = st_t[name].grad
        assert ref is not None and float(ref.norm()) > 0, name
        rel = float((prm.grad.cpu() - ref).norm() / ref.norm())
        worst_t = max(worst_t, rel)
        assert rel < 2e-2, (name, rel)
    for name, prm in net.named_parameters():
        ref = st_r[name].grad
        rel = float((prm.grad.cpu() - ref).norm() / ref.norm())
        worst_r = max(worst_r, rel)
        assert rel < 2e-2, (name, rel)
    print("coupled e2e gradients: worst relative error  transition params", worst_t, " renderer params", worst_r)


def test_backward_glue_kernels_vs_torch(dev):
    """The three kernels that replaced ATen glue in the transition model's backward (csrc/nf_host.hip): nf_colsum (bias gradients: column sums of a
    row-major matrix or of a column slice of a wider one, aligned and unaligned, with the second copy), nf_relu_bwd_add and nf_cconv_split_db,
    each against the torch expression it stands for.  Column sums: a fixed summation order, so two runs are bit-equal; vs float64 within 2e-6 relative."""
    from neurofluid_amd import _lib
    from neurofluid_amd._lib import check, ptr
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    for rows, cols, lda, off in ((4913, 64, 64, 0), (4913, 96, 96, 0), (1, 64, 64, 0), (0, 32, 32, 0), (777, 3, 3, 0), (5000, 64, 195, 65), (300, 37, 40, 2)):
        full = torch.randn(max(rows, 1), lda, generator=g).to(dev)
        a = full[:, off:]
        out, out2 = torch.full((cols,), 7.0, device=dev), torch.full((cols,), 7.0, device=dev)
        check(lib.nf_colsum(a.data_ptr(), rows, cols, lda, ptr(out), ptr(out2), _lib.stream()), "nf_colsum")
        ref = full[:rows, off:off + cols].double().sum(0)
        scale = full[:rows, off:off + cols].double().abs().sum(0).clamp_min(1e-30)
        assert float(((out.double() - ref).abs() / scale).max()) < 2e-6 if rows else bool((out == 0).all())
        assert torch.equal(out, out2)
        again = torch.empty(cols, device=dev)
        check(lib.nf_colsum(a.data_ptr(), rows, cols, lda, ptr(again), None, _lib.stream()), "nf_colsum")
        assert torch.equal(again, out)
    dx, prev, res = (torch.randn(1000, 64, generator=g).to(dev) for _ in range(3))
    o = torch.empty_like(dx)
    check(lib.nf_relu_bwd_add(ptr(dx), ptr(prev), ptr(res), ptr(o), dx.numel(), _lib.stream()), "nf_relu_bwd_add")
    assert torch.equal(o, torch.where(prev > 0, dx, torch.zeros_like(dx)) + res)
    check(lib.nf_relu_bwd_add(ptr(dx), ptr(prev), None, ptr(o), dx.numel(), _lib.stream()), "nf_relu_bwd_add")
    assert torch.equal(o, torch.where(prev > 0, dx, torch.zeros_like(dx)))
    cin, cout = 96, 64
    dB = torch.randn(cin, 65 * cout, generator=g).to(dev)
    dK, dW = torch.empty(64, cin, cout, device=dev), torch.empty(cout, cin, device=dev)
    check(lib.nf_cconv_split_db(ptr(dB), cin, cout, ptr(dK), ptr(dW), _lib.stream()), "nf_cconv_split_db")
    assert torch.equal(dK, dB[:, :64 * cout].reshape(cin, 64, cout).permute(1, 0, 2))
    assert torch.equal(dW, dB[:, 64 * cout:].t())
