"""the SSIM kernel (csrc/nf_metrics.hip) and PSNR against the values the reference notebook returned
(tests/golden/f4_ssim.npz) and against the CPU oracle on larger images."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def test_ssim_psnr_vs_reference_notebook_values(dev, golden):
    from neurofluid_amd import metrics
    g = golden("f4_ssim")
    for c in "abcde":
        p, t = torch.from_numpy(g[f"{c}_pred"]).to(dev), torch.from_numpy(g[f"{c}_gt"]).to(dev)
        # fp32 tolerance: the kernel applies the window separably (22 taps) where conv2d sums 121 products
        assert abs(float(metrics.ssim(p, t)) - float(g[f"{c}_ssim"])) <= 2e-6, c
        per = metrics.ssim(p, t, size_average=False).cpu().numpy()
        assert np.abs(per - g[f"{c}_ssim_per_image"]).max() <= 2e-6, c
        assert abs(float(metrics.psnr(p, t)) - float(g[f"{c}_psnr"])) <= 1e-4, c


def test_ssim_frame_sized_vs_oracle_and_properties(dev):
    from neurofluid_amd import metrics
    from oracle import metrics_oracle as mo
    g = torch.Generator().manual_seed(3)
    gt = torch.rand(2, 3, 400, 400, generator=g)
    pred = (gt + 0.05 * torch.randn(2, 3, 400, 400, generator=g)).clamp(0, 1)
    want = mo.ssim(pred, gt, size_average=False)
    got = metrics.ssim(pred.to(dev), gt.to(dev), size_average=False).cpu()
    assert (got - want).abs().max() <= 2e-6
    # identical images: exactly 1 up to fp32 rounding of the variance terms; symmetric in its arguments
    one = metrics.ssim(gt.to(dev), gt.to(dev))
    assert abs(float(one) - 1.0) <= 1e-6
    a, b = metrics.ssim(pred.to(dev), gt.to(dev)), metrics.ssim(gt.to(dev), pred.to(dev))
    assert abs(float(a) - float(b)) <= 1e-6
    with pytest.raises(RuntimeError):
        metrics.ssim(pred, gt)                     # CPU tensors: no fallback
    with pytest.raises(NotImplementedError):
        metrics.ssim(pred.to(dev), gt.to(dev), w_size=7)


def test_lpips_architecture_vs_oracle_with_stand_in_weights(dev, tmp_path):
    """SURVEY 8(f) row 4, LPIPS (utils/evaluate_images.ipynb cell 6: lpips.LPIPS(net='vgg')): neurofluid_amd.metrics.LPIPS (13 convolutions as im2col +
    nf_gemm_f32) against the oracle's restatement of the package's architecture, with a deterministic stand-in for the pretrained state dict (same keys
    and shapes — the real weights cannot be fetched here; with them the class computes the package's numbers).  Odd image sizes exercise the pools' floor."""
    from neurofluid_amd import metrics
    from oracle import metrics_oracle as mo
    sd = mo.lpips_random_weights(seed=5)
    path = str(tmp_path / "lpips_vgg.pt")
    torch.save(sd, path)                              # the documented hand-over: a saved state dict
    net = metrics.LPIPS(path, device=dev)
    g = torch.Generator().manual_seed(2)
    for shape in ((2, 3, 64, 64), (1, 3, 37, 50)):
        gt = torch.rand(*shape, generator=g)
        pred = (gt + 0.1 * torch.randn(*shape, generator=g)).clamp(0, 1)
        want = float(mo.lpips(pred, gt, sd))
        got = float(net(pred.to(dev), gt.to(dev)))
        assert want > 1e-5 and abs(got - want) <= 2e-5 * want + 1e-9, (shape, got, want)
        assert abs(float(net(gt.to(dev), gt.to(dev)))) <= 1e-12             # identical images: distance 0
        assert abs(float(net(gt.to(dev), pred.to(dev))) - got) <= 1e-6 * got      # symmetric
    with pytest.raises(RuntimeError):
        net(pred, gt)                                  # CPU tensors: no fallback
