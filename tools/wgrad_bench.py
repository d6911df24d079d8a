"""nf_nerf_wgrad alone on random operands (dev tool): time per launch, TFLOP/s, and a float64 check of four of its GEMMs.
usage: python tools/wgrad_bench.py [rows ...]"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neurofluid_amd import _lib, ops

lib = _lib.load()
dev = torch.device("cuda:0")
cx, cd = int(os.environ.get("CX", 198)), int(os.environ.get("CD", 54))      # default: the configs' encodings (63 + 9 + 63 + 63, 27 + 27)
ptr = lambda t: ctypes.c_void_p(t.data_ptr())
ACT, DPRE = 2432, 2436
rows_list = [int(a) for a in sys.argv[1:]] or [8000, 72000]
nsl = int(os.environ.get("NSL", 22))
for n in rows_list:
    g = torch.Generator(device=dev); g.manual_seed(1)
    dpre = torch.randn(n, DPRE, device=dev, generator=g)
    acts = torch.randn(n, ACT, device=dev, generator=g)
    x = torch.randn(n, cx + cd, device=dev, generator=g)
    X = ops.rows_to_tiles(x, cx, cd)
    blob = torch.empty(lib.nf_nerf_wgrad_floats(cx, cd), device=dev)
    wsp = torch.empty(lib.nf_nerf_wgrad_workspace_floats(cx, cd, nsl), device=dev)
    colsum = torch.empty(DPRE, device=dev)
    st = _lib.stream()
    def run():
        rc = lib.nf_nerf_wgrad(ptr(dpre), ptr(acts), ptr(X), cx, cd, n, nsl, ptr(wsp), ptr(blob), ptr(colsum), st)
        assert rc == 0
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    flop = 2.0 * n * blob.numel()
    # checks: layer 0 (X-fed), layer 2, the skip layer's activation half, bias sums
    o0 = 0
    w0 = blob[:256 * cx].view(256, cx).double()
    r0 = dpre[:, :256].double().t() @ x[:, :cx].double()
    off = 256 * cx + 256 * 256
    w2 = blob[off:off + 65536].view(256, 256).double()
    r2 = dpre[:, 512:768].double().t() @ acts[:, 256:512].double()
    tot = blob.numel()
    w_rgb = blob[tot - 3 * 128:].view(3, 128).double()
    r_rgb = dpre[:, 2432:2435].double().t() @ acts[:, 2304:2432].double()
    w_sig = blob[tot - 3 * 128 - 256:tot - 3 * 128].view(1, 256).double()
    r_sig = dpre[:, 2435:2436].double().t() @ acts[:, 1792:2048].double()
    o_dir = tot - 3 * 128 - 256 - 128 * (256 + cd)
    w_dir = blob[o_dir:o_dir + 128 * (256 + cd)].view(128, 256 + cd).double()
    r_dir = dpre[:, 2304:2432].double().t() @ torch.cat([acts[:, 2048:2304], x[:, cx:]], 1).double()
    err_heads = max(((w_rgb - r_rgb).abs().max() / r_rgb.abs().max()).item(), ((w_sig - r_sig).abs().max() / r_sig.abs().max()).item(),
                    ((w_dir - r_dir).abs().max() / r_dir.abs().max()).item(),
                    ((colsum[2432:2436].double() - dpre[:, 2432:2436].double().sum(0)).abs().max() / n ** 0.5).item())
    o4 = 256 * cx + 3 * 65536
    w4 = blob[o4:o4 + 256 * (cx + 256)].view(256, cx + 256).double()
    r4 = dpre[:, 1024:1280].double().t() @ torch.cat([x[:, :cx], acts[:, 768:1024]], 1).double()
    err_heads = max(err_heads, ((w4 - r4).abs().max() / r4.abs().max()).item())
    err = max(err_heads, ((w0 - r0).abs().max() / r0.abs().max()).item(), ((w2 - r2).abs().max() / r2.abs().max()).item(),
              ((colsum[:2432].double() - dpre[:, :2432].double().sum(0)).abs().max() / n ** 0.5).item())
    print("rows %6d: %7.1f us per launch (wgrad + reduce), %6.1f TFLOP/s, rel err %.2e" % (n, us, flop / us / 1e6, err))
