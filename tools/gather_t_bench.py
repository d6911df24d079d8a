"""Times nf_cconv_gather_bwd (k_cconv_gather_t) alone on the 4913-particle step's fluid<->fluid pairs (dev tool).
usage: python tools/gather_t_bench.py [path/to/alternative/libneurofluid_hip.so]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import shutil
    from neurofluid_amd import build
    shutil.copy(sys.argv[1], os.path.join(os.path.dirname(build.__file__), "lib", "libneurofluid_hip.so"))
import torch
import bench
from neurofluid_amd import _lib
from neurofluid_amd._lib import check, ptr
from neurofluid_amd.transmodel import ParticleNet, cconv_pairs

dev = torch.device("cuda:0")
sc = bench.build_scene(400)
pn = ParticleNet(gravity=(0, 0, -9.81)); pn.load_state_dict(sc["trans_state"], strict=True); pn = pn.to(dev)
P = sc["P"].to(dev)
_, _, _, aux = pn._forward_impl(P, torch.zeros_like(P), sc["box"].to(dev), sc["bn"].to(dev), keep=True)
f_rs, f_idx, f_pw, f_pc = aux["f"]
t_pw, t_pc = cconv_pairs(aux["pos_new"], aux["pos_new"], f_rs, f_idx, aux["f_d2"], float(pn.filter_extent), pn.use_window, negate=True)
n = P.shape[0]
lib = _lib.load()
print("pairs", f_idx.shape[0], "per particle %.1f" % (f_idx.shape[0] / n))
for cout in (64, 3):
    dy = torch.randn(n, cout, device=dev)
    dG = torch.empty(n, 65 * cout, device=dev)
    for _ in range(3):
        check(lib.nf_cconv_gather_bwd(ptr(dy), cout, ptr(f_rs), ptr(f_idx), ptr(t_pw), ptr(t_pc), n, ptr(dG), _lib.stream()), "x")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        check(lib.nf_cconv_gather_bwd(ptr(dy), cout, ptr(f_rs), ptr(f_idx), ptr(t_pw), ptr(t_pc), n, ptr(dG), _lib.stream()), "x")
    e1.record(); torch.cuda.synchronize()
    # reference: the same sum as one scatter-add over (pair, corner) entries
    npair = f_idx.shape[0]
    counts = (f_rs[1:] - f_rs[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(n, device=dev), counts)
    ref = torch.zeros(n, 65, cout, device=dev, dtype=torch.float64)
    w = t_pw[:npair * 8].view(npair, 8).double()
    cells = t_pc[:npair * 8].view(npair, 8).long()
    g = dy[f_idx.long()].double()
    for k in range(8):
        ref.index_put_((rows, cells[:, k]), w[:, k, None] * g, accumulate=True)
    ref[:, 64, :] = dy.double()
    err = float((dG.view(n, 65, cout).double() - ref).abs().max() / ref.abs().max())
    print("cout %d: %.1f us per launch, max err / max |ref| %.2e" % (cout, e0.elapsed_time(e1) / 20 * 1e3, err))
