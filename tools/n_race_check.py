"""Dev tool: run-to-run bit-equality of the training MLP kernels (a race shows as a run that differs); with REF=path the outputs are also
compared with / written to that file (one library writes, another checks).  usage: [REF=/tmp/x.pt] python tools/n_race_check.py [rows]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neurofluid_amd import synthetic as ro
from neurofluid_amd import ops, _lib
from neurofluid_amd._lib import ptr, check
dev = torch.device("cuda:0")
st = ro.deterministic_nerf_state()
W = [st[f"nerf_coarse.{k}.weight"].to(dev) for k in ops.NERF_LAYER_NAMES]
B = [st[f"nerf_coarse.{k}.bias"].to(dev) for k in ops.NERF_LAYER_NAMES]
packed = ops.pack_nerf(W, B, 198, 54)
packed_n = ops.pack_nerf_n(packed, 198, 54)
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 17000
g = torch.Generator(device=dev); g.manual_seed(1)
X = torch.rand((n + 31) // 32 * 32 * 256, device=dev, generator=g) * 2 - 1
n_rows = torch.tensor([n], dtype=torch.int32, device=dev)
row_sample = torch.arange(n, dtype=torch.int32, device=dev)
packed_t = torch.empty(lib.nf_nerf_packed_bwd_floats(), device=dev)
P = _lib.NerfParams()
for i in range(12):
    P.w[i], P.b[i] = W[i].data_ptr(), B[i].data_ptr()
check(lib.nf_nerf_pack_bwd(ctypes.byref(P), 198, 54, ptr(packed_t), _lib.stream()))
packed_tn = torch.empty_like(packed_t)
check(lib.nf_nerf_pack_bwd_n(ptr(packed_t), ptr(packed_tn), _lib.stream()))
gr = torch.randn(n, 4, device=dev, generator=g)
res = {}
for rep in range(6):
    out = torch.zeros(n, 4, device=dev)
    acts = torch.zeros(ops._round_rows(n) * 2432, device=dev)
    amask = torch.zeros(lib.nf_nerf_amask_words(ops._round_rows(n)), dtype=torch.int32, device=dev)
    check(lib.nf_nerf_mlp_fwd_n2(ptr(packed_n), 198, 54, ptr(X), ptr(n_rows), n, ptr(row_sample), ptr(out), ptr(acts), ptr(amask), _lib.stream()))
    d1 = torch.zeros(ops._round_rows(n), 2436, device=dev); d2 = torch.zeros_like(d1)
    check(lib.nf_nerf_mlp_bwd_n(ptr(packed), ptr(packed_tn), 198, 54, ptr(acts), ptr(n_rows), n, ptr(row_sample), ptr(out), ptr(gr), ptr(d1), _lib.stream()))
    check(lib.nf_nerf_mlp_bwd_n2(ptr(packed), ptr(packed_tn), 198, 54, ptr(amask), ptr(n_rows), n, ptr(row_sample), ptr(out), ptr(gr), ptr(d2), _lib.stream()))
    torch.cuda.synchronize()
    cur = {"out": out, "acts": acts[:n * 2432], "amask": amask[:(n + 31) // 32 * 2560], "dpre_legacy": d1[:n], "dpre_bits": d2[:n]}
    if not res:
        res = {k: v.clone() for k, v in cur.items()}
    for k, v in cur.items():
        bad = int((v != res[k]).sum())
        if bad:
            print("run %d: %s differs from run 0 in %d elements" % (rep, k, bad))
            if k == "acts":
                idx = (v != res[k]).nonzero().flatten()
                rows, cols = idx // 2432, idx % 2432
                ur = torch.unique(rows)
                print("   rows %s ... (%d distinct), tiles %s; cols of first row: %s; values now %s vs %s" %
                      (ur[:8].tolist(), ur.numel(), torch.unique(ur // 32)[:8].tolist(), cols[rows == ur[0]].tolist()[:40],
                       v[idx[:4]].tolist(), res[k][idx[:4]].tolist()))
    print("run %d: legacy == bits: %s" % (rep, bool(torch.equal(d1[:n], d2[:n]))))
ref = os.environ.get("REF")
if ref:
    if os.path.exists(ref):
        want = torch.load(ref)
        for k in res:
            print("vs %s: %s %s" % (ref, k, "equal" if torch.equal(res[k].cpu(), want[k]) else "DIFFERENT (%d elements)" % int((res[k].cpu() != want[k]).sum())))
    else:
        torch.save({k: v.cpu() for k, v in res.items()}, ref)
        print("wrote", ref)
