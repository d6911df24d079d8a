import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neurofluid_amd import synthetic as ro
from neurofluid_amd import ops, _lib
dev = torch.device("cuda:0")
lib = _lib.load()
st = ro.deterministic_nerf_state()
W = [st[f"nerf_fine.{k}.weight"].to(dev) for k in ops.NERF_LAYER_NAMES]
B = [st[f"nerf_fine.{k}.bias"].to(dev) for k in ops.NERF_LAYER_NAMES]
print([tuple(w.shape) for w in W])
packed = ops.pack_nerf(W, B, 198, 54)
packed_n = ops.pack_nerf_n(packed, 198, 54)
packed_t = torch.empty(lib.nf_nerf_packed_bwd_floats(), device=dev)
P = _lib.NerfParams()
for i in range(12):
    P.w[i], P.b[i] = W[i].data_ptr(), B[i].data_ptr()
_lib.check(lib.nf_nerf_pack_bwd(ctypes.byref(P), 198, 54, packed_t.data_ptr(), _lib.stream()))
packed_tn = torch.empty_like(packed_t)
_lib.check(lib.nf_nerf_pack_bwd_n(packed_t.data_ptr(), packed_tn.data_ptr(), _lib.stream()))
g = torch.Generator().manual_seed(5)
n = live = 64
x = (torch.rand(n, 252, generator=g) * 2 - 1).to(dev)
X = ops.rows_to_tiles(x, 198, 54)
n_rows = torch.tensor([live], dtype=torch.int32, device=dev)
row_sample = torch.arange(n, dtype=torch.int32, device=dev)
out = torch.zeros(n, 4, device=dev)
acts = torch.zeros((n + 31) // 32 * 32 * 2432, device=dev)
amask = torch.zeros(lib.nf_nerf_amask_words(n), dtype=torch.int32, device=dev)
_lib.check(lib.nf_nerf_mlp_fwd_n2(packed_n.data_ptr(), 198, 54, X.data_ptr(), n_rows.data_ptr(), n, row_sample.data_ptr(), out.data_ptr(), acts.data_ptr(), amask.data_ptr(), _lib.stream()))
gout = torch.randn(n, 4, generator=g).to(dev)
d3 = torch.zeros((n + 31) // 32 * 32, 2436, device=dev)
dX = torch.full(((n + 31) // 32 * 32, 252), -5.0, device=dev)
_lib.check(lib.nf_nerf_mlp_bwd_n3(packed.data_ptr(), packed_tn.data_ptr(), 198, 54, amask.data_ptr(), n_rows.data_ptr(), n, row_sample.data_ptr(), out.data_ptr(), gout.data_ptr(), d3.data_ptr(), dX.data_ptr(), _lib.stream()))
dd = d3[:live].double()
t0 = dd[:, 0:256] @ W[0].double()
t4 = dd[:, 1024:1280] @ W[4].double()[:, :198]
td = dd[:, 2304:2432] @ W[9].double()[:, 256:]
gx = dX[:live].double()
for name, want, got in (("pos t0+t4", t0 + t4, gx[:, :198]), ("pos t0 only", t0, gx[:, :198]), ("pos t4 only", t4, gx[:, :198]), ("dir", td, gx[:, 198:])):
    print(name, "rel err", float((got - want).abs().max() / want.abs().max()), " got absmax", float(got.abs().max()), "want absmax", float(want.abs().max()))
e = (gx[:, :198] - (t0 + t4)).abs()
print("pos err by column block of 32:", [round(float(e[:, k:k + 32].max()), 6) for k in range(0, 198, 32)])
print("pos err by row:", [round(float(v), 6) for v in e.max(1).values[:8].tolist()])
ed = (gx[:, 198:] - td).abs()
print("dir err by column:", [round(float(v), 6) for v in ed.max(0).values.tolist()])
print("untouched -5 count in live rows:", int((dX[:live] == -5.0).sum()))
print("dir col 52/53 got:", gx[:4, 198 + 52:].tolist(), "want:", td[:4, 52:].tolist())
print("pos col 194..197 got:", gx[:3, 194:198].tolist(), "want:", (t0 + t4)[:3, 194:198].tolist())
# is the wrong value some other column's value?
for r in range(2):
    for cand in range(54):
        if abs(float(gx[r, 198 + 53] - td[r, cand])) < 1e-7: print("row", r, "dir col 53 holds the value of dir col", cand)
    for cand in range(198):
        if abs(float(gx[r, 197] - (t0 + t4)[r, cand])) < 1e-7: print("row", r, "pos col 197 holds value of pos col", cand)
