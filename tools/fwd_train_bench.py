"""Training-forward MLP kernels (activations saved) on synthetic rows (dev tool): tile-per-wave nf_nerf_mlp_fwd vs
tile-per-workgroup nf_nerf_mlp_fwd_n, and nf_nerf_mlp_bwd, at the row counts given.  usage: python tools/fwd_train_bench.py [rows ...]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neurofluid_amd import synthetic as ro
from neurofluid_amd import ops, _lib
from neurofluid_amd._lib import ptr, check

dev = torch.device("cuda:0")
st = ro.deterministic_nerf_state()
names = ops.NERF_LAYER_NAMES
W = [st[f"nerf_coarse.{k}.weight"].to(dev) for k in names]
B = [st[f"nerf_coarse.{k}.bias"].to(dev) for k in names]
packed = ops.pack_nerf(W, B, 198, 54)
packed_n = ops.pack_nerf_n(packed, 198, 54)
lib = _lib.load()
rows = [int(a) for a in sys.argv[1:]] or [202 * 32, 1024 * 32, 2048 * 32, 2250 * 32]
for n in rows:
    X = torch.rand((n + 31) // 32 * 32 * 256, device=dev) * 2 - 1
    n_rows = torch.tensor([n], dtype=torch.int32, device=dev)
    row_sample = torch.arange(n, dtype=torch.int32, device=dev)
    out = torch.zeros(n, 4, device=dev)
    acts = torch.empty(ops._round_rows(n) * 2432, device=dev)
    def t(fn, it=6):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / it * 1e3
    a = t(lambda: check(lib.nf_nerf_mlp_fwd(ptr(packed), 198, 54, ptr(X), ptr(n_rows), n, ptr(row_sample), ptr(out), ptr(acts), _lib.stream())))
    b = t(lambda: check(lib.nf_nerf_mlp_fwd_n(ptr(packed_n), 198, 54, ptr(X), ptr(n_rows), n, ptr(row_sample), ptr(out), ptr(acts), _lib.stream())))
    print("rows %6d (%4d tiles): fwd tile-per-wave %7.1f us (%5.1f TFLOP/s)   tile-per-workgroup %7.1f us (%5.1f TFLOP/s)" %
          (n, (n + 31) // 32, a, n * 1.331968 / a, b, n * 1.331968 / b))
    packed_t = torch.empty(lib.nf_nerf_packed_bwd_floats(), device=dev)
    P = _lib.NerfParams()
    for i in range(12):
        P.w[i], P.b[i] = W[i].data_ptr(), B[i].data_ptr()
    check(lib.nf_nerf_pack_bwd(ctypes.byref(P), 198, 54, ptr(packed_t), _lib.stream()))
    packed_tn = torch.empty_like(packed_t)
    check(lib.nf_nerf_pack_bwd_n(ptr(packed_t), ptr(packed_tn), _lib.stream()))
    g = torch.randn(n, 4, device=dev)
    d1 = torch.zeros(ops._round_rows(n), 2436, device=dev); d2 = torch.zeros_like(d1)
    args = lambda d, pt=None: (ptr(packed), ptr(pt if pt is not None else packed_t), 198, 54, ptr(acts), ptr(n_rows), n, ptr(row_sample), ptr(out), ptr(g), ptr(d), _lib.stream())
    a = t(lambda: check(lib.nf_nerf_mlp_bwd(*args(d1))))
    b = t(lambda: check(lib.nf_nerf_mlp_bwd_n(*args(d2, packed_tn))))
    print("             bwd tile-per-wave %7.1f us (%5.1f TFLOP/s)   tile-per-workgroup %7.1f us (%5.1f TFLOP/s)   bit-equal: %s" %
          (a, n * 1.331968 / a, b, n * 1.331968 / b, bool(torch.equal(d1[:n], d2[:n]))))
