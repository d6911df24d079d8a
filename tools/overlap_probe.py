"""What would pipelining the fine pass's data gradient and weight gradient over two row halves buy (dev tool)?  Times, on synthetic rows:
  serial     bwd_n(N) ; wgrad(N)                                        — the step's order today
  pipelined  bwd_n(N/2) ; [bwd_n(N/2) on stream 1 || wgrad(N/2) on stream 2] ; wgrad(N/2)
usage: python tools/overlap_probe.py [rows]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neurofluid_amd import synthetic as ro
from neurofluid_amd import ops, _lib
from neurofluid_amd._lib import ptr, check

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 72000
st = ro.deterministic_nerf_state()
names = ops.NERF_LAYER_NAMES
W = [st[f"nerf_coarse.{k}.weight"].to(dev) for k in names]
B = [st[f"nerf_coarse.{k}.bias"].to(dev) for k in names]
packed = ops.pack_nerf(W, B, 198, 54)
lib = _lib.load()
packed_t = torch.empty(lib.nf_nerf_packed_bwd_floats(), device=dev)
P = _lib.NerfParams()
for i in range(12):
    P.w[i], P.b[i] = W[i].data_ptr(), B[i].data_ptr()
check(lib.nf_nerf_pack_bwd(ctypes.byref(P), 198, 54, ptr(packed_t), _lib.stream()))
packed_tn = torch.empty_like(packed_t)
check(lib.nf_nerf_pack_bwd_n(ptr(packed_t), ptr(packed_tn), _lib.stream()))


def make(n):
    d = {}
    d["n"] = n
    d["X"] = torch.rand((n + 31) // 32 * 32 * 256, device=dev) * 2 - 1
    d["n_rows"] = torch.tensor([n], dtype=torch.int32, device=dev)
    d["rs"] = torch.arange(n, dtype=torch.int32, device=dev)
    d["out"] = torch.rand(n, 4, device=dev)
    d["acts"] = torch.randn(ops._round_rows(n) * 2432, device=dev)
    d["g"] = torch.randn(n, 4, device=dev)
    d["dpre"] = torch.zeros(ops._round_rows(n), 2436, device=dev)
    d["blob"] = torch.empty(lib.nf_nerf_wgrad_floats(198, 54), device=dev)
    d["wsp"] = torch.empty(lib.nf_nerf_wgrad_workspace_floats(198, 54, 22), device=dev)
    d["cs"] = torch.empty(2436, device=dev)
    return d


def bwd(d, s):
    check(lib.nf_nerf_mlp_bwd_n(ptr(packed), ptr(packed_tn), 198, 54, ptr(d["acts"]), ptr(d["n_rows"]), d["n"], ptr(d["rs"]), ptr(d["out"]), ptr(d["g"]),
                                ptr(d["dpre"]), s.cuda_stream))


def wgrad(d, s):
    check(lib.nf_nerf_wgrad(ptr(d["dpre"]), ptr(d["acts"]), ptr(d["X"]), 198, 54, d["n"], 22, ptr(d["wsp"]), ptr(d["blob"]), ptr(d["cs"]), s.cuda_stream))


full, h1, h2 = make(N), make(N // 2), make(N - N // 2)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def serial():
    bwd(full, s1); wgrad(full, s1)


def pipelined():
    bwd(h1, s1)
    ev = torch.cuda.Event(); ev.record(s1)
    s2.wait_event(ev)
    bwd(h2, s1)
    wgrad(h1, s2)
    ev2 = torch.cuda.Event(); ev2.record(s2)
    s1.wait_event(ev2)
    wgrad(h2, s1)


def timeit(fn, it=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s1)
    for _ in range(it):
        fn()
    e1.record(s1)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for rep in range(2):
    a = timeit(serial)
    b = timeit(pipelined)
    c = timeit(lambda: bwd(full, s1)); d = timeit(lambda: wgrad(full, s1))
    e = timeit(lambda: bwd(h1, s1)); f = timeit(lambda: wgrad(h1, s1))
    print("rows %d: serial %.0f us (bwd %.0f + wgrad %.0f), pipelined over two halves %.0f us (half: bwd %.0f, wgrad %.0f)" % (N, a, c, d, b, e, f), flush=True)
