"""Training MLP kernels alone on synthetic rows (dev tool): nf_nerf_mlp_fwd_n (activations saved), nf_nerf_mlp_bwd_n, nf_nerf_wgrad at the
row counts given (default: the fine / coarse pass of a 4 x 1024-ray train_renderer step).  With a -DNF_N_TIMING library the per-phase
cycle table of the forward / backward kernel is printed (tools/ab_n.py builds the variants).  usage: python tools/n_bench.py [rows ...]"""
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
raw = ctypes.CDLL(_lib.LIB_PATH)
has_prof = hasattr(raw, "nf_dev_n_prof")
FWD_PH = ["l0 bias+x", "l0 store", "l0 barrier", "bias(+x)", "hidden part", "sigma head", "store", "barrier", "dir branch", "dir store", "barrier", "rgb head",
          "end barrier"]
BWD_PH = ["slot 9", "barrier", "dir part", "slot store", "barrier", "hidden part", "slot 0", "end barrier"]


def prof(kern, names_, mfma_cycles):
    buf = (ctypes.c_ulonglong * 128)()
    assert raw.nf_dev_n_prof(buf, 1) == 0
    for w in (0, 3):
        v = [buf[kern * 64 + w * 16 + p] for p in range(16)]
        n = max(v[15], 1)
        tot = sum(v[:15])
        print("    wave %d: %.0f cycles per tile (MFMA floor %d = %.2f): " % (w, tot / n, mfma_cycles, mfma_cycles / (tot / n)) +
              "  ".join("%s %.0f" % (names_[p], v[p] / n) for p in range(len(names_))))


def span(kern, fn, label, ntiles):
    """one launch alone: the tiles' start / end stamps (s_memtime) against the launch's HIP-event time"""
    if not has_prof: return
    import numpy as np
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    buf = (ctypes.c_ulonglong * (2 * 4096 * 2))()
    assert raw.nf_dev_n_trace(buf) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(2, 4096, 2)[kern, :min(ntiles, 4096)].astype(np.int64)
    a = a[0::8]                  # workgroups go round-robin over the 8 XCDs, and every XCD has its own s_memtime base: XCD 0's tiles
    t0 = a[:, 0].min()
    st, en = np.sort(a[:, 0] - t0), np.sort(a[:, 1] - t0)
    q = lambda v, f: int(v[min(len(v) - 1, int(f * len(v)))])
    print("    %s: one launch %.1f us between events; span first start -> last end %d ticks (%.0f ticks/us); tile starts at 25/50/75/100 %%: %d %d %d %d; "
          "tile ends at 25/50/75/100 %%: %d %d %d %d; mean tile duration %d" %
          (label, us, en[-1], en[-1] / us, q(st, .25), q(st, .5), q(st, .75), st[-1], q(en, .25), q(en, .5), q(en, .75), en[-1], int((a[:, 1] - a[:, 0]).mean())))


def t(fn, it=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(it):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


rows = [int(a) for a in sys.argv[1:]] or [63000, 17000]
nsl = int(os.environ.get("NSL", 22))
for n in rows:
    g = torch.Generator(device=dev); g.manual_seed(1)
    X = torch.rand((n + 31) // 32 * 32 * 256, device=dev, generator=g) * 2 - 1
    n_rows = torch.tensor([n], dtype=torch.int32, device=dev)
    row_sample = torch.arange(n, dtype=torch.int32, device=dev)
    out = torch.zeros(n, 4, device=dev)
    acts = torch.empty(ops._round_rows(n) * 2432, device=dev)
    if has_prof: raw.nf_dev_n_prof(None, 1)
    amask = torch.zeros(lib.nf_nerf_amask_words(ops._round_rows(n)), dtype=torch.int32, device=dev)
    a = t(lambda: check(lib.nf_nerf_mlp_fwd_n2(ptr(packed_n), 198, 54, ptr(X), ptr(n_rows), n, ptr(row_sample), ptr(out), ptr(acts), ptr(amask), _lib.stream())))
    print("rows %6d (%4d tiles): fwd_n %7.1f us (%5.1f TFLOP/s of 1.332 MFLOP rows)" % (n, (n + 31) // 32, a, n * 1.331968 / a))
    if has_prof: prof(0, FWD_PH, (2 * (1 + 100) + 8 * 2 * (1 + 128) + 2 * 100 + 157) * 64)
    span(0, lambda: check(lib.nf_nerf_mlp_fwd_n2(ptr(packed_n), 198, 54, ptr(X), ptr(n_rows), n, ptr(row_sample), ptr(out), ptr(acts), ptr(amask), _lib.stream())), "fwd", (n + 31) // 32)
    packed_t = torch.empty(lib.nf_nerf_packed_bwd_floats(), device=dev)
    P = _lib.NerfParams()
    for i in range(12):
        P.w[i], P.b[i] = W[i].data_ptr(), B[i].data_ptr()
    check(lib.nf_nerf_pack_bwd(ctypes.byref(P), 198, 54, ptr(packed_t), _lib.stream()))
    packed_tn = torch.empty_like(packed_t)
    check(lib.nf_nerf_pack_bwd_n(ptr(packed_t), ptr(packed_tn), _lib.stream()))
    gr = torch.randn(n, 4, device=dev, generator=g)
    d2 = torch.zeros(ops._round_rows(n), 2436, device=dev)
    b0 = t(lambda: check(lib.nf_nerf_mlp_bwd_n(ptr(packed), ptr(packed_tn), 198, 54, ptr(acts), ptr(n_rows), n, ptr(row_sample), ptr(out), ptr(gr), ptr(d2),
                                               _lib.stream())))
    cs0 = float(d2[:n].double().abs().sum())
    if has_prof: raw.nf_dev_n_prof(None, 1)
    b = t(lambda: check(lib.nf_nerf_mlp_bwd_n2(ptr(packed), ptr(packed_tn), 198, 54, ptr(amask), ptr(n_rows), n, ptr(row_sample), ptr(out), ptr(gr), ptr(d2),
                                               _lib.stream())))
    print("             bwd_n2 %6.1f us (%5.1f TFLOP/s of 1.114 MFLOP rows)   [bwd_n, masks from the activations: %.1f us]   checksums %.6e %.6e" %
          (b, n * 1.114112 / b, b0, float(d2[:n].double().abs().sum()), cs0))
    if has_prof: prof(1, BWD_PH, (64 * 2 + 8 * 256) * 64)
    span(1, lambda: check(lib.nf_nerf_mlp_bwd_n2(ptr(packed), ptr(packed_tn), 198, 54, ptr(amask), ptr(n_rows), n, ptr(row_sample), ptr(out), ptr(gr), ptr(d2), _lib.stream())), "bwd", (n + 31) // 32)
    blob = torch.empty(lib.nf_nerf_wgrad_floats(198, 54), device=dev)
    wsp = torch.empty(lib.nf_nerf_wgrad_workspace_floats(198, 54, nsl), device=dev)
    colsum = torch.empty(2436, device=dev)
    c = t(lambda: check(lib.nf_nerf_wgrad(ptr(d2), ptr(acts), ptr(X), 198, 54, n, nsl, ptr(wsp), ptr(blob), ptr(colsum), _lib.stream())))
    print("             wgrad %7.1f us (%5.1f TFLOP/s of 2 x rows x %d weights)   checksum %.6e   out checksum %.6e" %
          (c, 2.0 * n * blob.numel() / c / 1e6, blob.numel(), float(blob.double().abs().sum()), float(out.double().abs().sum())))
    print("             sum %7.1f us" % (a + b + c))
