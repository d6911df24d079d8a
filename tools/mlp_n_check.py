"""nf_nerf_mlp_fwd_n (tile per workgroup) against nf_nerf_mlp_fwd (tile per wave): bit-equality of rgbsigma and of the saved
activations, and launch times at training-step sizes (dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neurofluid_amd import synthetic as ro
from neurofluid_amd import ops, _lib
from neurofluid_amd._lib import ptr, check

dev = torch.device("cuda:0")
st = ro.deterministic_nerf_state()
names = ops.NERF_LAYER_NAMES
W = [st[f"nerf_fine.{k}.weight"].to(dev) for k in names]
B = [st[f"nerf_fine.{k}.bias"].to(dev) for k in names]
packed = ops.pack_nerf(W, B, 198, 54)
packed_n = ops.pack_nerf_n(packed, 198, 54)
lib = _lib.load()
for n in [int(a) for a in sys.argv[1:]] or [5000, 15000, 20000, 67000, 200000]:
    g = torch.Generator().manual_seed(n)
    x = (torch.rand(n, 252, generator=g) * 2 - 1).to(dev)
    X = ops.rows_to_tiles(x, 198, 54)
    n_rows = torch.tensor([n], dtype=torch.int32, device=dev)
    row_sample = torch.randperm(n, generator=g).to(torch.int32).to(dev)
    res = {}
    for name, fn, blob in (("wave", lib.nf_nerf_mlp_fwd, packed), ("wg", lib.nf_nerf_mlp_fwd_n, packed_n)):
        for save in (False, True):
            out = torch.full((n, 4), float("nan"), device=dev)
            acts = torch.full(((n + 31) // 32 * 32 * 2432,), float("nan"), device=dev) if save else None
            ms = []
            for it in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                check(fn(ptr(blob), 198, 54, ptr(X), ptr(n_rows), n, ptr(row_sample), ptr(out), ptr(acts), _lib.stream()))
                e1.record(); torch.cuda.synchronize()
                ms.append(e0.elapsed_time(e1))
            res[(name, save)] = (out, acts[:n * 2432] if save else None, min(ms))
    for save in (False, True):
        a, b = res[("wave", save)], res[("wg", save)]
        same = torch.equal(a[0], b[0]) and (not save or torch.equal(a[1], b[1]))
        print(f"rows {n:7d} save={int(save)}: tile/wave {a[2]*1e3:8.1f} us   tile/workgroup {b[2]*1e3:8.1f} us   bit-equal {same}"
              + ("" if same else f"  max|d| {float((a[0]-b[0]).abs().max()):.3g}"))
