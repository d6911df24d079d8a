"""Micro-benchmark of the MFMA MLP kernel on synthetic feature rows (dev tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neurofluid_amd import synthetic as ro
from neurofluid_amd import ops, _lib
from neurofluid_amd._lib import ptr, check

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768 * 10
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
st = ro.deterministic_nerf_state()
names = ops.NERF_LAYER_NAMES
W = [st[f"nerf_coarse.{k}.weight"].to(dev) for k in names]
B = [st[f"nerf_coarse.{k}.bias"].to(dev) for k in names]
packed = ops.pack_nerf(W, B, 198, 54)
lib = _lib.load()
X = torch.rand((n + 31) // 32 * 32 * 256, device=dev) * 2 - 1
n_rows = torch.tensor([n], dtype=torch.int32, device=dev)
row_sample = torch.arange(n, dtype=torch.int32, device=dev)
out = torch.zeros(n, 4, device=dev)
for it in range(iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib.nf_nerf_mlp_fwd(ptr(packed), 198, 54, ptr(X), ptr(n_rows), n, ptr(row_sample), ptr(out), None, _lib.stream()))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(f"iter {it}: rows {n} {ms:.3f} ms  {n*1331968/ms/1e9:.1f} TFLOP/s")
if len(sys.argv) > 3 and sys.argv[3] == "ring":
    ws = torch.empty(lib.nf_nerf_stream_floats(198, 54), dtype=torch.float32, device=dev)
    check(lib.nf_nerf_pack_stream(ptr(packed), 198, 54, ptr(ws), _lib.stream()))
    out3 = torch.zeros(n, 4, device=dev)
    for it in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.nf_nerf_mlp_fwd_l(ptr(packed), ptr(ws), 198, 54, ptr(X), ptr(n_rows), n, ptr(row_sample), ptr(out3), _lib.stream()))
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print(f"ring iter {it}: rows {n} {ms:.3f} ms  {n*1331968/ms/1e9:.1f} TFLOP/s")
    err = (out3 - out).abs()
    print("ring vs direct: max abs err rgb", float(err[:, :3].max()), "sigma", float(err[:, 3].max()))
if len(sys.argv) > 3 and sys.argv[3] == "fp16v3":
    ph = ops.pack_nerf_h2(W, B, 198, 54)
    T = X.numel() // (32 * 256)
    Xh = X.view(T, 16, 2, 64, 4).permute(0, 1, 3, 2, 4).reshape(-1).to(torch.float16).contiguous()
    if T % 2:
        Xh = torch.cat([Xh, torch.zeros(Xh.numel() // T, dtype=Xh.dtype, device=dev)])
    out2 = torch.zeros(n, 4, device=dev)
    for it in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.nf_nerf_mlp_fwd_h2(ptr(ph.blob), 198, 54, ptr(Xh), ptr(n_rows), n, ptr(row_sample), ptr(out2), _lib.stream()))
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print(f"fp16 v3 iter {it}: rows {n} {ms:.3f} ms  {n*1331968/ms/1e9:.1f} TFLOP/s")
    err = (out2 - out).abs()
    print("v3 max abs err rgb", float(err[:, :3].max()), "sigma", float(err[:, 3].max()), "sigma scale", float(out[:, 3].abs().max()))
if len(sys.argv) > 3 and sys.argv[3] == "split":
    ps = ops.pack_nerf_s(W, B, 198, 54)
    out2 = torch.zeros(n, 4, device=dev)
    for it in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.nf_nerf_mlp_fwd_s(ptr(ps.blob), 198, 54, ptr(X), ptr(n_rows), n, ptr(row_sample), ptr(out2), _lib.stream()))
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print(f"split iter {it}: rows {n} {ms:.3f} ms  {n*1331968/ms/1e9:.1f} TFLOP/s fp32-equivalent ({3*n*1331968/ms/1e9:.0f} TFLOP/s of fp16 MFMA)")
    err = (out2 - out).abs()
    print("split max abs err rgb", float(err[:, :3].max()), "sigma", float(err[:, 3].max()), "sigma scale", float(out[:, 3].abs().max()))
if len(sys.argv) > 3 and sys.argv[3] == "asmonly":
    # timing of the hand-scheduled kernel alone + equality with the direct kernel's rows (A/B variant libraries)
    wa = torch.empty(lib.nf_nerf_stream_a_floats(198, 54), dtype=torch.float32, device=dev)
    check(lib.nf_nerf_pack_stream_a(ptr(packed), 198, 54, ptr(wa), _lib.stream()))
    ws = torch.empty(lib.nf_nerf_stream_floats(198, 54), dtype=torch.float32, device=dev)
    check(lib.nf_nerf_pack_stream(ptr(packed), 198, 54, ptr(ws), _lib.stream()))
    out3 = torch.zeros(n, 4, device=dev)
    check(lib.nf_nerf_mlp_fwd_l(ptr(packed), ptr(ws), 198, 54, ptr(X), ptr(n_rows), n, ptr(row_sample), ptr(out3), _lib.stream()))
    out4 = torch.full((n, 4), 7.0, device=dev)
    for it in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.nf_nerf_mlp_fwd_a(ptr(packed), ptr(wa), 198, 54, ptr(X), ptr(n_rows), n, ptr(row_sample), ptr(out4), _lib.stream()))
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print(f"asm iter {it}: rows {n} {ms:.3f} ms  {n*1331968/ms/1e9:.1f} TFLOP/s", flush=True)
    print("asm vs ring: bit-equal", bool(torch.equal(out3, out4)))
if len(sys.argv) > 3 and sys.argv[3] == "asm":
    # hand-scheduled kernel (nf_mlp_a.hip) vs the compiler-scheduled ring kernel: must be bit-identical
    ws = torch.empty(lib.nf_nerf_stream_floats(198, 54), dtype=torch.float32, device=dev)
    check(lib.nf_nerf_pack_stream(ptr(packed), 198, 54, ptr(ws), _lib.stream()))
    wa = torch.empty(lib.nf_nerf_stream_a_floats(198, 54), dtype=torch.float32, device=dev)
    check(lib.nf_nerf_pack_stream_a(ptr(packed), 198, 54, ptr(wa), _lib.stream()))
    out3 = torch.zeros(n, 4, device=dev)
    out4 = torch.full((n, 4), 7.0, device=dev)
    for name, fn, w, o in (("ring", lib.nf_nerf_mlp_fwd_l, ws, out3), ("asm", lib.nf_nerf_mlp_fwd_a, wa, out4)):
        for it in range(iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(fn(ptr(packed), ptr(w), 198, 54, ptr(X), ptr(n_rows), n, ptr(row_sample), ptr(o), _lib.stream()))
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            print(f"{name} iter {it}: rows {n} {ms:.3f} ms  {n*1331968/ms/1e9:.1f} TFLOP/s", flush=True)
    err = (out4 - out3).abs()
    print("asm vs ring: max abs err rgb", float(err[:, :3].max()), "sigma", float(err[:, 3].max()), "bit-equal", bool(torch.equal(out3, out4)),
          "rows differing", int((err.max(1)[0] > 0).sum()))
    bad = torch.nonzero(err.max(1)[0] > 0).flatten()[:8].tolist()
    for r in bad:
        print(" row", r, out3[r].tolist(), out4[r].tolist())
    if len(sys.argv) > 4:
        eq = (err.max(1)[0] == 0)
        for t0 in (0, 32, 64, 96, 128, 32 * 1024, 32 * 2048):
            print("tile", t0 // 32, "".join("1" if bool(e) else "." for e in eq[t0:t0 + 32].tolist()),
                  " sigma finite", "".join("1" if bool(e) else "x" for e in torch.isfinite(out4[t0:t0 + 32, 3]).tolist()))
        print("equal rows per tile (first 16 tiles):", eq[:512].view(16, 32).sum(1).tolist())
        per_tile = eq.view(-1, 32).sum(1)
        print("fully equal tiles:", int((per_tile == 32).sum()), "of", per_tile.numel(), " partially:", int(((per_tile > 0) & (per_tile < 32)).sum()))
        full = torch.nonzero(per_tile == 32).flatten()
        print("first equal tiles:", full[:20].tolist(), " last:", full[-5:].tolist())
        print("equal tiles by (tile % 4):", [int((per_tile.view(-1, 4)[:, w] == 32).sum()) for w in range(4)])
        print("equal tiles by round:", [int((per_tile[r * 1024:(r + 1) * 1024] == 32).sum()) for r in range(per_tile.numel() // 1024)])
if len(sys.argv) > 3 and sys.argv[3] == "fp16asm":
    # hand-scheduled fp16 kernel (nf_mlp_ha.hip) vs the compiler-scheduled one (nf_mlp_h2.hip): must be bit-identical
    ph = ops.pack_nerf_h2(W, B, 198, 54)
    T = X.numel() // (32 * 256)
    Xh = X.view(T, 16, 2, 64, 4).permute(0, 1, 3, 2, 4).reshape(-1).to(torch.float16).contiguous()
    if T % 2:
        Xh = torch.cat([Xh, torch.zeros(Xh.numel() // T, dtype=Xh.dtype, device=dev)])
    out2 = torch.zeros(n, 4, device=dev)
    out5 = torch.full((n, 4), 7.0, device=dev)
    for name, fn, o, blob in (("h2", lib.nf_nerf_mlp_fwd_h2, out2, ph.blob), ("ha", lib.nf_nerf_mlp_fwd_ha, out5, ph.blob_ha)):
        for it in range(iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(fn(ptr(blob), 198, 54, ptr(Xh), ptr(n_rows), n, ptr(row_sample), ptr(o), _lib.stream()))
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            print(f"{name} iter {it}: rows {n} {ms:.3f} ms  {n*1331968/ms/1e9:.1f} TFLOP/s", flush=True)
    err = (out5 - out2).abs()
    print("ha vs h2: max abs err rgb", float(err[:, :3].max()), "sigma", float(err[:, 3].max()), "bit-equal", bool(torch.equal(out2, out5)),
          "rows differing", int((err.max(1)[0] > 0).sum()), " vs fp32: rgb", float((out5 - out)[:, :3].abs().max()))
    bad = torch.nonzero(err.max(1)[0] > 0).flatten()[:8].tolist()
    for r in bad:
        print(" row", r, out2[r].tolist(), out5[r].tolist())
    eq = (err.max(1)[0] == 0)
    per_pair = eq[:n // 64 * 64].view(-1, 64).sum(1)
    print("fully equal pairs:", int((per_pair == 64).sum()), "of", per_pair.numel(), " by wave:", [int((per_pair.view(-1, 4)[:, w] == 64).sum()) for w in range(4)] if per_pair.numel() % 4 == 0 else "")
if len(sys.argv) > 3 and sys.argv[3] == "fp16asmonly":
    ph = ops.pack_nerf_h2(W, B, 198, 54)
    T = X.numel() // (32 * 256)
    Xh = X.view(T, 16, 2, 64, 4).permute(0, 1, 3, 2, 4).reshape(-1).to(torch.float16).contiguous()
    if T % 2:
        Xh = torch.cat([Xh, torch.zeros(Xh.numel() // T, dtype=Xh.dtype, device=dev)])
    out2 = torch.zeros(n, 4, device=dev)
    out5 = torch.full((n, 4), 7.0, device=dev)
    check(lib.nf_nerf_mlp_fwd_h2(ptr(ph.blob), 198, 54, ptr(Xh), ptr(n_rows), n, ptr(row_sample), ptr(out2), _lib.stream()))
    for it in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.nf_nerf_mlp_fwd_ha(ptr(ph.blob_ha), 198, 54, ptr(Xh), ptr(n_rows), n, ptr(row_sample), ptr(out5), _lib.stream()))
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print(f"ha iter {it}: rows {n} {ms:.3f} ms  {n*1331968/ms/1e9:.1f} TFLOP/s", flush=True)
    print("ha vs h2: bit-equal", bool(torch.equal(out2, out5)))
    if os.environ.get("NF_HA_TIMING_READ"):
        tt = ph.blob_ha[1336 * 1024:1336 * 1024 + 8].view(torch.int32).cpu().tolist()
        pairs = (n + 63) // 64
        rounds = (pairs + 1023) // 1024
        print(f"timing: {tt[0]} shader cycles, {tt[1]} ticks of 10 ns -> {tt[0] / max(tt[1], 1) * 0.1:.3f} GHz, {tt[0] / (rounds * 2672):.2f} cycles per MFMA over {rounds} pairs per wave", flush=True)
