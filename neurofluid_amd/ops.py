"""Tensor-level wrappers over the C ABI (include/neurofluid_hip.h).

These are the operator boundaries of SURVEY §8b:
  * ``ball_query``            — pytorch3d.ops.ball_query drop-in          (models/renderer.py:116-118)
  * ``fixed_radius_search``   — Open3D FixedRadiusSearch drop-in          (models/transmodel.py:86-95)
  * ``render_chunk``          — the fused body of RenderNet.forward       (models/renderer.py:211-270)
All tensors must be CUDA (ROCm) fp32/int tensors; torch only provides memory and the stream.
"""
import ctypes
import os
import math

import torch

from . import _lib
from ._lib import check, ptr


# Opt-in experiment (default off = 1): inference passes with at least OVERLAP_MIN_ROWS rows pipelined in OVERLAP_CHUNKS row
# chunks over two streams (features of chunk c + 1 behind the MLP of chunk c; render_pass).  Measured with 4 chunks: the
# kernels DO co-run (rocprofv3 trace: separate queues, overlapping intervals), but the co-running feature kernel takes 4x
# its solo time and the MLP chunk 1.35x — the matrix kernel leaves neither issue slots nor power for the feature stage's
# double-precision VALU work: fp32 frame 34.9 -> 35.3 ms, fp16 6.7 -> 7.0 ms, split 14.3 -> 14.1 ms.
OVERLAP_CHUNKS = 1
OVERLAP_MIN_ROWS = 131072

# bench.py sets PROFILE = {"mlp": [], "rows": []} to collect (start, end) HIP events around every MLP launch
# (recorded on the stream the kernel is launched on) and the executed-row counts.
PROFILE = None


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("neurofluid_amd ops need tensors on the GPU (no CPU fallback)")


# ------------------------------------------------------------------------------------------------
# grid
# ------------------------------------------------------------------------------------------------
class Grid:
    """Uniform cell grid over a point cloud; `ws` is the opaque device workspace of nf_grid_build."""

    def __init__(self, points, cell, bbox, ws, firstk=True):
        self.points, self.cell, self.bbox, self.ws, self.firstk = points, float(cell), bbox, ws, firstk

    def need_firstk(self):
        if not self.firstk:
            raise RuntimeError("this grid was built for radius search only (build_grid(..., firstk=False))")

    @property
    def n(self):
        return self.points.shape[0]

    def aabb_words(self):
        """The exact bounds of the points as the build left them on the device: 6 int32 words (see decode_aabb)."""
        off = _lib.load().nf_grid_points_aabb_offset()
        return self.ws[off:off + 24].view(torch.int32)


def decode_aabb(words):
    """6 order-preserving uint32 words (as Python ints, possibly negative from an int32 view) -> (lo xyz, hi xyz) floats."""
    import struct
    out = []
    for w in words:
        u = w & 0xffffffff
        b = (u & 0x7fffffff) if (u & 0x80000000) else (~u & 0xffffffff)
        out.append(struct.unpack("<f", struct.pack("<I", b))[0])
    return tuple(out)


def build_grid(points, cell, bbox=None, firstk=True):
    """points (N,3) fp32 contiguous.  bbox=(xmin,ymin,zmin,xmax,ymax,zmax); if None it is computed
    from the points (one device->host sync).  firstk=False builds the cell lists only (fixed-radius search)."""
    _require_cuda(points)
    lib = _lib.load()
    pts = points.detach().contiguous().float()
    if bbox is None:
        if pts.shape[0] == 0:
            bbox = (0.0, 0.0, 0.0, 0.0, 0.0, 0.0)
        else:
            lo, hi = torch.aminmax(pts, dim=0)
            bbox = tuple(lo.tolist()) + tuple(hi.tolist())
    bb = (ctypes.c_float * 6)(*[float(v) for v in bbox])
    nbytes = lib.nf_grid_workspace_bytes(pts.shape[0], float(cell), bb)
    if nbytes == 0:
        raise RuntimeError("nf_grid_workspace_bytes: bad grid parameters")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=pts.device)
    check(lib.nf_grid_build(ptr(pts), pts.shape[0], float(cell), bb, ptr(ws), nbytes, int(bool(firstk)), _lib.stream()),
          "nf_grid_build")
    return Grid(pts, cell, tuple(bbox), ws, bool(firstk))


def ball_query(p1, p2, radius, K, return_nn=True):
    """pytorch3d.ops.ball_query(p1 (N,P1,3), p2 (N,P2,3), radius, K) -> (dists (N,P1,K) squared, pad 0;
    idx (N,P1,K) int64, pad -1; nn (N,P1,K,3), pad 0).  First K in index order."""
    _require_cuda(p1, p2)
    lib = _lib.load()
    N, P1, _ = p1.shape
    dists = torch.empty(N, P1, K, dtype=torch.float32, device=p1.device)
    idx = torch.empty(N, P1, K, dtype=torch.int64, device=p1.device)
    nn = torch.empty(N, P1, K, 3, dtype=torch.float32, device=p1.device) if return_nn else None
    for b in range(N):
        same = b > 0 and p2[b].data_ptr() == p2[0].data_ptr()
        if not same:
            grid = build_grid(p2[b], radius)
        q = p1[b].detach().contiguous().float()
        check(lib.nf_ball_query_firstk(ptr(grid.ws), ptr(grid.points), ptr(q), P1, float(radius), K, ptr(dists[b]),
                                       ptr(idx[b]), ptr(nn[b]) if return_nn else None, _lib.stream()),
              "nf_ball_query_firstk")
    return dists, idx, nn


def grid_ball_query(grid, queries, radius, K):
    """Single-cloud variant on a prebuilt grid: queries (Q,3)."""
    if radius > grid.cell * (1 + 1e-6):
        raise RuntimeError("query radius exceeds the grid cell edge")
    grid.need_firstk()
    lib = _lib.load()
    q = queries.detach().contiguous().float()
    Q = q.shape[0]
    dists = torch.empty(Q, K, dtype=torch.float32, device=q.device)
    idx = torch.empty(Q, K, dtype=torch.int64, device=q.device)
    nn = torch.empty(Q, K, 3, dtype=torch.float32, device=q.device)
    check(lib.nf_ball_query_firstk(ptr(grid.ws), ptr(grid.points), ptr(q), Q, float(radius), K, ptr(dists), ptr(idx),
                                   ptr(nn), _lib.stream()), "nf_ball_query_firstk")
    return dists, idx, nn


def radius_row_splits(grid, queries, radius, ignore_query_point=True):
    """Counts + device-side scan -> row_splits (Q+1) int64 (no host sync)."""
    if radius > grid.cell * (1 + 1e-6):
        raise RuntimeError("query radius exceeds the grid cell edge")
    lib = _lib.load()
    q = queries.detach().contiguous().float()
    Q = q.shape[0]
    rs = torch.empty(Q + 1, dtype=torch.int64, device=q.device)
    nb = lib.nf_radius_scan_workspace_bytes(Q)
    sws = torch.empty(nb, dtype=torch.uint8, device=q.device)
    check(lib.nf_radius_count(ptr(grid.ws), ptr(q), Q, float(radius), int(ignore_query_point), ptr(rs), ptr(sws), nb,
                              _lib.stream()), "nf_radius_count")
    return rs


def round_pairs(n):
    """Capacity bucket of a pair-count-sized buffer (<= 12.5 % slack, multiples of 16 384): the pair count of a
    moving fluid drifts every step, and exact sizes send the caching allocator back to hipMalloc (4-7 ms, inside a
    frame) each time a 2 MB size class is crossed."""
    n = max(int(n), 1)
    step = max(16384, 1 << max(n.bit_length() - 4, 0))
    return (n + step - 1) // step * step


def radius_fill(grid, queries, radius, row_splits, capacity, ignore_query_point=True):
    lib = _lib.load()
    q = queries.detach().contiguous().float()
    idx = torch.empty(round_pairs(capacity), dtype=torch.int32, device=q.device)[:max(capacity, 1)]
    d2 = torch.empty(round_pairs(capacity), dtype=torch.float32, device=q.device)[:max(capacity, 1)]
    check(lib.nf_radius_fill(ptr(grid.ws), ptr(q), q.shape[0], float(radius), int(ignore_query_point), ptr(row_splits),
                             ptr(idx), ptr(d2), capacity, _lib.stream()), "nf_radius_fill")
    return idx, d2


def nearest(points, queries, return_idx=False):
    """Exact nearest neighbour of every query among points: (nq,) float64 distances [, (nq,) int32 indices]."""
    lib = _lib.load()
    pts = points.detach().contiguous().float()
    q = queries.detach().contiguous().float()
    dist = torch.empty(q.shape[0], dtype=torch.float64, device=q.device)
    idx = torch.empty(q.shape[0], dtype=torch.int32, device=q.device) if return_idx else None
    check(lib.nf_nearest(ptr(pts), pts.shape[0], ptr(q), q.shape[0], ptr(dist), ptr(idx), _lib.stream()), "nf_nearest")
    return (dist, idx) if return_idx else dist


def fixed_radius_search(points, queries, radius, ignore_query_point=True, grid=None):
    """Open3D FixedRadiusSearch contract -> (neighbors_index int32 (nnz), neighbors_row_splits int64 (Q+1),
    neighbors_distance fp32 (nnz) = squared distance).  One host sync (nnz sizes the outputs)."""
    _require_cuda(points, queries)
    if grid is None:
        grid = build_grid(points, radius)
    rs = radius_row_splits(grid, queries, radius, ignore_query_point)
    nnz = int(rs[-1].item())
    idx, d2 = radius_fill(grid, queries, radius, rs, nnz, ignore_query_point)
    return idx[:nnz], rs, d2[:nnz]


# ------------------------------------------------------------------------------------------------
# plain fp32 GEMM on the matrix pipe (nf_gemm.hip) — the dense products of the training path that are not fused elsewhere
# ------------------------------------------------------------------------------------------------
def gemm(a, b, out=None, accumulate=False, relu_a=False, splits=None):
    """out (+)= opA(a) @ b for 2-D fp32 CUDA VIEWS (transposed / column-sliced views are fine: each operand needs a unit
    stride along one of its axes); opA = relu when relu_a.  `out` (M, N) must have unit column stride.
    splits=None picks a split-K factor that fills the chip when the output has few 128x128 tiles."""
    _require_cuda(a, b, out)
    lib = _lib.load()
    assert a.dim() == 2 and b.dim() == 2 and a.shape[1] == b.shape[0] and a.dtype == b.dtype == torch.float32
    M, K = a.shape
    N = b.shape[1]
    if out is None:
        assert not accumulate
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    assert out.shape == (M, N) and out.dtype == torch.float32 and (N <= 1 or out.stride(1) == 1)
    if 1 not in (a.stride(0), a.stride(1)) and min(a.shape) > 1:
        a = a.contiguous()
    if 1 not in (b.stride(0), b.stride(1)) and min(b.shape) > 1:
        b = b.contiguous()

    def strides(t):          # a size-1 axis may carry any stride: give it the one the kernel wants
        s0, s1 = t.stride()
        if t.shape[1] == 1 and s0 != 1:
            s1 = 1
        if t.shape[0] == 1 and s1 != 1 and s0 != 1:
            s0 = 1
        return s0, s1
    sa_m, sa_k = strides(a)
    sb_k, sb_n = strides(b)
    if splits is None:
        tiles = ((M + 127) // 128) * ((N + 127) // 128)      # (an output of <= 128 rows gets ONE tile row that fits it: nf_gemm.hip)
        splits = max(1, min(512 // max(tiles, 1), K // 256, 64)) if tiles < 128 else 1
    wsp = torch.empty(lib.nf_gemm_f32_workspace_floats(M, N, splits), dtype=torch.float32, device=a.device) if splits > 1 else None
    ldc = out.stride(0) if M > 1 else max(N, out.stride(0))
    check(lib.nf_gemm_f32(M, N, K, ptr(a), sa_m, sa_k, int(relu_a), ptr(b), sb_k, sb_n, ptr(out), ldc, int(accumulate),
                          int(splits), ptr(wsp), _lib.stream()), "nf_gemm_f32")
    return out


# ------------------------------------------------------------------------------------------------
# NeRF weights
# ------------------------------------------------------------------------------------------------
NERF_LAYER_NAMES = [f"xyz_encoding_{i}.0" for i in range(1, 9)] + ["xyz_encoding_final", "dir_encoding.0", "sigma", "rgb.0"]


def feature_dims(enc_flags):
    lib = _lib.load()
    cx, cd, qx, qd = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    lib.nf_render_feature_dims(enc_flags, ctypes.byref(cx), ctypes.byref(cd), ctypes.byref(qx), ctypes.byref(qd))
    return cx.value, cd.value, qx.value, qd.value


def pack_nerf(weights, biases, cx, cd, out=None):
    """weights/biases: 12 contiguous fp32 CUDA tensors in NERF_LAYER_NAMES order -> packed blob."""
    lib = _lib.load()
    n = lib.nf_nerf_packed_floats(cx, cd)
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=weights[0].device)
    P = _lib.NerfParams()
    keep = []
    for i in range(12):
        w = weights[i].detach().contiguous().float()
        b = biases[i].detach().contiguous().float()
        keep += [w, b]
        P.w[i] = w.data_ptr()
        P.b[i] = b.data_ptr()
    check(lib.nf_nerf_pack(ctypes.byref(P), cx, cd, ptr(out), _lib.stream()), "nf_nerf_pack")
    return out


def pack_nerf_n(packed, cx, cd):
    """The tile-per-workgroup kernels' arrangement of a packed blob (nf_nerf_pack_n): nf_nerf_mlp_fwd_n's `packed_n`."""
    lib = _lib.load()
    out = torch.empty_like(packed)
    check(lib.nf_nerf_pack_n(ptr(packed), cx, cd, ptr(out), _lib.stream()), "nf_nerf_pack_n")
    return out


# which LDS-ring kernel serves the fp32 inference passes: "a" = hand-scheduled (nf_mlp_a.hip, generated asm), "l" = the
# compiler-scheduled one (nf_mlp_l.hip).  Bit-identical results; the streams differ (no padding slots in "a").
RING_KERNEL = os.environ.get("NF_RING_KERNEL", "a")


MLP_A_SHAPES = tuple((qx, qd) for qx in (8, 9, 16, 17, 24, 25) for qd in (4, 7))


def pack_nerf_stream(packed, cx, cd, kind=None):
    """Weight stream of the LDS-ring fp32 kernels (nf_nerf_mlp_fwd_a / _l) from the packed blob, or None when there is no ring
    kernel for the feature row (the direct-from-L2 kernel nf_nerf_mlp_fwd then serves the pass).  The tensor carries the kernel
    it was packed for (`nf_kind`)."""
    kind = kind or RING_KERNEL
    shape = ((cx + 7) // 8, (cd + 7) // 8)
    # "a" is instantiated for every feature row the encoding flags of models/renderer.py:30-44 can give; "l" for the default one
    if shape not in (MLP_A_SHAPES if kind == "a" else ((25, 7),)):
        return None
    lib = _lib.load()
    if kind == "a":
        out = torch.empty(lib.nf_nerf_stream_a_floats(cx, cd), dtype=torch.float32, device=packed.device)
        check(lib.nf_nerf_pack_stream_a(ptr(packed), cx, cd, ptr(out), _lib.stream()), "nf_nerf_pack_stream_a")
    else:
        out = torch.empty(lib.nf_nerf_stream_floats(cx, cd), dtype=torch.float32, device=packed.device)
        check(lib.nf_nerf_pack_stream(ptr(packed), cx, cd, ptr(out), _lib.stream()), "nf_nerf_pack_stream")
    out.nf_kind = kind
    return out


def _ring_fwd(lib, wstream):
    """The ring kernel a weight stream was packed for.  The kind travels as an attribute of the tensor, which .to() / .clone() /
    .detach() / pickling drop: a stream without it is refused (the two layouts differ in size and padding; feeding an "a" stream
    to the "l" kernel would read out of bounds) — re-pack with pack_nerf_stream instead of copying a stream."""
    kind = getattr(wstream, "nf_kind", None)
    if kind == "a":
        return lib.nf_nerf_mlp_fwd_a, "nf_nerf_mlp_fwd_a"
    if kind == "l":
        return lib.nf_nerf_mlp_fwd_l, "nf_nerf_mlp_fwd_l"
    raise RuntimeError("weight stream without its kernel kind (nf_kind): it was copied or moved after pack_nerf_stream; pack it again")


# Which kernel serves the fp16-MFMA weight stream: "ha" = the hand-scheduled instruction stream (nf_mlp_ha.hip, generated by
# csrc/gen_mlp_ha.py), "h2" = the compiler-scheduled kernel it restates (nf_mlp_h2.hip; same stream, same X layout, bit-identical results,
# kept as the reference of the equality tests).
FP16_KERNEL = os.environ.get("NF_FP16_KERNEL", "ha")


def _fp16_fwd(lib, kind=None):
    kind = kind or FP16_KERNEL
    if kind == "ha":
        return lib.nf_nerf_mlp_fwd_ha, "nf_nerf_mlp_fwd_ha"
    if kind == "h2":
        return lib.nf_nerf_mlp_fwd_h2, "nf_nerf_mlp_fwd_h2"
    raise ValueError("NF_FP16_KERNEL must be 'ha' or 'h2', not %r" % (kind,))


class PackedH2:
    """Weight streams of the fp16-MFMA MLP: `blob` = nf_nerf_pack_h2's (the operand of nf_nerf_mlp_fwd_h2), `blob_ha` = the same blocks
    without the bias K-steps + the bias table (nf_nerf_pack_ha: the operand of nf_nerf_mlp_fwd_ha, the kernel that runs)."""

    def __init__(self, blob, blob_ha=None):
        self.blob = blob
        self.blob_ha = blob_ha

    def stream(self, kind):
        return self.blob_ha if kind == "ha" else self.blob


def pack_nerf_h2(weights, biases, cx, cd):
    lib = _lib.load()
    out = torch.empty(lib.nf_nerf_packed_h2_bytes(), dtype=torch.uint8, device=weights[0].device)
    P = _lib.NerfParams()
    keep = []
    for i in range(12):
        w = weights[i].detach().contiguous().float()
        b = biases[i].detach().contiguous().float()
        keep += [w, b]
        P.w[i], P.b[i] = w.data_ptr(), b.data_ptr()
    check(lib.nf_nerf_pack_h2(ctypes.byref(P), cx, cd, ptr(out), _lib.stream()), "nf_nerf_pack_h2")
    out_ha = torch.empty(lib.nf_nerf_packed_ha_bytes(), dtype=torch.uint8, device=out.device)
    check(lib.nf_nerf_pack_ha(ptr(out), ptr(out_ha), _lib.stream()), "nf_nerf_pack_ha")
    return PackedH2(out, out_ha)


class PackedS:
    """Weight stream of the split-precision MLP (nf_nerf_pack_s / nf_nerf_mlp_fwd_s)."""

    def __init__(self, blob):
        self.blob = blob


def pack_nerf_s(weights, biases, cx, cd):
    lib = _lib.load()
    out = torch.empty(lib.nf_nerf_packed_s_bytes(), dtype=torch.uint8, device=weights[0].device)
    P = _lib.NerfParams()
    keep = []
    for i in range(12):
        w = weights[i].detach().contiguous().float()
        b = biases[i].detach().contiguous().float()
        keep += [w, b]
        P.w[i], P.b[i] = w.data_ptr(), b.data_ptr()
    check(lib.nf_nerf_pack_s(ctypes.byref(P), cx, cd, ptr(out), _lib.stream()), "nf_nerf_pack_s")
    return PackedS(out)


# ------------------------------------------------------------------------------------------------
# one render pass (coarse or fine) of a ray chunk
# ------------------------------------------------------------------------------------------------
class PassBuffers:
    """Per-pass device buffers; kept so that backward can reuse row lists / activations."""
    pass


def _round_rows(n):
    """Row-buffer capacity: rounded up in geometric steps (<= 12.5 % slack) so that a slowly growing
    active-row count (the fluid spreads frame after frame) re-uses the cached block instead of
    asking the device allocator for a fresh multi-GB one every few frames."""
    n = max(int(n), 1)
    step = max(8192, 1 << max(n.bit_length() - 4, 0))
    return (n + step - 1) // step * step


class Workspace:
    """Grow-only scratch arena for the inference path.  The per-pass scratch (candidate / row lists,
    feature tiles X, rgbsigma) is several GB per 400x400 frame and its size drifts with the scene;
    going through the device allocator for it every pass means an occasional multi-GB hipMalloc in
    the middle of a frame (tens of ms).  Everything here is consumed on the launch stream in order,
    so the coarse and the fine pass (and the next frame) can re-use the same storage."""

    def __init__(self):
        self._buf = {}
        self.row_cap = {}        # (R, S) -> row capacity of a render pass learnt from earlier calls (see render_pass)

    def side_stream(self, device):
        """The second stream of the feature / MLP pipeline (render_pass)."""
        if getattr(self, "_side", None) is None or self._side.device != device:
            self._side = torch.cuda.Stream(device=device)
        return self._side

    def get(self, name, numel, dtype, device):
        nbytes = max(int(numel), 1) * torch.empty(0, dtype=dtype).element_size()
        cur = self._buf.get(name)
        if cur is None or cur.numel() < nbytes or cur.device != device:
            cur = None
            self._buf[name] = None                       # drop the old block before growing
            cur = torch.empty(nbytes + nbytes // 4, dtype=torch.uint8, device=device)
            self._buf[name] = cur
        return cur[:nbytes].view(dtype)[:max(int(numel), 1)]


class HostFetch:
    """A few int32 words read back WITHOUT stalling the launch stream: asynchronous copies into a pinned buffer + an event
    recorded behind them.  The training forward enqueues the fetch of a pass's row count right behind its search kernel —
    i.e. in front of the feature / MLP / composite launches — and waits for the event only when the whole forward is
    enqueued: the count is on the host long before, and the GPU never runs dry behind a mid-pass `.item()`.
    Pinned buffers come from a small per-device ring (allocating pinned memory costs ~1 ms); a slot is reused only after
    its words were read."""
    _ring = {}

    def __init__(self, device):
        key = (device.type, device.index)
        ring = HostFetch._ring.setdefault(key, {"bufs": [torch.empty(16, dtype=torch.int32).pin_memory() for _ in range(8)], "i": 0})
        self.buf = ring["bufs"][ring["i"] % 8]
        ring["i"] += 1
        self.n = 0
        self.ev = None

    def add(self, words):
        """words: 1-D int32 device tensor.  Returns the slice of the result its values will occupy."""
        k = words.numel()
        assert self.n + k <= self.buf.numel() and words.dtype == torch.int32
        self.buf[self.n:self.n + k].copy_(words, non_blocking=True)
        sl = slice(self.n, self.n + k)
        self.n += k
        self.ev = torch.cuda.Event()
        self.ev.record()
        return sl

    def get(self):
        self.ev.synchronize()
        return self.buf[:self.n].tolist()


def render_pass(grid, particles, rays, z, z_table, S, radius, K, enc_flags, use_mask, ro, packed, cx, cd,
                white_bg=True, save_acts=False, max_rows=None, packed_h=None, ws=None, need_weights=True, wstream=None,
                optimistic=False, caps=None, after_search=None, noise=None, packed_n=None, pre_mlp=None):
    """Runs classify -> search -> features -> MLP -> composite for R rays x S samples.
    z: (R,S) per-ray depths or None (then z_table (S,) is shared by all rays).
    Returns a PassBuffers with rgb, depth, opacity, weights, num_nn, mask_sum and the row lists.

    Row-buffer sizing.  The number of active rows is known on the device only.  Exact sizing reads it back (one host
    sync in the middle of the pass).  With optimistic=True and a capacity table (`caps`: (R, S) -> rows; the workspace's for
    inference, the module's for training) that has seen this shape before, the
    pass runs WITHOUT any host round trip against the capacity learnt then (kernels clamp to it); `b.cap` is set and
    the caller must compare `b.n_rows` with it once the frame is enqueued and redo the call on overflow
    (autograd._run_passes does: one sync per call, at its end, instead of one in the middle of each pass)."""
    lib = _lib.load()
    dev = rays.device
    R = rays.shape[0]
    st = _lib.stream()
    n_samp = R * S
    if radius > grid.cell * (1 + 1e-6):
        raise RuntimeError("search radius exceeds the grid cell edge")
    grid.need_firstk()
    b = PassBuffers()
    b.R, b.S = R, S
    if ws is not None and save_acts:
        raise RuntimeError("the scratch arena is for the inference path; backward needs per-pass buffers")

    def scratch(name, numel, dtype):
        if ws is None:
            return torch.empty(numel, dtype=dtype, device=dev)
        return ws.get(name, numel, dtype, dev)

    b.num_nn = torch.empty(n_samp, dtype=torch.int32, device=dev)
    b.K = K         # the mask of a sample is num_nn == K: no separate mask array (nf_composite_* derive the bit from num_nn)
    b.rgbsigma = scratch("rgbsigma", n_samp * 4, torch.float32).view(n_samp, 4)
    cand = scratch("cand", n_samp, torch.int32)
    counters = torch.zeros(2, dtype=torch.int32, device=dev)
    b.counters = counters
    check(lib.nf_render_classify(ptr(grid.ws), ptr(rays), ptr(z), ptr(z_table), R, S, float(radius), int(use_mask), ptr(b.num_nn),
                                 None, ptr(cand), ptr(counters[0:1]), st), "nf_render_classify")
    if max_rows is None:
        max_rows = n_samp
    max_rows = min(max_rows, n_samp)
    b.max_rows = max_rows
    _, _, qx, qd = feature_dims(enc_flags)
    # NOTE: row lists are sized for the worst case of this chunk (every sample active) unless the
    # caller bounds max_rows; the renderer module sizes chunks so this stays modest.
    b.row_sample = scratch("row_sample", n_samp, torch.int32)
    b.row_nbr = scratch("row_nbr", n_samp * K, torch.int32)
    check(lib.nf_render_search(ptr(grid.ws), ptr(rays), ptr(z), ptr(z_table), R, S, float(radius), K, int(use_mask),
                               ptr(cand), ptr(counters[0:1]), ptr(b.num_nn), None,
                               ptr(b.row_sample), ptr(b.row_nbr), ptr(counters[1:2]), st), "nf_render_search")
    b.n_rows = counters[1:2]
    if after_search is not None:
        after_search(b.n_rows)          # e.g. HostFetch.add: the count is final here, everything below is enqueued behind it
    b.cap = None
    alloc_rows = 0
    cap_key = (R, S)
    if caps is None and ws is not None:
        caps = ws.row_cap
    if max_rows >= n_samp and optimistic and caps is not None and cap_key in caps:
        max_rows = min(caps[cap_key], n_samp)        # no sync: verified by the caller at the end of the call
        b.max_rows = b.cap = max_rows
    elif max_rows >= n_samp:
        # exact sizing of the per-row buffers: one host sync per pass (the caller can avoid it by
        # passing a static max_rows bound, e.g. under hipGraph capture)
        max_rows = int(counters[1].item())
        b.max_rows = max_rows
        if caps is not None:
            caps[cap_key] = max(caps.get(cap_key, 0), _round_rows(max_rows + max_rows // 4 + 4096))
            alloc_rows = caps[cap_key]      # size the buffers for the capacity runs that follow (no regrowth in a later call)
    b.n_active = max_rows
    tiles = (max_rows + 31) // 32
    rows_alloc = _round_rows(max(max_rows, alloc_rows))
    x16 = packed_h is not None and not isinstance(packed_h, PackedS)      # the fp16-MFMA MLPs take fp16 operands (half the bytes)
    b.X = scratch("X", rows_alloc // 32 * (qx + qd) * (128 if x16 else 256), torch.float32)
    b.acts = torch.empty(rows_alloc * 2432, dtype=torch.float32, device=dev) if save_acts else None
    b.amask = None          # the ReLU masks as bits (nf_nerf_mlp_fwd_n2 -> nf_nerf_mlp_bwd_n2), training forward of the default feature row
    x_tile = (qx + qd) * (128 if x16 else 256)           # floats of X per 32-row tile
    ro_per_ray = int(ro.dim() == 2)

    def features(row0, nrows_t, mx, stream_):
        check(lib.nf_render_features(ptr(particles), ptr(rays), ptr(z), ptr(z_table), R, S, float(radius), K, enc_flags,
                                     ptr(ro), ro_per_ray, b.row_sample.data_ptr() + 4 * row0, b.row_nbr.data_ptr() + 4 * K * row0,
                                     ptr(nrows_t), mx, b.X.data_ptr() + 4 * x_tile * (row0 // 32), int(x16), stream_),
              "nf_render_features")

    def mlp(row0, nrows_t, mx, stream_):
        X_, rs_ = b.X.data_ptr() + 4 * x_tile * (row0 // 32), b.row_sample.data_ptr() + 4 * row0
        if isinstance(packed_h, PackedS):       # split precision: hi + lo fp16 operands, 3 MFMAs per product (inference only)
            check(lib.nf_nerf_mlp_fwd_s(ptr(packed_h.blob), cx, cd, X_, ptr(nrows_t), mx, rs_, ptr(b.rgbsigma), stream_),
                  "nf_nerf_mlp_fwd_s")
        elif isinstance(packed_h, PackedH2):      # fp16-MFMA, two tiles per wave (inference only); X holds an even tile count
            fn, name = _fp16_fwd(lib)
            check(fn(ptr(packed_h.stream(FP16_KERNEL)), cx, cd, X_, ptr(nrows_t), mx, rs_, ptr(b.rgbsigma), stream_), name)
        elif packed_h is not None:
            raise RuntimeError("unknown fp16 weight stream (expected PackedH2 or PackedS)")
        elif wstream is not None and not save_acts:      # fp32, weight stream shared through LDS (inference)
            fn, name = _ring_fwd(lib, wstream)
            check(fn(ptr(packed), ptr(wstream), cx, cd, X_, ptr(nrows_t), mx, rs_, ptr(b.rgbsigma), stream_), name)
        elif save_acts and (qx, qd) == (25, 7):
            # training forward (activations saved; launches of a few hundred to a few thousand tiles): a tile per WORKGROUP
            # (nf_mlp_n.hip) — a third of the per-tile latency of the tile-per-wave kernel, bit-identical outputs.  (The
            # inference passes stay on the ring kernel at every size: its sums differ in the place of the bias, and results
            # must not depend on how a frame is cut into calls.)
            if packed_n is not None:                        # packed ahead by the caller (a captured step: beside its front-end kernels)
                b.packed_n = packed_n
            if getattr(b, "packed_n", None) is None:        # the tile-per-workgroup kernels' own arrangement of the blob
                b.packed_n = torch.empty_like(packed)
                check(lib.nf_nerf_pack_n(ptr(packed), cx, cd, ptr(b.packed_n), stream_), "nf_nerf_pack_n")
            b.amask = torch.empty(lib.nf_nerf_amask_words(rows_alloc), dtype=torch.int32, device=dev)
            check(lib.nf_nerf_mlp_fwd_n2(ptr(b.packed_n), cx, cd, X_, ptr(nrows_t), mx, rs_, ptr(b.rgbsigma), ptr(b.acts), ptr(b.amask), stream_),
                  "nf_nerf_mlp_fwd_n2")
        else:
            check(lib.nf_nerf_mlp_fwd(ptr(packed), cx, cd, X_, ptr(nrows_t), mx, rs_, ptr(b.rgbsigma), ptr(b.acts), stream_),
                  "nf_nerf_mlp_fwd")

    # Large inference passes are cut into row chunks and pipelined over two streams: the feature stage of chunk c + 1 (VALU /
    # HBM-write work, <= 84 registers per wave, no LDS) runs on a side stream while the MLP of chunk c (one wave per SIMD on
    # the matrix pipe, ~390 of the 512 registers) runs on the launch stream — the two kernels co-reside on the same CUs,
    # so after the first chunk the feature stage costs no wall time.  Chunks are row ranges of the capacity; the kernels
    # clamp to the true row count (one tiny device op derives the per-chunk counts).
    n_chunks = OVERLAP_CHUNKS if (ws is not None and not save_acts and OVERLAP_CHUNKS > 1 and max_rows >= OVERLAP_MIN_ROWS) else 1
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if n_chunks == 1:
        features(0, b.n_rows, max_rows, st)
        if PROFILE is not None:
            e0.record()
        if pre_mlp is not None:
            pre_mlp()               # e.g. the join with a stream that packed the weights while classify / search / features ran
        mlp(0, b.n_rows, max_rows, st)
        if PROFILE is not None:
            e1.record()
            PROFILE["mlp"].append((e0, e1))
    else:
        main = torch.cuda.current_stream()
        side = ws.side_stream(dev)
        chunk = (max_rows + n_chunks - 1) // n_chunks
        chunk = (chunk + 63) // 64 * 64
        starts = torch.arange(n_chunks, dtype=torch.int32, device=dev) * chunk
        counts = (b.n_rows - starts).clamp_(0, chunk)           # rows of every chunk, on the device
        starts.record_stream(side)
        counts.record_stream(side)
        ev = torch.cuda.Event()
        ev.record(main)                                         # row lists (search) and counts are complete
        side.wait_event(ev)
        mlp_ms = []
        for c in range(n_chunks):
            row0 = c * chunk
            mx = min(chunk, max_rows - row0)
            if mx <= 0:
                break
            with torch.cuda.stream(side):
                features(row0, counts[c:c + 1], mx, side.cuda_stream)
                evc = torch.cuda.Event()
                evc.record(side)
            main.wait_event(evc)
            if PROFILE is not None:
                a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a0.record(main)
            mlp(row0, counts[c:c + 1], mx, st)
            if PROFILE is not None:
                a1.record(main)
                mlp_ms.append((a0, a1))
        b.overlap_keep = (starts, counts)
        if PROFILE is not None:
            PROFILE["mlp"].extend(mlp_ms)
            PROFILE["rows"].extend([0] * (len(mlp_ms) - 1))     # the pass's rows are booked once, below
    if PROFILE is not None:
        PROFILE["rows"].append(max_rows if b.cap is None else b.n_rows)      # capacity run: the caller resolves the true count
    b.rgb = torch.empty(R, 3, dtype=torch.float32, device=dev)
    b.depth = torch.empty(R, dtype=torch.float32, device=dev)
    b.opacity = torch.empty(R, dtype=torch.float32, device=dev)
    # the weights are only needed by the pass that is resampled from (the coarse one)
    b.weights = torch.empty(R, S, dtype=torch.float32, device=dev) if need_weights else None
    b.mask_sum = torch.empty(R, dtype=torch.float32, device=dev)
    b.gate = int(bool(use_mask))       # rgbsigma is defined where mask = 1 only (nobody writes the rest)
    b.noise = noise
    if noise is not None:       # noise_std > 0 (models/renderer.py:193-196): (R, S) = noise_std * randn, added to sigma before the ReLU
        check(lib.nf_composite_fwd_noise(ptr(b.rgbsigma), ptr(z), ptr(z_table), ptr(rays), None, b.gate, R, S, int(white_bg), ptr(noise),
                                         ptr(b.rgb), ptr(b.depth), ptr(b.opacity), ptr(b.weights), ptr(b.mask_sum), ptr(b.num_nn), K, st),
              "nf_composite_fwd_noise")
        return b
    check(lib.nf_composite_fwd(ptr(b.rgbsigma), ptr(z), ptr(z_table), ptr(rays), None, b.gate, R, S, int(white_bg),
                               ptr(b.rgb), ptr(b.depth), ptr(b.opacity), ptr(b.weights), ptr(b.mask_sum), ptr(b.num_nn), K, st),
          "nf_composite_fwd")
    return b


def importance_sample(z_table0, weights0, u_table, n_importance, zero_row=None):
    """zero_row: importance_zero_row(...) of the same tables — the shared output row of every ray that hit nothing."""
    lib = _lib.load()
    R, S0 = weights0.shape
    z1 = torch.empty(R, S0 + n_importance, dtype=torch.float32, device=weights0.device)
    check(lib.nf_importance_sample(ptr(z_table0), ptr(weights0), ptr(u_table), R, S0, n_importance, ptr(zero_row), ptr(z1),
                                   _lib.stream()), "nf_importance_sample")
    return z1


def coarse_perturb(z_table, rnd, perturb):
    """coarse_sample_ray's perturb branch (utils/ray_utils.py:247-253): per-ray depths from the shared table and the caller's
    uniform draws rnd (R, S)."""
    lib = _lib.load()
    R, S = rnd.shape
    rnd = rnd.detach().contiguous().float()
    z = torch.empty(R, S, dtype=torch.float32, device=rnd.device)
    check(lib.nf_coarse_perturb(ptr(z_table), ptr(rnd), float(perturb), R, S, ptr(z), _lib.stream()), "nf_coarse_perturb")
    return z


def importance_sample_rays(z0, weights0, u, n_importance):
    """ImportanceSampling(det=False): per-ray coarse depths z0 (R, S0) and per-ray uniform draws u (R, N_imp)."""
    lib = _lib.load()
    R, S0 = weights0.shape
    u = u.detach().contiguous().float()
    assert tuple(z0.shape) == (R, S0) and tuple(u.shape) == (R, n_importance)
    z1 = torch.empty(R, S0 + n_importance, dtype=torch.float32, device=weights0.device)
    check(lib.nf_importance_sample_rays(ptr(z0), ptr(weights0), ptr(u), R, S0, n_importance, ptr(z1), _lib.stream()),
          "nf_importance_sample_rays")
    return z1


def importance_zero_row(z_table0, u_table, n_importance):
    """The resampled depths of a ray with all-zero weights, computed by the general path of the same kernel."""
    w = torch.zeros(1, z_table0.shape[0], dtype=torch.float32, device=z_table0.device)
    return importance_sample(z_table0, w, u_table, n_importance)[0].contiguous()


# ------------------------------------------------------------------------------------------------
# layout helpers (host-side plumbing for unit tests and the standalone NeRF.forward)
# ------------------------------------------------------------------------------------------------
def rows_to_tiles(x, cx, cd):
    """row-major features (n, cx+cd) -> MLP operand layout X[tile][q][h*32+j][4] (see nf_render_features)."""
    n = x.shape[0]
    qx, qd = (cx + 7) // 8, (cd + 7) // 8
    npad = (n + 31) // 32 * 32
    f = torch.zeros(npad, 8 * (qx + qd), dtype=torch.float32, device=x.device)
    f[:n, :cx] = x[:, :cx]
    f[:n, 8 * qx:8 * qx + cd] = x[:, cx:cx + cd]
    return f.view(npad // 32, 32, qx + qd, 2, 4).permute(0, 2, 3, 1, 4).contiguous().view(-1)


def tiles_to_rows(X, n, cx, cd):
    qx, qd = (cx + 7) // 8, (cd + 7) // 8
    T = (n + 31) // 32
    f = X[:T * (qx + qd) * 256].view(T, qx + qd, 2, 32, 4).permute(0, 3, 1, 2, 4).reshape(T * 32, 8 * (qx + qd))
    return torch.cat([f[:n, :cx], f[:n, 8 * qx:8 * qx + cd]], 1)


def mlp_rows(packed, cx, cd, x, save_acts=False, packed_h=None, wstream=None):
    """NeRF.forward on row-major features x (n, cx+cd) through the MFMA kernel -> (n,4) [rgb, sigma]."""
    lib = _lib.load()
    n = x.shape[0]
    X = rows_to_tiles(x.detach().float(), cx, cd)
    n_rows = torch.tensor([n], dtype=torch.int32, device=x.device)
    row_sample = torch.arange(n, dtype=torch.int32, device=x.device)
    out = torch.zeros(n, 4, dtype=torch.float32, device=x.device)
    acts = torch.empty(n * 2432, dtype=torch.float32, device=x.device) if save_acts else None
    if isinstance(packed_h, PackedS):
        check(lib.nf_nerf_mlp_fwd_s(ptr(packed_h.blob), cx, cd, ptr(X), ptr(n_rows), n, ptr(row_sample), ptr(out), _lib.stream()),
              "nf_nerf_mlp_fwd_s")
        return out
    if packed_h is not None:
        # fp32 operand layout [tile][q][lane][4] -> fp16 layout [tile][t][lane][8]: K-step t = groups 2t, 2t+1
        T = X.numel() // ((qx_ := (cx + 7) // 8) + (qd_ := (cd + 7) // 8)) // 256
        Xh = X.view(T, (qx_ + qd_) // 2, 2, 64, 4).permute(0, 1, 3, 2, 4).reshape(-1).to(torch.float16).contiguous()
        if isinstance(packed_h, PackedH2):
            if T % 2:       # the two-tiles-per-wave kernel reads whole tile pairs
                Xh = torch.cat([Xh, torch.zeros(Xh.numel() // T, dtype=Xh.dtype, device=Xh.device)])
            fn, name = _fp16_fwd(lib)
            check(fn(ptr(packed_h.stream(FP16_KERNEL)), cx, cd, ptr(Xh), ptr(n_rows), n, ptr(row_sample), ptr(out), _lib.stream()), name)
            return out
        raise RuntimeError("unknown fp16 weight stream (expected PackedH2 or PackedS)")
    if wstream is not None:
        fn, name = _ring_fwd(lib, wstream)
        check(fn(ptr(packed), ptr(wstream), cx, cd, ptr(X), ptr(n_rows), n, ptr(row_sample), ptr(out), _lib.stream()), name)
        return out
    check(lib.nf_nerf_mlp_fwd(ptr(packed), cx, cd, ptr(X), ptr(n_rows), n, ptr(row_sample), ptr(out), ptr(acts),
                              _lib.stream()), "nf_nerf_mlp_fwd")
    return (out, acts.view(n, 2432)) if save_acts else out


def debug_features(particles, rays, near, far, S, radius, K, enc_flags, ro):
    """Coarse-pass classify + search + features only; returns row-major features of the active rows."""
    lib = _lib.load()
    dev = rays.device
    t = torch.linspace(0, 1, S)
    z_table = (near * (1 - t) + far * t).to(dev)
    grid = build_grid(particles, radius)
    R = rays.shape[0]
    n = R * S
    num_nn = torch.empty(n, dtype=torch.int32, device=dev)
    cand = torch.empty(n, dtype=torch.int32, device=dev)
    counters = torch.zeros(2, dtype=torch.int32, device=dev)
    row_sample = torch.empty(n, dtype=torch.int32, device=dev)
    row_nbr = torch.empty(n * K, dtype=torch.int32, device=dev)
    st = _lib.stream()
    rays = rays.contiguous().float()
    check(lib.nf_render_classify(ptr(grid.ws), ptr(rays), None, ptr(z_table), R, S, float(radius), 1, ptr(num_nn), None,
                                 ptr(cand), ptr(counters[0:1]), st))
    check(lib.nf_render_search(ptr(grid.ws), ptr(rays), None, ptr(z_table), R, S, float(radius), K, 1, ptr(cand),
                               ptr(counters[0:1]), ptr(num_nn), None, ptr(row_sample), ptr(row_nbr),
                               ptr(counters[1:2]), st))
    nr = int(counters[1].item())
    cx, cd, qx, qd = feature_dims(enc_flags)
    X = torch.empty(max((nr + 31) // 32, 1) * (qx + qd) * 256, device=dev)
    check(lib.nf_render_features(ptr(grid.points), ptr(rays), None, ptr(z_table), R, S, float(radius), K, enc_flags,
                                 ptr(ro.contiguous().float()), 0, ptr(row_sample), ptr(row_nbr), ptr(counters[1:2]), nr,
                                 ptr(X), 0, st))
    return dict(features=tiles_to_rows(X, nr, cx, cd), row_sample=row_sample[:nr], num_nn=num_nn,
                row_nbr=row_nbr[:nr * K].view(nr, K))
