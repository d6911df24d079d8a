"""ctypes front-end of oracle/csrc/nf_oracle.c (TEST INFRASTRUCTURE — see oracle/__init__.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libnf_oracle.so")
_lib = None


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
            os.path.join(_HERE, "csrc", "nf_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B", "_build/libnf_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.nfo_radius_count.restype = ctypes.c_int64
        # libgomp would start one thread per VISIBLE core (256 on the GPU boxes, under a 16-CPU quota): cap it
        _lib.nfo_set_threads(min(len(os.sched_getaffinity(0)), 16))
    return _lib


def set_threads(n):
    """OpenMP threads of the C oracle (its query loops are parallel; results do not depend on the count)."""
    lib = _load()
    lib.nfo_set_threads(int(n))
    return int(lib.nfo_max_threads())


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def ball_query_firstk(p1, p2, radius, K):
    """pytorch3d.ops.ball_query restatement for ONE cloud.
    p1 (Q,3) queries, p2 (P,3) points -> dists2 (Q,K) f32, idx (Q,K) i64, nn (Q,K,3) f32.
    Reference call site: /root/reference/models/renderer.py:112-122."""
    lib = _load()
    p1 = np.ascontiguousarray(p1, dtype=np.float32).reshape(-1, 3)
    p2 = np.ascontiguousarray(p2, dtype=np.float32).reshape(-1, 3)
    Q = p1.shape[0]
    d = np.empty((Q, K), np.float32)
    i = np.empty((Q, K), np.int64)
    nn = np.empty((Q, K, 3), np.float32)
    lib.nfo_ball_query_firstk(_p(p1), ctypes.c_int64(Q), _p(p2), ctypes.c_int64(p2.shape[0]),
                              ctypes.c_float(radius), ctypes.c_int(K), _p(d), _p(i), _p(nn))
    return d, i, nn


def fixed_radius_search(points, queries, radius, ignore_query_point=True):
    """Open3D FixedRadiusSearch restatement -> (neighbors_index i32 (nnz), row_splits i64 (Q+1),
    neighbors_distance f32 (nnz) = squared L2)."""
    lib = _load()
    pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
    q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, 3)
    Q = q.shape[0]
    counts = np.empty((Q,), np.int64)
    tot = lib.nfo_radius_count(_p(q), ctypes.c_int64(Q), _p(pts), ctypes.c_int64(pts.shape[0]),
                               ctypes.c_float(radius), ctypes.c_int(int(ignore_query_point)), _p(counts))
    rs = np.zeros((Q + 1,), np.int64)
    np.cumsum(counts, out=rs[1:])
    idx = np.empty((tot,), np.int32)
    d2 = np.empty((tot,), np.float32)
    lib.nfo_radius_csr(_p(q), ctypes.c_int64(Q), _p(pts), ctypes.c_int64(pts.shape[0]),
                       ctypes.c_float(radius), ctypes.c_int(int(ignore_query_point)), _p(rs), _p(idx), _p(d2))
    return idx, rs, d2


def ball_query_contraction_sensitivity(p1, p2, radius, K, inclusive=False):
    """How many decisions of the ball query (strict d2 < r2, first K by index) change when `dx^2 + dy^2 + dz^2` is contracted to
    FMAs (nvcc's default for pytorch3d's kernel) instead of evaluated as mul + add (this oracle, the HIP kernel).
    inclusive=True: the same question for Open3D's FixedRadiusSearch (d2 <= r2; pass K above any neighbour count).
    -> dict(pairs, flipped_pairs, queries_with_a_different_list, in_ball_pairs_with_another_d2)."""
    lib = _load()
    p1 = np.ascontiguousarray(p1, dtype=np.float32).reshape(-1, 3)
    p2 = np.ascontiguousarray(p2, dtype=np.float32).reshape(-1, 3)
    out = np.zeros((4,), np.int64)
    lib.nfo_ball_query_contraction_sensitivity(_p(p1), ctypes.c_int64(p1.shape[0]), _p(p2), ctypes.c_int64(p2.shape[0]),
                                               ctypes.c_float(radius), ctypes.c_int(K), ctypes.c_int(int(inclusive)), _p(out))
    return dict(pairs=int(out[0]), flipped_pairs=int(out[1]), queries_with_a_different_list=int(out[2]),
                in_ball_pairs_with_another_d2=int(out[3]))
