// nf_common.h — shared declarations of libneurofluid_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/neurofluid_hip.h"

#define NF_WAVE 64

void nf_set_error(const char* fmt, ...);

#define NF_CHECK_ARG(cond, msg)                                   \
    do {                                                          \
        if (!(cond)) { nf_set_error("%s: %s", __func__, msg); return NF_EINVAL; } \
    } while (0)

// true the first time a call site runs on the CURRENT device (hipFuncSetAttribute belongs to a device's copy of the function:
// a process-wide "done" flag would leave every device but the first at the 64 KB default).  `flags` = the call site's own
// static bool[64].  The flags are the library's only process-wide state besides the thread-local error string: they are read and
// written atomically, and a lost race only repeats an idempotent hipFuncSetAttribute, so concurrent callers are safe.
static inline bool nf_first_use_on_device(bool* flags)
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return true;
    if (__atomic_load_n(&flags[d], __ATOMIC_ACQUIRE)) return false;
    __atomic_store_n(&flags[d], true, __ATOMIC_RELEASE);
    return true;
}

#define NF_CHECK_LAUNCH()                                                             \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) { nf_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_)); return NF_ELAUNCH; } \
    } while (0)

// ---------------------------------------------------------------------------------------------
// Grid workspace layout (device memory, caller-owned):
//   [NfGridHeader | cell_start (n_cells+1) | cell_dil (n_cells) | sorted_idx (n) | sorted_pos (n float4)
//    | tmp_cell (n) | tmp_list (n) | cell_fill (n_cells)]
// ---------------------------------------------------------------------------------------------
struct NfGridHeader {
    float origin[3];
    float inv_cell[3];
    int dims[3];
    int n_points;
    int n_cells;
    int off_cell_start;  // byte offsets from the start of the workspace
    int off_cell_dil;
    int off_sorted_idx;
    int off_sorted_pos;
    int off_tmp_cell;
    int off_tmp_list;
    int off_cell_fill;
    int off_cell_aabb;   // float[6] per cell: min xyz, max xyz of the contained points (+inf/-inf when empty)
    int off_cell_rec;    // 3 x float4 per cell: {start, end, min original index, -} {lo.xyz, hi.x} {hi.y, hi.z, -, -}
    int off_dil_start;   // int[n_cells+1]: start of the cell's DILATED list (all points of its 27-neighbourhood)
    int off_dil_pos;     // float4[sum cell_dil] (<= 27 n): xyz + index bits, ascending ORIGINAL index inside each list
    int off_dil_box;     // 2 x float4 per 16 consecutive dil_pos entries: {lo.xyz, -} {hi.xyz, -} (chunk AABB)
    unsigned pt_lo[3], pt_hi[3];   // exact AABB of the points (order-preserving uint encoding; device-written at build)
    // The cell lists cover the SUB-BOX [sub0, sub0 + subd) of the grid's cells (cell coordinates are always computed on the full
    // grid: origin / inv_cell / dims): cell_start is indexed ((z - sub0z) * subd_y + (y - sub0y)) * subd_x + (x - sub0x) and holds
    // subd_x * subd_y * subd_z + 1 entries.  Ordinary builds: sub0 = 0, subd = dims.  The transition step's build
    // (nf_trans.hip:k_trans_stage1) bins on the static container grid and lists only the cells the cloud occupies.
    int sub0[3], subd[3];
};

// order-preserving float <-> uint (atomicMin / atomicMax on floats of either sign)
__device__ __forceinline__ unsigned nf_f2ord(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float nf_ord2f(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

struct NfGridView {
    float ox, oy, oz;
    float icx, icy, icz;
    int dx, dy, dz;
    int s0x, s0y, s0z, sdx, sdy, sdz;      // sub-box of listed cells (NfGridHeader::sub0 / subd)
    int n_points;
    const int* cell_start;
    const int* cell_dil;
    const int* sorted_idx;
    const float4* sorted_pos;
    const float* cell_aabb;
    const float4* cell_rec;
    const int* dil_start;
    const float4* dil_pos;
    const float4* dil_box;
    float plo[3], phi[3];   // exact AABB of the points (+inf / -inf when the grid is empty)
};

__device__ __forceinline__ NfGridView nf_grid_view(const void* ws)
{
    const NfGridHeader* h = (const NfGridHeader*)ws;
    const char* b = (const char*)ws;
    NfGridView v;
    v.ox = h->origin[0]; v.oy = h->origin[1]; v.oz = h->origin[2];
    v.icx = h->inv_cell[0]; v.icy = h->inv_cell[1]; v.icz = h->inv_cell[2];
    v.dx = h->dims[0]; v.dy = h->dims[1]; v.dz = h->dims[2];
    v.s0x = h->sub0[0]; v.s0y = h->sub0[1]; v.s0z = h->sub0[2];
    v.sdx = h->subd[0]; v.sdy = h->subd[1]; v.sdz = h->subd[2];
    v.n_points = h->n_points;
    v.cell_start = (const int*)(b + h->off_cell_start);
    v.cell_dil = (const int*)(b + h->off_cell_dil);
    v.sorted_idx = (const int*)(b + h->off_sorted_idx);
    v.sorted_pos = (const float4*)(b + h->off_sorted_pos);
    v.cell_aabb = (const float*)(b + h->off_cell_aabb);
    v.cell_rec = (const float4*)(b + h->off_cell_rec);
    v.dil_start = (const int*)(b + h->off_dil_start);
    v.dil_pos = (const float4*)(b + h->off_dil_pos);
    v.dil_box = (const float4*)(b + h->off_dil_box);
    for (int d = 0; d < 3; ++d) { v.plo[d] = nf_ord2f(h->pt_lo[d]); v.phi[d] = nf_ord2f(h->pt_hi[d]); }
    return v;
}

__device__ __forceinline__ int nf_cell_coord(float p, float o, float ic, int d)
{
    int c = (int)floorf((p - o) * ic);
    return c < 0 ? 0 : (c >= d ? d - 1 : c);
}

// Sum over a wave on the DPP path: four rotations inside the 16-lane rows, then the four rows through SGPRs — no LDS traffic (a
// __shfl_xor butterfly is six ds_bpermute round trips through the LDS crossbar).  Uniform result.
__device__ __forceinline__ float nf_wave_sum(float v)
{
#define NF_ROR_ADD(n) v += __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x120 | (n), 0xf, 0xf, false));
    NF_ROR_ADD(1) NF_ROR_ADD(2) NF_ROR_ADD(4) NF_ROR_ADD(8)
#undef NF_ROR_ADD
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}

// fp32 squared distance exactly as the oracle (oracle/csrc/nf_oracle.c:d2f): mul + add chain, d = 0,1,2,
// no FMA contraction.
__device__ __forceinline__ float nf_dist2(float qx, float qy, float qz, float px, float py, float pz)
{
    float dx = __fsub_rn(qx, px), dy = __fsub_rn(qy, py), dz = __fsub_rn(qz, pz);
    float s = __fmul_rn(dx, dx);
    s = __fadd_rn(s, __fmul_rn(dy, dy));
    s = __fadd_rn(s, __fmul_rn(dz, dz));
    return s;
}

// sample position o + d*z with separate mul / add (torch: rays_o + rays_d * z)
__device__ __forceinline__ float nf_madd_nofma(float o, float d, float z) { return __fadd_rn(o, __fmul_rn(d, z)); }

// ------------------------------------------------------------------------------------------------
// first-K-by-index search core (used by nf_grid.hip: ball query op, nf_render.hip: fused search)
// ------------------------------------------------------------------------------------------------
#define BQ_BLOCK 128
#define BQ_LDS_INTS(K) ((K) * BQ_BLOCK)   // K neighbour indices per thread, [k][thread] (conflict-free)

// fp32 squared distance from a query to a cell's particle AABB, same op order as nf_dist2.  Because
// fp32 sub/mul/add are monotone, box_d2 <= nf_dist2(query, p) for every particle p of the cell, so
// "box_d2 < r2" never rejects a cell that holds an in-radius particle.
__device__ __forceinline__ float nf_box_dist2(const float* __restrict__ bb, float qx, float qy, float qz)
{
    float dx = fmaxf(fmaxf(__fsub_rn(bb[0], qx), __fsub_rn(qx, bb[3])), 0.f);
    float dy = fmaxf(fmaxf(__fsub_rn(bb[1], qy), __fsub_rn(qy, bb[4])), 0.f);
    float dz = fmaxf(fmaxf(__fsub_rn(bb[2], qz), __fsub_rn(qz, bb[5])), 0.f);
    float s = __fmul_rn(dx, dx);
    s = __fadd_rn(s, __fmul_rn(dy, dy));
    s = __fadd_rn(s, __fmul_rn(dz, dz));
    return s;
}

// Cheap exact rejection ahead of any table lookup: a point within `radius` of some particle lies inside the
// particles' AABB grown by the radius (margin: 1e-4 relative, far above the fp32 rounding of the distance test).
// Most samples of an image lie outside — and clamp into a (non-empty) border cell, the expensive case below.
__device__ __forceinline__ bool nf_near_points_aabb(const NfGridView& g, float qx, float qy, float qz, float radius)
{
    const float rm = radius * 1.0001f + 1e-6f;
    return qx >= g.plo[0] - rm && qx <= g.phi[0] + rm && qy >= g.plo[1] - rm && qy <= g.phi[1] + rm &&
           qz >= g.plo[2] - rm && qz <= g.phi[2] + rm;
}

// true iff some cell of the 27-neighbourhood has its particle AABB within (strictly) radius
__device__ __forceinline__ bool nf_any_cell_in_reach(const NfGridView& g, float qx, float qy, float qz, float r2)
{
    int cx = nf_cell_coord(qx, g.ox, g.icx, g.dx);
    int cy = nf_cell_coord(qy, g.oy, g.icy, g.dy);
    int cz = nf_cell_coord(qz, g.oz, g.icz, g.dz);
    if (g.cell_dil[(cz * g.dy + cy) * g.dx + cx] == 0) return false;
    for (int z = max(cz - 1, 0); z <= min(cz + 1, g.dz - 1); ++z)
        for (int y = max(cy - 1, 0); y <= min(cy + 1, g.dy - 1); ++y)
            for (int x = max(cx - 1, 0); x <= min(cx + 1, g.dx - 1); ++x) {
                int c = (z * g.dy + y) * g.dx + x;     // empty cells carry a +inf/-inf box: distance = inf
                if (nf_box_dist2(g.cell_aabb + 6 * c, qx, qy, qz) < r2) return true;
            }
    return false;
}

// First-K-by-index search.  Every cell owns a DILATED list: all points of its 27-cell neighbourhood, merged in
// ascending original index at grid-build time.  A query therefore walks ONE list in exactly the reference's scan
// order (pytorch3d scans p2 in index order) restricted to the points that can be in range, appends hits (they
// arrive sorted: no insertion, no per-cell bookkeeping) and stops at the K-th.  The list is walked in aligned
// chunks of 16 entries, each with a precomputed AABB: a chunk whose box is not within the radius cannot hold a hit
// (nf_box_dist2 never exceeds the distance to a contained point) and is skipped without touching its entries —
// consecutive indices are spatially coherent in practice (lattice fill order, SPH emitters), so a ball that covers
// ~15 % of the 27-cell volume rejects most chunks.  Candidates are fetched 4 at a time (independent loads in flight).
// li = K indices [k * BQ_BLOCK + tid]; nzmask bit k = (d2 of slot k != 0).  K <= 32.
#define NF_DIL_CHUNK 16
__device__ __forceinline__ int firstk_search(const NfGridView& g, float qx, float qy, float qz, float r2, int K,
                                             int* li, int tid, unsigned& nzmask)
{
    int cx = nf_cell_coord(qx, g.ox, g.icx, g.dx);
    int cy = nf_cell_coord(qy, g.oy, g.icy, g.dy);
    int cz = nf_cell_coord(qz, g.oz, g.icz, g.dz);
    const int cell = (cz * g.dy + cy) * g.dx + cx;
    const int s = g.dil_start[cell], e = g.dil_start[cell + 1];
    int cnt = 0;
    nzmask = 0u;
    for (int cb = s & ~(NF_DIL_CHUNK - 1); cb < e; cb += NF_DIL_CHUNK) {
        const float4 blo = g.dil_box[2 * (cb / NF_DIL_CHUNK)], bhi = g.dil_box[2 * (cb / NF_DIL_CHUNK) + 1];
        const float bb[6] = {blo.x, blo.y, blo.z, bhi.x, bhi.y, bhi.z};
        if (!(nf_box_dist2(bb, qx, qy, qz) < r2)) continue;
        const int t1 = min(cb + NF_DIL_CHUNK, e);
        for (int t = max(cb, s); t < t1; t += 4) {
            float4 p[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) p[u] = g.dil_pos[min(t + u, t1 - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (t + u < t1 && cnt < K) {
                    float d2 = nf_dist2(qx, qy, qz, p[u].x, p[u].y, p[u].z);
                    if (d2 < r2) {
                        li[cnt * BQ_BLOCK + tid] = __float_as_int(p[u].w);
                        nzmask |= (d2 != 0.f ? 1u : 0u) << cnt;
                        ++cnt;
                    }
                }
            }
        }
        if (cnt >= K) break;
    }
    return cnt;
}

// host-side helper shared by the grid functions
int nf_grid_make_header(int n_points, float cell, const float bbox[6], NfGridHeader* h, size_t* total_bytes);
// the first two launches of nf_trans_step (nf_trans.hip): grid build beside the container half, then the fluid half of the front
int nf_trans_stage12(const nf_trans_step_t* s, const float* pos, const float* vel, float* num_nbrs, int32_t* host_flag3, int step_id,
                     nf_stream_t stream);
