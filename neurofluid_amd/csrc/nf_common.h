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
    int pad_[3];
};

struct NfGridView {
    float ox, oy, oz;
    float icx, icy, icz;
    int dx, dy, dz;
    int n_points;
    const int* cell_start;
    const int* cell_dil;
    const int* sorted_idx;
    const float4* sorted_pos;
    const float* cell_aabb;
    const float4* cell_rec;
};

__device__ __forceinline__ NfGridView nf_grid_view(const void* ws)
{
    const NfGridHeader* h = (const NfGridHeader*)ws;
    const char* b = (const char*)ws;
    NfGridView v;
    v.ox = h->origin[0]; v.oy = h->origin[1]; v.oz = h->origin[2];
    v.icx = h->inv_cell[0]; v.icy = h->inv_cell[1]; v.icz = h->inv_cell[2];
    v.dx = h->dims[0]; v.dy = h->dims[1]; v.dz = h->dims[2];
    v.n_points = h->n_points;
    v.cell_start = (const int*)(b + h->off_cell_start);
    v.cell_dil = (const int*)(b + h->off_cell_dil);
    v.sorted_idx = (const int*)(b + h->off_sorted_idx);
    v.sorted_pos = (const float4*)(b + h->off_sorted_pos);
    v.cell_aabb = (const float*)(b + h->off_cell_aabb);
    v.cell_rec = (const float4*)(b + h->off_cell_rec);
    return v;
}

__device__ __forceinline__ int nf_cell_coord(float p, float o, float ic, int d)
{
    int c = (int)floorf((p - o) * ic);
    return c < 0 ? 0 : (c >= d ? d - 1 : c);
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
#define BQ_LDS_INTS(K) (((K) + 27) * BQ_BLOCK)   // K indices + 27 cell keys per thread (squared distances are not
                                                  // kept: only "d2 != 0" per slot, as a bit mask in a register)

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

// Sorted insertion of index j into the per-thread ascending list held in LDS as list[k * BQ_BLOCK + tid];
// `nzmask` bit k = (squared distance of slot k != 0) travels with the slots.  Returns the new count.
__device__ __forceinline__ int firstk_insert(int* li, unsigned& nzmask, int cnt, int K, int j, bool nz, int tid)
{
    int pos = cnt < K ? cnt : K - 1;  // slot that is overwritten / appended
    while (pos > 0 && li[(pos - 1) * BQ_BLOCK + tid] > j) {
        li[pos * BQ_BLOCK + tid] = li[(pos - 1) * BQ_BLOCK + tid];
        --pos;
    }
    li[pos * BQ_BLOCK + tid] = j;
    // bits [pos, K-1) shift up by one, bit pos takes `nz`, bits >= K are dropped
    const unsigned low = nzmask & ((1u << pos) - 1u);
    const unsigned high = (nzmask >> pos) << (pos + 1);
    nzmask = (low | high | ((nz ? 1u : 0u) << pos)) & ((K >= 32) ? 0xffffffffu : ((1u << K) - 1u));
    return cnt < K ? cnt + 1 : K;
}

// First-K-by-index search.  Cells are index-sorted, so a cell's first entry is its minimum index.
// Cells are visited in ascending order of that minimum (keys in LDS); once the list is full and the
// smallest remaining key exceeds the current K-th index, no remaining particle can enter the list.
// Candidates are fetched 4 at a time (independent loads in flight) — entries past the break point can
// never enter the list, so testing them is harmless.  lk = 27 keys [c * BQ_BLOCK + tid].  K <= 32.
__device__ __forceinline__ int firstk_search(const NfGridView& g, float qx, float qy, float qz, float r2_, int K,
                                             int* li, int* lk, int tid, unsigned& nzmask)
{
    const float r2 = r2_;
    int cx = nf_cell_coord(qx, g.ox, g.icx, g.dx);
    int cy = nf_cell_coord(qy, g.oy, g.icy, g.dy);
    int cz = nf_cell_coord(qz, g.oz, g.icz, g.dz);
    const int BIG = 0x7fffffff;
    int nvalid = 0;
    nzmask = 0u;
    for (int c = 0; c < 27; ++c) {
        int x = cx + (c % 3) - 1, y = cy + ((c / 3) % 3) - 1, z = cz + (c / 9) - 1;
        int key = BIG;
        if (x >= 0 && x < g.dx && y >= 0 && y < g.dy && z >= 0 && z < g.dz) {
            const float4* rec = g.cell_rec + 3 * ((z * g.dy + y) * g.dx + x);   // one 48-byte record per cell
            const float4 r0 = rec[0];
            if (__float_as_int(r0.y) > __float_as_int(r0.x)) {
                const float4 r1 = rec[1], r2 = rec[2];
                const float bb[6] = {r1.x, r1.y, r1.z, r1.w, r2.x, r2.y};
                if (nf_box_dist2(bb, qx, qy, qz) < r2_) { key = __float_as_int(r0.z); ++nvalid; }
            }
        }
        lk[c * BQ_BLOCK + tid] = key;
    }
    int cnt = 0;
    while (nvalid > 0) {
        int best = BIG, bc = 0;
        for (int c = 0; c < 27; ++c) {
            int k = lk[c * BQ_BLOCK + tid];
            if (k < best) { best = k; bc = c; }
        }
        if (best == BIG) break;
        if (cnt == K && best > li[(K - 1) * BQ_BLOCK + tid]) break;
        lk[bc * BQ_BLOCK + tid] = BIG;
        --nvalid;
        int x = cx + (bc % 3) - 1, y = cy + ((bc / 3) % 3) - 1, z = cz + (bc / 9) - 1;
        const float4 r0 = g.cell_rec[3 * ((z * g.dy + y) * g.dx + x)];
        int s = __float_as_int(r0.x), e = __float_as_int(r0.y);
        bool done = false;
        for (int t = s; t < e && !done; t += 4) {
            float4 p[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) p[u] = g.sorted_pos[min(t + u, e - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (t + u >= e) break;
                int j = __float_as_int(p[u].w);
                if (cnt == K && j > li[(K - 1) * BQ_BLOCK + tid]) { done = true; break; }  // cell is index-sorted
                float d2 = nf_dist2(qx, qy, qz, p[u].x, p[u].y, p[u].z);
                if (d2 < r2) cnt = firstk_insert(li, nzmask, cnt, K, j, d2 != 0.f, tid);
            }
        }
    }
    return cnt;
}

// host-side helper shared by the grid functions
int nf_grid_make_header(int n_points, float cell, const float bbox[6], NfGridHeader* h, size_t* total_bytes);
