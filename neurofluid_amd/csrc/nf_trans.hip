// nf_trans.hip — front half of the transition step (ParticleNet.forward, models/transmodel.py:151-163) for the inference path:
//
//   nf_trans_prepare   ONE workgroup: gravity integration (B1, :100-104) + the fluid cell grid of the integrated
//                      positions (counting sort in LDS, stable in original index) — replaces k_trans_integrate and the
//                      six k_grid_* / scan launches of nf_grid_build(with_firstk_lists = 0)
//   nf_trans_front     fixed-radius search of both clouds, the row-entry lists of the fluid pairs (what the G-free
//                      convolutions of nf_cconv_gf.hip consume) and layer 0 (conv0_obstacle, conv0_fluid, dense0_fluid)
//   nf_trans_stage12   (C++ linkage: nf_trans_step's first two launches)  the same two stages as the fused step runs them.
//                      Clouds up to nf_trans_all_pairs_max_points() particles: k_trans_stage1b (integration + an ALL-PAIRS
//                      search, no grid) -> k_trans_front_rows (fluid half from the rows; the container half beside it);
//                      larger clouds: k_trans_stage1 (grid build beside the container half) -> k_trans_front (grid walk).
// Neighbour rows have a fixed PITCH (capacity per particle): no offsets to compute, no host round trip inside the step.
// The true counts stay on the device; a count above its pitch is reported through pinned host words, and the host redoes
// THAT step on the exact CSR path before ParticleNet.forward returns (neurofluid_amd/transmodel.py).
#include "nf_common.h"
#include <math.h>
#include <string.h>

#define TP_BLOCK 1024
#define TP_MAX_PER_THREAD 16            // n <= 16 384 particles
#define TP_MAX_LDS_INTS 39936           // cell counters + the scatter list, 156 KB of LDS: n_cells + n_points <= this
#define TS_MAX_LDS_INTS 38911           // the same for k_trans_stage1 (152 KB dynamic next to 5.3 KB of static LDS)

extern "C" int nf_trans_prepare_limits(int* max_points, int* max_cells)
{
    if (max_points) *max_points = TP_BLOCK * TP_MAX_PER_THREAD;
    if (max_cells) *max_cells = TS_MAX_LDS_INTS;     /* n_cells + n_points must not exceed this */
    return NF_OK;
}

__global__ void __launch_bounds__(TP_BLOCK) k_trans_prepare(NfGridHeader h, void* __restrict__ ws, const float* __restrict__ pos,
                                                            const float* __restrict__ vel, float gx, float gy, float gz, float dt,
                                                            float cell, float* __restrict__ pos_new, float* __restrict__ vel_new,
                                                            float* __restrict__ feats4)
{
    extern __shared__ int cells[];          // n_cells counters -> starts -> ends, then the scatter list (n_points)
    __shared__ int s_scan[TP_BLOCK / 64];
    __shared__ float s_red[TP_BLOCK / 64][6];
    __shared__ NfGridHeader hh;
    char* b = (char*)ws;
    const int n = h.n_points, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // ---- integrate; the exact bounds of the integrated cloud
    const float g[3] = {gx, gy, gz};
    float px[TP_MAX_PER_THREAD], py[TP_MAX_PER_THREAD], pz[TP_MAX_PER_THREAD];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int u = 0; u < TP_MAX_PER_THREAD; ++u) {
        const int i = u * TP_BLOCK + tid;
        px[u] = py[u] = pz[u] = 0.f;
        if (i < n) {
            float pn[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float v = vel[3 * i + d];
                const float vn = v + g[d] * dt;                       // same expressions as k_trans_integrate
                pn[d] = pos[3 * i + d] + (v + vn) / 2 * dt;
                pos_new[3 * i + d] = pn[d];
                vel_new[3 * i + d] = vn;
                feats4[4 * i + 1 + d] = vn;
                lo[d] = fminf(lo[d], pn[d]); hi[d] = fmaxf(hi[d], pn[d]);
            }
            feats4[4 * i] = 1.f;
            px[u] = pn[0]; py[u] = pn[1]; pz[u] = pn[2];
        }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { lo[d] = fminf(lo[d], __shfl_xor(lo[d], o, 64)); hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o, 64)); }
    if (lane == 0) { for (int d = 0; d < 3; ++d) { s_red[wv][d] = lo[d]; s_red[wv][3 + d] = hi[d]; } }
    __syncthreads();
    if (tid == 0) {
        // The grid of THIS step hugs the cloud: the caller's bbox (the container, static: no host round trip) only bounds it —
        // a few hundred cells instead of the container's ~29 000, which every pass below (zero, scan, store) walks.  Points
        // outside the bbox land in border cells (the search stays exact, include/neurofluid_hip.h); the workspace offsets are
        // those of the caller's header, computed for the larger grid.
        hh = h;
        int ncell = 1;
        for (int d = 0; d < 3; ++d) {
            float l = INFINITY, u2 = -INFINITY;
            for (int w2 = 0; w2 < TP_BLOCK / 64; ++w2) { l = fminf(l, s_red[w2][d]); u2 = fmaxf(u2, s_red[w2][3 + d]); }
            const float blo = h.origin[d], bhi = h.origin[d] + (float)h.dims[d] / h.inv_cell[d];
            l = fminf(fmaxf(l, blo), bhi); u2 = fminf(fmaxf(u2, blo), bhi);
            if (!(u2 >= l)) { l = blo; u2 = blo; }
            const float ext = u2 - l;
            float c = cell;
            if (ext / c > (float)(NF_GRID_MAX_DIM - 1)) c = ext / (float)(NF_GRID_MAX_DIM - 1);
            int dim = (int)floorf(ext / c) + 1;
            dim = max(1, min(dim, min(NF_GRID_MAX_DIM, h.dims[d])));
            hh.origin[d] = l; hh.inv_cell[d] = 1.0f / c; hh.dims[d] = dim; hh.sub0[d] = 0; hh.subd[d] = dim;
            ncell *= dim;
        }
        for (int d = 0; d < 3; ++d) {           // exact bounds of the points (all waves)
            float l = INFINITY, u2 = -INFINITY;
            for (int w2 = 0; w2 < TP_BLOCK / 64; ++w2) { l = fminf(l, s_red[w2][d]); u2 = fmaxf(u2, s_red[w2][3 + d]); }
            hh.pt_lo[d] = nf_f2ord(l); hh.pt_hi[d] = nf_f2ord(u2);
        }
        hh.n_cells = ncell;
        *(NfGridHeader*)ws = hh;
    }
    __syncthreads();
    const int nc = hh.n_cells;
    int* cell_start = (int*)(b + hh.off_cell_start);
    int* tmp_list = cells + nc;
    int* sorted_idx = (int*)(b + hh.off_sorted_idx);
    float4* sorted_pos = (float4*)(b + hh.off_sorted_pos);
    for (int c = tid; c < nc; c += TP_BLOCK) cells[c] = 0;
    __syncthreads();
    // ---- count
    int mycell[TP_MAX_PER_THREAD];
#pragma unroll
    for (int u = 0; u < TP_MAX_PER_THREAD; ++u) {
        const int i = u * TP_BLOCK + tid;
        mycell[u] = -1;
        if (i < n) {
            const int cx = nf_cell_coord(px[u], hh.origin[0], hh.inv_cell[0], hh.dims[0]);
            const int cy = nf_cell_coord(py[u], hh.origin[1], hh.inv_cell[1], hh.dims[1]);
            const int cz = nf_cell_coord(pz[u], hh.origin[2], hh.inv_cell[2], hh.dims[2]);
            mycell[u] = (cz * hh.dims[1] + cy) * hh.dims[0] + cx;
            atomicAdd(&cells[mycell[u]], 1);
        }
    }
    __syncthreads();
    // ---- exclusive scan of the cell counts (each thread a contiguous run, block scan of the run sums)
    const int per = (nc + TP_BLOCK - 1) / TP_BLOCK;
    const int c0 = tid * per, c1 = min(c0 + per, nc);
    int run = 0;
    for (int c = c0; c < c1; ++c) run += cells[c];
    {
        int x = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
        if (lane == 63) s_scan[wv] = x;
        __syncthreads();
        if (wv == 0) {
            int s2 = lane < TP_BLOCK / 64 ? s_scan[lane] : 0;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { const int y = __shfl_up(s2, o, 64); if (lane >= o) s2 += y; }
            if (lane < TP_BLOCK / 64) s_scan[lane] = s2;
        }
        __syncthreads();
        int base = (wv ? s_scan[wv - 1] : 0) + x - run;
        for (int c = c0; c < c1; ++c) { const int cnt = cells[c]; cells[c] = base; cell_start[c] = base; base += cnt; }
        if (tid == TP_BLOCK - 1) cell_start[nc] = s_scan[TP_BLOCK / 64 - 1];
    }
    __syncthreads();
    // ---- scatter (arrival order), cells[] ends up holding the END of every cell
#pragma unroll
    for (int u = 0; u < TP_MAX_PER_THREAD; ++u)
        if (mycell[u] >= 0) tmp_list[atomicAdd(&cells[mycell[u]], 1)] = u * TP_BLOCK + tid;
    __syncthreads();
    // ---- stable order inside each cell: rank = number of same-cell points with a smaller original index
#pragma unroll
    for (int u = 0; u < TP_MAX_PER_THREAD; ++u) {
        const int c = mycell[u];
        if (c < 0) continue;
        const int i = u * TP_BLOCK + tid;
        const int s2 = c ? cells[c - 1] : 0, e = cells[c];
        int rank = 0;
        for (int t = s2; t < e; ++t) rank += (tmp_list[t] < i);
        sorted_idx[s2 + rank] = i;
        sorted_pos[s2 + rank] = make_float4(px[u], py[u], pz[u], __int_as_float(i));
    }
}

extern "C" int nf_trans_prepare(const float* pos, const float* vel, const float gravity[3], float dt, int n, float cell,
                                const float bbox[6], void* grid_ws, size_t ws_bytes, float* pos_new, float* vel_new,
                                float* feats4, nf_stream_t stream)
{
    NF_CHECK_ARG(pos && vel && gravity && grid_ws && pos_new && vel_new && feats4, "null pointer");
    NfGridHeader h;
    size_t tot = 0;
    NF_CHECK_ARG(nf_grid_make_header(n, cell, bbox, &h, &tot) == NF_OK, "bad grid parameters");
    NF_CHECK_ARG(ws_bytes >= tot, "workspace too small");
    NF_CHECK_ARG(n > 0 && n <= TP_BLOCK * TP_MAX_PER_THREAD && h.n_cells + n <= TP_MAX_LDS_INTS,
                 "cloud or grid too large for the single-workgroup build (use nf_trans_integrate + nf_grid_build)");
    const size_t lds = (size_t)(h.n_cells + n) * sizeof(int);
    static bool attr_set[64] = {};              // per DEVICE
    if (nf_first_use_on_device(attr_set))
        hipFuncSetAttribute((const void*)k_trans_prepare, hipFuncAttributeMaxDynamicSharedMemorySize, TP_MAX_LDS_INTS * (int)sizeof(int));
    hipLaunchKernelGGL(k_trans_prepare, dim3(1), dim3(TP_BLOCK), lds, (hipStream_t)stream, h, grid_ws, pos, vel, gravity[0],
                       gravity[1], gravity[2], dt, cell, pos_new, vel_new, feats4);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// ball -> cube map of the pair offsets (identical to nf_cconv.hip:ball_to_cube)
__device__ __forceinline__ void tr_ball_to_cube(float& x, float& y, float& z)
{
    // sphere -> cylinder -> cube, volume preserving (identical to nf_cconv.hip:ball_to_cube)
    float sq = x * x + y * y + z * z;
    float nrm = sqrtf(sq);
    float xy2 = x * x + y * y;
    if (sq < 1e-12f) { x = y = z = 0.f; }
    else if (1.25f * z * z > xy2) {
        float s = sqrtf(3.f * nrm / (nrm + fabsf(z)));
        x *= s; y *= s; z = copysignf(nrm, z);
    } else {
        float s = nrm / sqrtf(xy2);
        x *= s; y *= s; z *= 1.5f;
    }
    float sq2 = x * x + y * y;
    float nxy = sqrtf(sq2);
    const float four_over_pi = 1.2732395447351628f;
    if (sq2 < 1e-12f) { x = y = 0.f; }
    else if (fabsf(y) <= fabsf(x)) {
        float t = copysignf(nxy, x);
        y = t * four_over_pi * atanf(y / x);
        x = t;
    } else {
        float t = copysignf(nxy, y);
        x = t * four_over_pi * atanf(x / y);
        y = t;
    }
}

// ================================================================================================
// Round 3: front kernel of the G-free inference step.  ONE launch per step does, for every particle i (a wave per
// (particle, cloud); blockIdx.y = 0 the fluid, 1 the container):
//   * the fixed-radius sweep (as k_trans_search) — hits are STAGED in the wave's LDS slice (neighbour, d^2, the trilinear
//     base node and fractions of the ball -> cube mapped offset, the window value), so that everything below runs with
//     lane = pair on dense data instead of inside the sparse sweep;
//   * fluid only: the ROW-ENTRY LISTS the continuous convolutions of conv1..3 consume (nf_cconv_gf.hip).  A pair touches the
//     2 x 2 x 2 filter nodes (cx + dx, cy + dy, cz + dz); its four row entries (dy, dz) go to row rho = (cz + dz) * 4 + (cy + dy):
//         { j | cx << 30,  w(dx = 0),  w(dx = 1) }          w = window * wx * wy * wz     (12 bytes)
//     bucketed by row (16 buckets per particle, offsets roff[i][0..16] as uint16), pair order kept inside a bucket.  A conv
//     layer then builds, for one row of 4 filter nodes at a time, Z[node][i][:] = sum_entries w * x[j][:] with REGISTER
//     accumulators (no read-modify-write, no scan of the other 15 rows' pairs) and feeds it to the matrix pipe;
//   * layer 0 (models/transmodel.py:116-120) in the patch-then-filter order of Open3D itself: the pairs are scattered
//     into a 64 x Cin patch in LDS (lane = pair: 8 x Cin float LDS atomics per 64 pairs, issued in lane order), then ONE
//     (64 Cin) x 32 product against the filter in LDS — 8 x fewer filter reads + FMAs than the pair-by-pair form of
//     k_trans_conv0 (which was bound by exactly those).
// Rows keep a fixed pitch (capacity per particle); a count above it is reported through overflow2 and the host redoes the
// step on the exact CSR path (ParticleNet._forward_impl) — nothing is poisoned, nothing surfaces later.
// ================================================================================================
#define TF_WAVES 8                       // particles in flight per workgroup (a wave each); the filter is staged once per workgroup
#define TF_MAXP 128                      // staged pairs per wave = the largest pitch this kernel serves

#ifndef TF_CHUNK
#define TF_CHUNK 64                      // pairs per round of the layer-0 patch build (round 4: 64 — a particle's ~40 pairs are ONE round of
#endif                                   // shuffles / scans / wave barriers instead of a full one and a nearly empty one)
struct TfStage {                         // per wave
    int j[TF_MAXP];                      // hits of the sweep, in hit order
    float d2[TF_MAXP];
    union {
        struct { float px[TF_MAXP], py[TF_MAXP], pz[TF_MAXP]; } pos;     // ... their positions (dead once the pair data is in registers)
        float feat[TF_CHUNK][4];         // then: the features of a round's neighbours
    } u;
    float iw[TF_CHUNK * 8];              // a round's (node, weight, pair) items, bucketed by node
    unsigned char it[TF_CHUNK * 8];
    int ccount[64], coff[64];
};

struct TfArgs {
    const void* grid[2];                 // 0: fluid (integrated positions), 1: container
    const float* q;                      // integrated positions (n x 3)
    const float* pos; const float* vel;  // FROM_STATE bodies integrate the positions themselves
    float gx, gy, gz, dt;
    const float* feats_f;                // fluid features [1, v] (n x 4)
    const float* feats_b;                // container normals (nb x 3)
    int n;
    float r2, extent;
    int use_window, pitch_f, pitch_b, relu_out;
    int32_t* counts2;                    // [2][n] true neighbour counts
    float* num_nbrs;                     // [n]
    int32_t* idx_f; float* d2_f;         // pitched rows (conv.nns)
    uint16_t* roff;                      // [n][20]
    uint32_t* ent;                       // [n][4 * pitch_f][3]
    const float* k_fluid; const float* b_fluid; const float* k_obst; const float* b_obst;
    const float* dense_w; const float* dense_b;
    float* a0;                           // n x 96
    unsigned long long* overflow2;       // [2] largest count seen above its pitch
    volatile int* host_flag;             // [3] or null: host-mapped (pinned) words; [0], [1] are written ONLY on overflow, [2]
                                         // receives step_id from the LAST workgroup to finish: the host spins on that word
                                         // (a HIP event recorded in the middle of a batch of launches completes with the batch)
    unsigned* done_ctr;                  // device counter of finished workgroups (zero between launches)
    int step_id;
};

// Cross-lane steps on the DPP path (VALU; __shfl_* compile to ds_bpermute_b32, a round trip through the LDS crossbar per step —
// a dependent chain of six of them is ~600 cycles).  Inclusive prefix sum over the wave: row_shr 1, 2, 4, 8 inside the 16-lane rows
// (out-of-row sources read 0), then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3 (the GCN scan).
__device__ __forceinline__ int tf_wave_scan_incl(int x)
{
#define TF_SHR(n) x += __builtin_amdgcn_update_dpp(0, x, 0x110 | (n), 0xf, 0xf, true);
    TF_SHR(1) TF_SHR(2) TF_SHR(4) TF_SHR(8)
#undef TF_SHR
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return x;
}
// max over the wave (non-negative ints), uniform result: row rotations, then the four rows through SGPRs
__device__ __forceinline__ int tf_wave_max(int v)
{
#define TF_ROR(n) v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x120 | (n), 0xf, 0xf, false));
    TF_ROR(1) TF_ROR(2) TF_ROR(4) TF_ROR(8)
#undef TF_ROR
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// Layer 0 as patch x filter for the NP particles a wave has patches of:  out[p][co] = sum_m patch_p[m] * Ks[m][co],  m = node * CI + ci.
// m = 8 k + s is split over the 8 lane groups s = lane >> 3 (8 CI values of k each); a lane holds 4 consecutive output channels
// (cog = lane & 7), so a filter row segment is ONE ds_read_b128 — the 8 groups' rows fall on distinct banks — shared by the NP
// particles, and the patches are stored transposed ([s][k]): four k of a group are one broadcast ds_read_b128.  (Before: 256
// ds_read_b32 and 128 FMAs per lane for EVERY particle — the whole 32 KB filter through the LDS port once per particle; that product,
// not the patch build, was the largest LDS consumer of the front kernel.)  The result is complete in the lanes of group 0.
template <int CI, int NP>
__device__ __forceinline__ void tf_gemv(const float* const (&pt)[NP], const float* __restrict__ Ks, float (&out)[NP][4])
{
    constexpr int KS = 8 * CI;
    const int lane = threadIdx.x & 63, s = lane >> 3, cog = lane & 7;
#pragma unroll
    for (int p = 0; p < NP; ++p) out[p][0] = out[p][1] = out[p][2] = out[p][3] = 0.f;
#pragma unroll
    for (int k4 = 0; k4 < KS; k4 += 4) {
        float pv[NP][4];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const float4 t = *(const float4*)(pt[p] + s * KS + k4);
            pv[p][0] = t.x; pv[p][1] = t.y; pv[p][2] = t.z; pv[p][3] = t.w;
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const float4 w = *(const float4*)(Ks + ((k4 + kk) * 8 + s) * 32 + cog * 4);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                out[p][0] = fmaf(pv[p][kk], w.x, out[p][0]); out[p][1] = fmaf(pv[p][kk], w.y, out[p][1]);
                out[p][2] = fmaf(pv[p][kk], w.z, out[p][2]); out[p][3] = fmaf(pv[p][kk], w.w, out[p][3]);
            }
        }
    }
#pragma unroll
    for (int o = 8; o < 64; o <<= 1)
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int c = 0; c < 4; ++c) out[p][c] += __shfl_xor(out[p][c], o, 64);
}

// interpolation data of one pair, exactly k_pair_precompute (nf_cconv.hip): base node (bx, by, bz), fractions, window
struct TfPair { int j, bx, by, bz; float fx, fy, fz, imp; };

__device__ __forceinline__ TfPair tf_pair_v(int j, float px, float py, float pz, float d2, float qx, float qy, float qz, float scale,
                                            float inv_r2, int use_window)
{
    TfPair P;
    P.j = j;
    float x = (px - qx) * scale, y = (py - qy) * scale, z = (pz - qz) * scale;
    tr_ball_to_cube(x, y, z);
    P.imp = 1.f;
    if (use_window) { const float tt = 1.f - d2 * inv_r2; P.imp = fminf(fmaxf(tt * tt * tt, 0.f), 1.f); }
    const float c[3] = {(x + 1.f) * 1.5f, (y + 1.f) * 1.5f, (z + 1.f) * 1.5f};
    int i0[3];
    float f[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float cc = fminf(fmaxf(c[d], 0.f), 3.f);
        const float fl = fminf(floorf(cc), 2.f);
        i0[d] = (int)fl;
        f[d] = cc - fl;
    }
    P.bx = i0[0]; P.by = i0[1]; P.bz = i0[2]; P.fx = f[0]; P.fy = f[1]; P.fz = f[2];
    return P;
}

__device__ __forceinline__ TfPair tf_pair(const TfStage& st, int t, float qx, float qy, float qz, float scale, float inv_r2, int use_window)
{
    return tf_pair_v(st.j[t], st.u.pos.px[t], st.u.pos.py[t], st.u.pos.pz[t], st.d2[t], qx, qy, qz, scale, inv_r2, use_window);
}

// LDS regions of the front body (static __shared__ in k_trans_front, carved from the dynamic block in k_trans_stage1)
struct TfLds {
    float* Ks;                          // 64 * CI * 32 floats: the layer-0 filter of this cloud
    TfStage* stage;                     // [TF_WAVES]
    int* rcnt; int* rcur; int* rbase;   // [TF_WAVES][16], [TF_WAVES][16], [TF_WAVES][17]
    int* list; float* bias;             // container half: the queued particles ([TF_LISTMAX] + the count), conv0_obstacle's bias row
};
#define TF_LISTMAX 512                  // particles one workgroup of the container half can be dealt
#define TF_LDS_BYTES(CI, NW) ((size_t)64 * (CI) * 32 * 4 + sizeof(TfStage) * (NW) + (size_t)(NW) * (16 + 16 + 17) * 4 + \
                              ((CI) == 3 ? (size_t)(TF_LISTMAX + 4 + 32) * 4 + 16 : 0))

template <int NW>
__device__ __forceinline__ TfLds tf_carve(char* base, int ci)
{
    TfLds L;
    L.Ks = (float*)base; base += (size_t)64 * ci * 32 * 4;
    L.stage = (TfStage*)base; base += sizeof(TfStage) * NW;
    L.rcnt = (int*)base; base += NW * 16 * 4;
    L.rcur = (int*)base; base += NW * 16 * 4;
    L.rbase = (int*)base; base += NW * 17 * 4;
    base = (char*)(((uintptr_t)base + 15) & ~(uintptr_t)15);
    L.list = (int*)base; base += (TF_LISTMAX + 4) * 4;
    L.bias = (float*)base;
    return L;
}

// per-wave LDS pointers of the front body
struct TfWave { TfStage* st; float* patch; int* rcnt; int* rcur; int* rbase; const float* Ks; };

// The neighbour features of a pair (lane = pair), requested with the pair data instead of inside the patch rounds
template <int WHICH>
__device__ __forceinline__ float4 tf_feat(const TfArgs& A, int j)
{
    if (!WHICH) return *(const float4*)(A.feats_f + 4 * (size_t)j);
    return make_float4(A.feats_b[3 * (size_t)j], A.feats_b[3 * (size_t)j + 1], A.feats_b[3 * (size_t)j + 2], 0.f);
}

// Everything behind the pair data of particle i (np staged pairs: P / F hold them with lane = pair, 64 per register set; the fluid
// half's row counts are in rcnt) except the product with the filter: the layer-0 patch (left in W.patch, transposed), roff + the row
// entries (fluid).
template <int WHICH>
__device__ __forceinline__ void tf_tail(const TfArgs& A, const TfWave& W, int i, int np, const TfPair (&P)[2], const float4 (&F)[2])
{
    static_assert(TF_CHUNK == 64, "a round of the patch build is one register set of pairs");
    constexpr int CI = WHICH ? 3 : 4;
    const int lane = threadIdx.x & 63;
    TfStage& st = *W.st;
    float* const patch = W.patch;
    int* const rcnt = W.rcnt;
    int* const rcur = W.rcur;
    int* const rbase = W.rbase;
    const float* const Ks = W.Ks;
    // ---- layer-0 patch P[node][ci] = sum_pairs w(pair, node) * feat[pair][ci], in rounds of 64 pairs WITHOUT float atomics
    // (ds_add_f32 with the same-address collisions of this scatter cost 44 of the kernel's 70 us): a pair's 8 (node, weight)
    // items are bucketed by node with integer LDS atomics (the returned slot orders a bucket), then lane = NODE walks its
    // bucket and accumulates in registers.
    float pacc[4] = {0.f, 0.f, 0.f, 0.f};
#ifndef TF_AB_SKIP_PATCH
    for (int t0 = 0; t0 < np; t0 += TF_CHUNK) {
        st.ccount[lane] = 0;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        // a round = one register set of pair data (lane = pair), as pass 2 left it
        const bool hi = t0 >= 64;
        const TfPair Q = hi ? P[1] : P[0];
        const float4 fq = hi ? F[1] : F[0];
        const bool mine = lane < TF_CHUNK && t0 + lane < np;
        int slot[8], cellk[8];
        float wk[8];
        if (mine) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
                wk[k] = Q.imp * ((dx ? Q.fx : 1.f - Q.fx) * (dy ? Q.fy : 1.f - Q.fy) * (dz ? Q.fz : 1.f - Q.fz));
                cellk[k] = ((Q.bz + dz) * 4 + (Q.by + dy)) * 4 + (Q.bx + dx);
                slot[k] = atomicAdd(&st.ccount[cellk[k]], 1);
            }
            *(float4*)st.u.feat[lane] = fq;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        const int mycnt = st.ccount[lane];
        const int myoff = tf_wave_scan_incl(mycnt) - mycnt;
        st.coff[lane] = myoff;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        if (mine) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int pos = st.coff[cellk[k]] + slot[k];
                st.iw[pos] = wk[k];
                st.it[pos] = (unsigned char)lane;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        const int cmax = tf_wave_max(mycnt);
        // (two items per trip: their weight -> pair -> feature reads are independent chains; the sum keeps the bucket's order)
        for (int e = 0; e < cmax; e += 2) {
            const bool ok0 = e < mycnt, ok1 = e + 1 < mycnt;
            const float w0 = ok0 ? st.iw[myoff + e] : 0.f, w1 = ok1 ? st.iw[myoff + e + 1] : 0.f;
            const int t0_ = ok0 ? st.it[myoff + e] : 0, t1_ = ok1 ? st.it[myoff + e + 1] : 0;
            const float4 f0 = *(const float4*)st.u.feat[t0_], f1 = *(const float4*)st.u.feat[t1_];
            if (ok0) { pacc[0] = fmaf(w0, f0.x, pacc[0]); pacc[1] = fmaf(w0, f0.y, pacc[1]); pacc[2] = fmaf(w0, f0.z, pacc[2]); pacc[3] = fmaf(w0, f0.w, pacc[3]); }
            if (ok1) { pacc[0] = fmaf(w1, f1.x, pacc[0]); pacc[1] = fmaf(w1, f1.y, pacc[1]); pacc[2] = fmaf(w1, f1.z, pacc[2]); pacc[3] = fmaf(w1, f1.w, pacc[3]); }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
#endif
    // the patch, transposed for tf_gemv: element m = node * CI + ci goes to [m & 7][m >> 3]
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
        if (ci < CI) { const int m = lane * CI + ci; patch[(m & 7) * (8 * CI) + (m >> 3)] = pacc[ci]; }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (!WHICH) {
        // ---- pass 3: exclusive scan of the 16 row counts -> roff
        int c = lane < 16 ? rcnt[lane] : 0;
        int x = c;                                          // (16 counts: the scan stays inside the first DPP row)
        x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);
        x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);
        x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);
        x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);
        if (lane < 16) rbase[lane] = x - c;
        if (lane == 15) rbase[16] = x;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        if (lane < 17) A.roff[(size_t)i * 20 + lane] = (uint16_t)rbase[lane];
        // ---- pass 4 (lane = pair): the four row entries of every pair; the LDS cursors advance in lane order, so a
        // bucket keeps the pair order
        uint32_t* ebase = A.ent + (size_t)i * (size_t)(4 * A.pitch_f) * 3;
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
#ifdef TF_AB_SKIP_ENT
            break;
#endif
            if (64 * c2 < np) {
                if (64 * c2 + lane < np) {
                    const TfPair& Q = P[c2];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int dy = r & 1, dz = r >> 1;
                        const int rho = (Q.bz + dz) * 4 + Q.by + dy;
                        const int e = rbase[rho] + atomicAdd(&rcur[rho], 1);
                        // the weights of k_pair_precompute, same expression and association
                        const float w0 = Q.imp * ((1.f - Q.fx) * (dy ? Q.fy : 1.f - Q.fy) * (dz ? Q.fz : 1.f - Q.fz));
                        const float w1 = Q.imp * (Q.fx * (dy ? Q.fy : 1.f - Q.fy) * (dz ? Q.fz : 1.f - Q.fz));
                        uint32_t* dst = ebase + 3 * (size_t)e;
                        dst[0] = (uint32_t)Q.j | ((uint32_t)Q.bx << 30);
                        dst[1] = __float_as_uint(w0);
                        dst[2] = __float_as_uint(w1);
                    }
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();          // the next particle of this wave reuses the stage / patch / counters
}

// Layer 0's product + output rows for the particles idx[0..NP) whose patches sit in pt[]: conv0 (+ bias, ReLU) from lane group 0, the
// fluid's Linear branch on the particle's own features from the upper half-wave.
template <int WHICH, int NP>
__device__ __forceinline__ void tf_layer0_out(const TfArgs& A, const float* __restrict__ Ks, const float* const (&pt)[NP], const int (&idx)[NP])
{
    constexpr int CI = WHICH ? 3 : 4;
    const int lane = threadIdx.x & 63;
    float o[NP][4];
#ifdef TF_AB_SKIP_GEMV
    for (int p = 0; p < NP; ++p) o[p][0] = o[p][1] = o[p][2] = o[p][3] = pt[p][lane];
#else
    tf_gemv<CI, NP>(pt, Ks, o);
#endif
    if (lane < 8) {
        const float4 b = *(const float4*)((WHICH ? A.b_obst : A.b_fluid) + lane * 4);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            float4 v = make_float4(o[p][0] + b.x, o[p][1] + b.y, o[p][2] + b.z, o[p][3] + b.w);
            if (A.relu_out) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
            *(float4*)(A.a0 + (size_t)idx[p] * 96 + (WHICH ? 0 : 32) + lane * 4) = v;
        }
    } else if (!WHICH && lane >= 32) {
        const int co = lane & 31;
        const float4 dw = *(const float4*)(A.dense_w + co * 4);
        const float db = A.dense_b[co];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const float4 f = *(const float4*)(A.feats_f + (size_t)idx[p] * 4);
            float s2 = db;
            s2 += f.x * dw.x; s2 += f.y * dw.y; s2 += f.z * dw.z; s2 += f.w * dw.w;
            A.a0[(size_t)idx[p] * 96 + 64 + co] = A.relu_out ? fmaxf(s2, 0.f) : s2;
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();          // the next particles of this wave reuse the stage / patch / counters
}

// The front body of one cloud (WHICH = 0 the fluid, 1 the container) for the particles blk * TF_WAVES + wave, + nblk * TF_WAVES, ...
// FROM_STATE: the query positions are integrated here from (pos, vel) — the same expressions as the grid build, bit for bit — so
// that the container half can run BESIDE the fluid grid build instead of behind it (k_trans_stage1).
// Latency: a particle's chain is position -> cell -> row ranges -> candidates -> (pair data) -> neighbour features, four dependent
// global round trips; with ~3 particles per wave that chain, not the arithmetic, was a third of the kernel.  The position of the
// particle after next and the row ranges of the next one are requested while the current one is processed, and the filter is
// staged behind the first requests instead of in front of them.
template <int WHICH, bool FROM_STATE, int NW>
__device__ __forceinline__ void tf_body(const TfArgs& A, const TfLds& L, int blk, int nblk)
{
    constexpr int CI = WHICH ? 3 : 4;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    TfStage& st = L.stage[wv];
    float* const patch = st.iw;          // (the patch of a particle is written when the rounds' items are dead: 256 of iw's floats)
    int* const rcnt = L.rcnt + wv * 16;
    int* const rcur = L.rcur + wv * 16;
    int* const rbase = L.rbase + wv * 17;
    const float* const Ks = L.Ks;
    const int pitch = WHICH ? A.pitch_b : A.pitch_f;
    const int cap = min(pitch, TF_MAXP);
    const NfGridView g = nf_grid_view(A.grid[WHICH]);
    const unsigned long long lt = (1ull << lane) - 1ull;
    const float radius = 0.5f * A.extent, inv_r2 = 1.f / (radius * radius), scale = 2.f / A.extent;
    // Work items.  Fluid: a wave walks CONSECUTIVE particle indices (index order is spatially coherent: their cell rows and candidates
    // overlap in the L1 / L2: 46.0 -> 44.2 us against indices strided by the launch width, round 4).  Container: see the pre-pass.
    const int per_wave = (A.n + nblk * NW - 1) / (nblk * NW);
    const int i0 = (blk * NW + wv) * per_wave;
    int k = 0, kstep = 1, kend = WHICH ? 0 : min(A.n, i0 + per_wave) - i0;

    auto load_q = [&](int ii, float& x, float& y, float& z) __attribute__((always_inline)) {
        x = y = z = 0.f;
        if (ii < A.n) {
            if constexpr (FROM_STATE) {
                const float g3[3] = {A.gx, A.gy, A.gz};
                float o[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const float v = A.vel[3 * (size_t)ii + d];
                    const float vn = v + g3[d] * A.dt;                    // k_trans_stage1's / k_trans_integrate's expressions
                    o[d] = A.pos[3 * (size_t)ii + d] + (v + vn) / 2 * A.dt;
                }
                x = o[0]; y = o[1]; z = o[2];
            } else {
                x = A.q[3 * (size_t)ii]; y = A.q[3 * (size_t)ii + 1]; z = A.q[3 * (size_t)ii + 2];
            }
        }
    };
    // row ranges of the 9 (z, y) rows around the query's cell (lanes 0..8); cells are listed for the grid's sub-box only
    auto load_ranges = [&](float x, float y, float z, int& rs, int& re) __attribute__((always_inline)) {
        const int cx = min(max(nf_cell_coord(x, g.ox, g.icx, g.dx) - g.s0x, 0), g.sdx - 1);
        const int cy = min(max(nf_cell_coord(y, g.oy, g.icy, g.dy) - g.s0y, 0), g.sdy - 1);
        const int cz = min(max(nf_cell_coord(z, g.oz, g.icz, g.dz) - g.s0z, 0), g.sdz - 1);
        rs = re = 0;
        if (lane < 9) {
            const int zz = cz - 1 + lane / 3, yy = cy - 1 + lane % 3;
            if (zz >= 0 && zz < g.sdz && yy >= 0 && yy < g.sdy) {
                const int r0 = (zz * g.sdy + yy) * g.sdx;
                rs = g.cell_start[r0 + max(cx - 1, 0)];
                re = g.cell_start[r0 + min(cx + 1, g.sdx - 1) + 1];
            }
        }
    };
    float qx = 0.f, qy = 0.f, qz = 0.f;
    int rs = 0, re = 0;
    bool first_particle = !WHICH;
    if (!WHICH && kend > 0) load_q(i0, qx, qy, qz);
    {   // the layer-0 filter of this cloud: requested now, parked in LDS behind the first requests of the particles
        const float4* ksrc = (const float4*)(WHICH ? A.k_obst : A.k_fluid);
        constexpr int KN4 = 64 * CI * 32 / 4, PER = (KN4 + 64 * NW - 1) / (64 * NW);
        float4 kv[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int t = threadIdx.x + u * 64 * NW;
            kv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < KN4) kv[u] = ksrc[t];
        }
        if constexpr (!WHICH) {
            if (g.sdx > 0 && kend > 0) load_ranges(qx, qy, qz, rs, re);
        } else {
            // Container pre-pass, a THREAD per particle: most particles of a fluid body have no container point in the 27 cells
            // around them (88 % of the watercube) — they get conv0_obstacle = its bias here, 64 at a time, and only the others are
            // queued for the wave-per-particle sweep below.  Particles are dealt to the workgroups with a stride (the ones near
            // the walls are consecutive indices: a contiguous deal gave some workgroups nothing but them).
            if (threadIdx.x == 0) L.list[TF_LISTMAX] = 0;
            if (threadIdx.x < 32) { const float v = A.b_obst[threadIdx.x]; L.bias[threadIdx.x] = A.relu_out ? fmaxf(v, 0.f) : v; }
            __syncthreads();
            for (int t = threadIdx.x; blk + nblk * t < A.n; t += 64 * NW) {
                const int ii = blk + nblk * t;
                float x, y, z;
                load_q(ii, x, y, z);
                int tot = 0;
                if (g.sdx > 0) {
                    const int cx = min(max(nf_cell_coord(x, g.ox, g.icx, g.dx) - g.s0x, 0), g.sdx - 1);
                    const int cy = min(max(nf_cell_coord(y, g.oy, g.icy, g.dy) - g.s0y, 0), g.sdy - 1);
                    const int cz = min(max(nf_cell_coord(z, g.oz, g.icz, g.dz) - g.s0z, 0), g.sdz - 1);
#pragma unroll
                    for (int r = 0; r < 9; ++r) {
                        const int zz = cz - 1 + r / 3, yy = cy - 1 + r % 3;
                        if (zz >= 0 && zz < g.sdz && yy >= 0 && yy < g.sdy) {
                            const int r0 = (zz * g.sdy + yy) * g.sdx;
                            tot += g.cell_start[r0 + min(cx + 1, g.sdx - 1) + 1] - g.cell_start[r0 + max(cx - 1, 0)];
                        }
                    }
                }
                if (tot > 0) L.list[atomicAdd(&L.list[TF_LISTMAX], 1)] = ii;
                else {
                    A.counts2[(size_t)A.n + ii] = 0;
                    float4* orow4 = (float4*)(A.a0 + (size_t)ii * 96);
#pragma unroll
                    for (int c = 0; c < 8; ++c) orow4[c] = ((const float4*)L.bias)[c];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int t = threadIdx.x + u * 64 * NW;
            if (t < KN4) ((float4*)L.Ks)[t] = kv[u];
        }
        __syncthreads();
    }
    if constexpr (WHICH) { k = wv; kstep = NW; kend = L.list[TF_LISTMAX]; }
    for (; k < kend; k += kstep) {
        const int i = WHICH ? L.list[k] : i0 + k;
        if (!first_particle) { load_q(i, qx, qy, qz); if (g.sdx > 0) load_ranges(qx, qy, qz, rs, re); }
        first_particle = false;
        if (lane < 16) { rcnt[lane] = 0; rcur[lane] = 0; }
        int cnt = 0;
        // ---- pass 1: the sweep (cell-major, the order of nf_radius_fill).  The first 64 candidates of EVERY row of a group are
        // requested before the first is looked at
#pragma unroll
        for (int grp = 0; grp < 2; ++grp) {
            const int R0 = grp ? 5 : 0, RN = grp ? 4 : 5;
            int rows_s[5], rows_e[5];
            float4 first[5];
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                rows_s[r] = rows_e[r] = 0;
                first[r] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < RN) {
                    rows_s[r] = __builtin_amdgcn_readlane(rs, R0 + r);
                    rows_e[r] = __builtin_amdgcn_readlane(re, R0 + r);
                    if (rows_s[r] + lane < rows_e[r]) first[r] = g.sorted_pos[rows_s[r] + lane];
                }
            }
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                if (r >= RN) continue;
                for (int t0 = rows_s[r]; t0 < rows_e[r]; t0 += 64) {
                    const int t = t0 + lane;
                    float4 p = first[r];
                    if (t0 != rows_s[r]) { p = make_float4(0.f, 0.f, 0.f, 0.f); if (t < rows_e[r]) p = g.sorted_pos[t]; }
                    bool hit = false;
                    float d2 = 0.f;
                    if (t < rows_e[r]) {
                        d2 = nf_dist2(qx, qy, qz, p.x, p.y, p.z);
                        hit = d2 <= A.r2 && !(p.x == qx && p.y == qy && p.z == qz);      // radius_search_ignore_query_points=True
                    }
                    const unsigned long long m = __ballot(hit);
                    if (hit) {
                        const int w = cnt + __popcll(m & lt);
                        if (w < cap) { st.j[w] = __float_as_int(p.w); st.d2[w] = d2; st.u.pos.px[w] = p.x; st.u.pos.py[w] = p.y; st.u.pos.pz[w] = p.z; }
                    }
                    cnt += __popcll(m);
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            A.counts2[(size_t)WHICH * A.n + i] = cnt;
            if (!WHICH) A.num_nbrs[i] = (float)cnt;
            if (cnt > pitch) {
                atomicMax(A.overflow2 + WHICH, (unsigned long long)cnt);
                if (A.host_flag) { A.host_flag[WHICH] = cnt; __threadfence_system(); }      // (any overflowing count will do as the flag)
            }
        }
        const int np = min(cnt, cap);
        const float cqx = qx, cqy = qy, cqz = qz;
        float* orow = A.a0 + (size_t)i * 96;
        if (WHICH && np == 0) {
            // no container point in reach (the large majority of a fluid body): conv0_obstacle = its bias
            if (lane < 32) { const float v = A.b_obst[lane]; orow[lane] = A.relu_out ? fmaxf(v, 0.f) : v; }
            continue;
        }
        // ---- pass 2 (lane = pair, dense): interpolation data of up to two chunks of 64 pairs (kept in registers for the passes
        // below), the pitched neighbour rows, row counts
        TfPair P[2];
        float4 F[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int t = 64 * c + lane;
            P[c].j = 0; P[c].bx = P[c].by = P[c].bz = 0; P[c].fx = P[c].fy = P[c].fz = P[c].imp = 0.f;
            F[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < np) {
                F[c] = tf_feat<WHICH>(A, st.j[t]);
                P[c] = tf_pair(st, t, cqx, cqy, cqz, scale, inv_r2, A.use_window);
                if (!WHICH) {
                    A.idx_f[(int64_t)i * pitch + t] = P[c].j; A.d2_f[(int64_t)i * pitch + t] = st.d2[t];
#pragma unroll
                    for (int r = 0; r < 4; ++r) atomicAdd(&rcnt[(P[c].bz + (r >> 1)) * 4 + P[c].by + (r & 1)], 1);
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        tf_tail<WHICH>(A, TfWave{&st, patch, rcnt, rcur, rbase, Ks}, i, np, P, F);
        {
            const float* const pt[1] = {patch};
            const int idx[1] = {i};
            tf_layer0_out<WHICH, 1>(A, Ks, pt, idx);
        }
    }
}

// The fluid half behind an ALL-PAIRS search (k_trans_stage1b wrote the pitched rows idx_f / d2_f in ascending neighbour index and the
// counts): a particle's chain is {its row, its count, its position} -> {the neighbours' positions and features} -> arithmetic, two
// global round trips instead of the grid walk's four (position -> cell -> row ranges -> candidates -> features).  No sweep, no staging
// of hits: the pair data is computed with lane = pair straight from the row.
template <int NW>
__device__ __forceinline__ void tf_body_rows(const TfArgs& A, const TfLds& L, int blk, int nblk)
{
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    TfStage& st = L.stage[wv];
    const TfWave W{&st, st.iw, L.rcnt + wv * 16, L.rcur + wv * 16, L.rbase + wv * 17, L.Ks};
    const int pitch = A.pitch_f, cap = min(pitch, TF_MAXP);
    const float radius = 0.5f * A.extent, inv_r2 = 1.f / (radius * radius), scale = 2.f / A.extent;
    // workgroup blk owns the particles [n blk / nblk, n (blk + 1) / nblk), its waves equal shares of them (consecutive indices: index
    // order is spatially coherent, so a wave's rows and gathers overlap in the L1 / L2)
    const int w0 = (int)((unsigned)A.n * (unsigned)blk / (unsigned)nblk), wn = (int)((unsigned)A.n * (unsigned)(blk + 1) / (unsigned)nblk) - w0;   // (n <= 16 384)
    int i = w0 + wn * wv / NW;
    const int i_end = w0 + wn * (wv + 1) / NW;

    struct Row { float qx, qy, qz; int cnt, j0, j1; float d0, d1; };
    auto load_row = [&](int ii, Row& r) __attribute__((always_inline)) {
        r.qx = r.qy = r.qz = r.d0 = r.d1 = 0.f; r.cnt = r.j0 = r.j1 = 0;
        if (ii < i_end) {
            r.qx = A.q[3 * (size_t)ii]; r.qy = A.q[3 * (size_t)ii + 1]; r.qz = A.q[3 * (size_t)ii + 2];
            r.cnt = A.counts2[ii];
            const int64_t base = (int64_t)ii * pitch;
            // (slots behind the count hold whatever the buffer held: they are never used as indices — every use is guarded by np)
            if (lane < cap) { r.j0 = A.idx_f[base + lane]; r.d0 = A.d2_f[base + lane]; }
            if (64 + lane < cap) { r.j1 = A.idx_f[base + 64 + lane]; r.d1 = A.d2_f[base + 64 + lane]; }
        }
    };
    Row cur;
    load_row(i, cur);
    {   // the layer-0 filter: requested behind the first row, parked in LDS
        const float4* ksrc = (const float4*)A.k_fluid;
        constexpr int KN4 = 64 * 4 * 32 / 4, PER = (KN4 + 64 * NW - 1) / (64 * NW);
        float4 kv[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int t = threadIdx.x + u * 64 * NW;
            kv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < KN4) kv[u] = ksrc[t];
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int t = threadIdx.x + u * 64 * NW;
            if (t < KN4) ((float4*)L.Ks)[t] = kv[u];
        }
        __syncthreads();
    }
    static_assert(sizeof(((TfStage*)0)->j) + sizeof(((TfStage*)0)->d2) >= 64 * 4 * sizeof(float), "room for a parked patch");
#if defined(TF_AB_ROWS_STOP) && TF_AB_ROWS_STOP == 1
    if (lane == 0 && i < i_end) A.a0[(size_t)i * 96] = (float)cur.cnt;
    return;
#endif
    int gcount = 0, gidx[2] = {0, 0};
    for (; i < i_end; ++i) {
#ifdef TF_AB_ROWAHEAD
        Row nxt;
        load_row(i + 1, nxt);
#endif
        if (lane < 16) { W.rcnt[lane] = 0; W.rcur[lane] = 0; }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        const int np = min(cur.cnt, cap);
        TfPair P[2];
        float4 F[2];
        float px[2], py[2], pz[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {               // all gathers of the particle in flight before the first is used
            const int t = 64 * c + lane, j = c ? cur.j1 : cur.j0;
            F[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            px[c] = py[c] = pz[c] = 0.f;
            if (t < np) {
                px[c] = A.q[3 * (size_t)j]; py[c] = A.q[3 * (size_t)j + 1]; pz[c] = A.q[3 * (size_t)j + 2];
                F[c] = tf_feat<0>(A, j);
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int t = 64 * c + lane;
            P[c].j = 0; P[c].bx = P[c].by = P[c].bz = 0; P[c].fx = P[c].fy = P[c].fz = P[c].imp = 0.f;
            if (t < np) {
                P[c] = tf_pair_v(c ? cur.j1 : cur.j0, px[c], py[c], pz[c], c ? cur.d1 : cur.d0, cur.qx, cur.qy, cur.qz, scale, inv_r2, A.use_window);
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(&W.rcnt[(P[c].bz + (r >> 1)) * 4 + P[c].by + (r & 1)], 1);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#if defined(TF_AB_ROWS_STOP) && TF_AB_ROWS_STOP == 2
        if (lane < 16) A.a0[(size_t)i * 96 + lane] = (float)W.rcnt[lane] + P[0].fx + P[1].fy + F[0].x + F[1].y;
        load_row(i + 1, cur);
        continue;
#endif
        // the patch of the first particle of a group of two parks in the (here unused) sweep staging, the second one in the rounds'
        // item buffer; one product with the filter serves both
        const bool second = gcount == 1;
        TfWave Wp = W;
        Wp.patch = second ? st.iw : (float*)st.j;
        tf_tail<0>(A, Wp, i, np, P, F);
        gidx[gcount++] = i;
        if (gcount == 2) {
            const float* const pt[2] = {(const float*)st.j, st.iw};
            const int idx[2] = {gidx[0], gidx[1]};
            tf_layer0_out<0, 2>(A, W.Ks, pt, idx);
            gcount = 0;
        } else if (i + 1 == i_end) {
            const float* const pt[1] = {(const float*)st.j};
            const int idx[1] = {gidx[0]};
            tf_layer0_out<0, 1>(A, W.Ks, pt, idx);
            gcount = 0;
        }
#ifdef TF_AB_ROWAHEAD
        cur = nxt;
#else
        // (requesting the next particle's row a particle ahead was measured: 32.5 vs 30.6 us — as in the grid walk, the early requests
        // queue in front of the current particle's gathers)
        load_row(i + 1, cur);
#endif
    }
}

// the completion word: the LAST workgroup of the launch to arrive writes step_id into the pinned host word
__device__ __forceinline__ void tf_arrive(const TfArgs& A, unsigned total)
{
    if (!A.host_flag || !A.done_ctr) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        // (this workgroup's overflow words, if it raised any, were followed by their own system-scope fence; the arrival itself is
        // a device-scope matter — a system-scope fence here, in every workgroup, costs the launch microseconds)
#ifdef TF_AB_SYSFENCE
        __threadfence_system();
#else
        __threadfence();
#endif
        const unsigned old = atomicAdd(A.done_ctr, 1u);
        if (old == total - 1) {
            *A.done_ctr = 0u;
            A.host_flag[2] = A.step_id;
            __threadfence_system();
        }
    }
}

// blockIdx.y = cloud when clouds == 3 (the stand-alone entry point: both halves in one launch), else the cloud clouds - 1
__global__ void __launch_bounds__(64 * TF_WAVES, 4) k_trans_front(TfArgs A, int clouds)
{
    __shared__ __attribute__((aligned(16))) char lds[TF_LDS_BYTES(4, TF_WAVES)];
    const int which = clouds == 3 ? (int)blockIdx.y : clouds - 1;
    if (which == 0) tf_body<0, false, TF_WAVES>(A, tf_carve<TF_WAVES>(lds, 4), blockIdx.x, gridDim.x);
    else tf_body<1, false, TF_WAVES>(A, tf_carve<TF_WAVES>(lds, 3), blockIdx.x, gridDim.x);
    tf_arrive(A, gridDim.x * gridDim.y);
}

// the fluid half behind k_trans_stage1b (rows + counts already written) in workgroups [0, nf); behind them (when the launch has any)
// workgroups of the container half, on the CUs' second workgroup slots
__global__ void __launch_bounds__(64 * TF_WAVES, 4) k_trans_front_rows(TfArgs A, int nf)
{
    __shared__ __attribute__((aligned(16))) char lds[TF_LDS_BYTES(4, TF_WAVES)];
    static_assert(TF_LDS_BYTES(3, TF_WAVES) <= TF_LDS_BYTES(4, TF_WAVES), "the container half fits the fluid half's LDS");
    if ((int)blockIdx.x < nf) tf_body_rows<TF_WAVES>(A, tf_carve<TF_WAVES>(lds, 4), blockIdx.x, nf);
    else tf_body<1, false, TF_WAVES>(A, tf_carve<TF_WAVES>(lds, 3), (int)blockIdx.x - nf, (int)gridDim.x - nf);
    tf_arrive(A, gridDim.x);
}

extern "C" int nf_trans_front_max_pitch(void) { return TF_MAXP; }

static int tf_ncu()
{   // (cached per device: hipGetDeviceProperties costs tens of microseconds, this runs once per step)
    static int cu_of[64] = {};
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        int v = __atomic_load_n(&cu_of[dev], __ATOMIC_RELAXED);
        if (!v && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) __atomic_store_n(&cu_of[dev], v, __ATOMIC_RELAXED);
        if (v > 0) ncu = v;
    }
    return ncu;
}

// two workgroups of 8 waves fit a CU (61 KB of LDS each: the filter is staged once per workgroup): size the grid so that
// every workgroup is resident at once and each wave walks the same number of particles
// workgroups of the container half: at most TF_LISTMAX particles are dealt to one (its pre-pass queues them in LDS)
static int tf_box_blocks(int blocks, int n)
{
    const int need = (n + TF_LISTMAX - 1) / TF_LISTMAX;
    return blocks < need ? need : blocks;
}

static int tf_blocks(int n, int wg_per_cu = 1)
{
    const int ncu = tf_ncu() * wg_per_cu, per = TF_WAVES, iters = (n + ncu * per - 1) / (ncu * per);
    const int blocks = (n + per * iters - 1) / (per * iters);
    return blocks < 1 ? 1 : blocks;
}

extern "C" int nf_trans_front(const void* fluid_grid, const void* box_grid, const float* queries, const float* fluid_feats,
                              const float* box_feats, int n, float radius, float extent, int use_window, int pitch_fluid,
                              int pitch_box, int32_t* counts2, float* num_fluid_nbrs, int32_t* idx_f, float* d2_f, uint16_t* roff,
                              uint32_t* entries, const float* kernel_fluid, const float* bias_fluid, const float* kernel_obstacle,
                              const float* bias_obstacle, const float* dense_w, const float* dense_b, float* out96,
                              int relu_out, int64_t* overflow2, int32_t* host_flag3, uint32_t* done_counter, int step_id,
                              nf_stream_t stream)
{
    NF_CHECK_ARG(fluid_grid && box_grid && queries && fluid_feats && box_feats && counts2 && num_fluid_nbrs && idx_f && d2_f && roff &&
                 entries && kernel_fluid && bias_fluid && kernel_obstacle && bias_obstacle && dense_w && dense_b && out96 && overflow2,
                 "null pointer");
    NF_CHECK_ARG(n > 0 && radius > 0.f && extent > 0.f, "bad n/radius/extent");
    NF_CHECK_ARG(pitch_fluid >= 1 && pitch_fluid <= TF_MAXP && pitch_box >= 1 && pitch_box <= TF_MAXP, "pitch must be in [1, nf_trans_front_max_pitch()]");
    NF_CHECK_ARG(!host_flag3 || done_counter, "the completion word needs the workgroup counter");
    TfArgs A;
    memset(&A, 0, sizeof(A));
    A.grid[0] = fluid_grid; A.grid[1] = box_grid; A.q = queries; A.feats_f = fluid_feats; A.feats_b = box_feats; A.n = n;
    A.r2 = radius * radius; A.extent = extent; A.use_window = use_window; A.pitch_f = pitch_fluid; A.pitch_b = pitch_box;
    A.relu_out = relu_out;
    A.counts2 = counts2; A.num_nbrs = num_fluid_nbrs; A.idx_f = idx_f; A.d2_f = d2_f; A.roff = roff; A.ent = entries;
    A.k_fluid = kernel_fluid; A.b_fluid = bias_fluid; A.k_obst = kernel_obstacle; A.b_obst = bias_obstacle;
    A.dense_w = dense_w; A.dense_b = dense_b; A.a0 = out96; A.overflow2 = (unsigned long long*)overflow2;
    A.host_flag = (volatile int*)host_flag3; A.done_ctr = done_counter; A.step_id = step_id;
#ifdef TF_AB_NO_BOX
    hipLaunchKernelGGL(k_trans_front, dim3(tf_blocks(n), 1), dim3(64 * TF_WAVES), 0, (hipStream_t)stream, A, 1);
#else
    hipLaunchKernelGGL(k_trans_front, dim3(tf_box_blocks(tf_blocks(n), n), 2), dim3(64 * TF_WAVES), 0, (hipStream_t)stream, A, 3);
#endif
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// ================================================================================================
// Round 4: stage 1 of the step = ONE launch that does, side by side,
//   * workgroup 0: gravity integration (B1) + the fluid cell grid of the integrated positions (ts_build: k_trans_prepare's
//     single-workgroup counting sort in LDS, with the per-thread particle count a TEMPLATE argument: every load of the
//     integration is issued up front — the guarded 16-way unrolled loop of k_trans_prepare walked its 5 live iterations one
//     global round trip after the other);
//   * workgroups 1...: the CONTAINER half of the front kernel (box sweep + conv0_obstacle).  It needs the integrated positions
//     only — which it computes itself from (pos, vel), bit-identical — and the static box grid, so it runs beside the grid build
//     instead of behind it; most particles of a body have no container point in reach and leave after the sweep.
// The fluid half (k_trans_front with clouds = 1) follows as its own launch: it needs the complete grid.
// Measured and set aside (round 4): the grid build spread over ceil(n / 512) workgroups — binning on the static container grid
// with returning global atomics, the occupied cell range reduced with integer atomics, the LAST workgroup to arrive (device-scope
// ticket behind an agent-scope release / acquire) scanning the occupied sub-box and ordering each cell by original index: 31.9 us
// against 22 us for the single workgroup.  Every hand-over is a global round trip behind a fence (release ~2-6 us, ticket, acquire,
// the counters' exchange, the per-particle (cell, slot) reads), and the finishing workgroup still walks all n particles alone; a
// cloud of 5 000 particles is too small for its build to be anything but a chain of latencies, and the chain is shortest when it
// stays in one workgroup's LDS.  (The sub-box fields of the grid header — cell lists for a sub-range of the grid's cells — were
// built for it and are kept: ordinary builds set them to the whole grid.)
// ================================================================================================
#define TS_BLOCK 1024
#define TS_WAVES (TS_BLOCK / 64)

struct TsArgs {
    NfGridHeader h;                        // the container grid (host-made): bounds + workspace offsets
    void* ws;
    const float* pos; const float* vel;
    float gx, gy, gz, dt, cell;
    float* pos_new; float* vel_new; float* feats4;
    int per;                               // particles per thread of workgroup 0: ceil(n / TS_BLOCK)
};

template <int PER>
__device__ __forceinline__ void ts_build(const TsArgs& S, char* lds)
{
    int* cells = (int*)lds;                // n_cells counters -> starts -> ends, then the scatter list (n_points)
    __shared__ int s_scan[TS_WAVES];
    __shared__ float s_red[TS_WAVES][6];
    __shared__ NfGridHeader hh;
    const NfGridHeader& h = S.h;
    char* b = (char*)S.ws;
    const int n = h.n_points, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // ---- integrate (all loads in flight at once: clamped indices, no guards); the exact bounds of the integrated cloud
    const float g[3] = {S.gx, S.gy, S.gz};
    float pv[PER][3], vv[PER][3];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int ic = min(u * TS_BLOCK + tid, n - 1);
#pragma unroll
        for (int d = 0; d < 3; ++d) { pv[u][d] = S.pos[3 * (size_t)ic + d]; vv[u][d] = S.vel[3 * (size_t)ic + d]; }
    }
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int i = u * TS_BLOCK + tid;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float v = vv[u][d];
            const float vn = v + g[d] * S.dt;                       // same expressions as k_trans_integrate
            pv[u][d] = pv[u][d] + (v + vn) / 2 * S.dt;
            vv[u][d] = vn;
        }
        if (i < n) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                S.pos_new[3 * (size_t)i + d] = pv[u][d];
                S.vel_new[3 * (size_t)i + d] = vv[u][d];
                lo[d] = fminf(lo[d], pv[u][d]); hi[d] = fmaxf(hi[d], pv[u][d]);
            }
            *(float4*)(S.feats4 + 4 * (size_t)i) = make_float4(1.f, vv[u][0], vv[u][1], vv[u][2]);
        }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { lo[d] = fminf(lo[d], __shfl_xor(lo[d], o, 64)); hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o, 64)); }
    if (lane == 0) { for (int d = 0; d < 3; ++d) { s_red[wv][d] = lo[d]; s_red[wv][3 + d] = hi[d]; } }
#if defined(TS_AB_STOP) && TS_AB_STOP == 1
    return;
#endif
    __syncthreads();
    if (wv == 0) {
        // the six bound columns, one per lane (min for 0..2, max for 3..5) over the waves' partial results
        float red = 0.f;
        if (lane < 6) {
            red = s_red[0][lane];
            for (int w2 = 1; w2 < TS_WAVES; ++w2) red = lane < 3 ? fminf(red, s_red[w2][lane]) : fmaxf(red, s_red[w2][lane]);
        }
        float bl[3], bu[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) { bl[d] = __shfl(red, d, 64); bu[d] = __shfl(red, 3 + d, 64); }
        if (lane == 0) {
            // The grid of THIS step hugs the cloud: the caller's bbox (the container, static: no host round trip) only bounds it —
            // a few hundred cells instead of the container's ~29 000, which every pass below (zero, scan, store) walks.  Points
            // outside the bbox land in border cells (the search stays exact, include/neurofluid_hip.h); the workspace offsets are
            // those of the caller's header, computed for the larger grid.
            NfGridHeader t = h;
            int ncell = 1;
            for (int d = 0; d < 3; ++d) {
                float l = bl[d], u2 = bu[d];
                t.pt_lo[d] = nf_f2ord(l); t.pt_hi[d] = nf_f2ord(u2);          // exact bounds of the points
                const float blo = h.origin[d], bhi = h.origin[d] + (float)h.dims[d] / h.inv_cell[d];
                l = fminf(fmaxf(l, blo), bhi); u2 = fminf(fmaxf(u2, blo), bhi);
                if (!(u2 >= l)) { l = blo; u2 = blo; }
                const float ext = u2 - l;
                float c = S.cell;
                if (ext / c > (float)(NF_GRID_MAX_DIM - 1)) c = ext / (float)(NF_GRID_MAX_DIM - 1);
                int dim = (int)floorf(ext / c) + 1;
                dim = max(1, min(dim, min(NF_GRID_MAX_DIM, h.dims[d])));
                t.origin[d] = l; t.inv_cell[d] = 1.0f / c; t.dims[d] = dim; t.sub0[d] = 0; t.subd[d] = dim;
                ncell *= dim;
            }
            t.n_cells = ncell;
            hh = t;
            *(NfGridHeader*)S.ws = t;
        }
    }
    __syncthreads();
    const int nc = hh.n_cells;
    int* cell_start = (int*)(b + hh.off_cell_start);
    int* tmp_list = cells + nc + 1;
    int* sorted_idx = (int*)(b + hh.off_sorted_idx);
    float4* sorted_pos = (float4*)(b + hh.off_sorted_pos);
    for (int c = tid; c < nc; c += TS_BLOCK) cells[c] = 0;
#if defined(TS_AB_STOP) && TS_AB_STOP == 2
    return;
#endif
    __syncthreads();
    // ---- count
    int mycell[PER], myslot[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int i = u * TS_BLOCK + tid;
        mycell[u] = -1; myslot[u] = 0;
        if (i < n) {
            const int cx = nf_cell_coord(pv[u][0], hh.origin[0], hh.inv_cell[0], hh.dims[0]);
            const int cy = nf_cell_coord(pv[u][1], hh.origin[1], hh.inv_cell[1], hh.dims[1]);
            const int cz = nf_cell_coord(pv[u][2], hh.origin[2], hh.inv_cell[2], hh.dims[2]);
            mycell[u] = (cz * hh.dims[1] + cy) * hh.dims[0] + cx;
            myslot[u] = atomicAdd(&cells[mycell[u]], 1);          // arrival slot inside the cell (the scatter below needs no second atomic)
        }
    }
#if defined(TS_AB_STOP) && TS_AB_STOP == 3
    return;
#endif
    __syncthreads();
    // ---- exclusive scan of the cell counts (each thread a contiguous run, block scan of the run sums)
    const int per = (nc + TS_BLOCK - 1) / TS_BLOCK;
    const int c0 = tid * per, c1 = min(c0 + per, nc);
    int run = 0;
    for (int c = c0; c < c1; ++c) run += cells[c];
    {
        int x = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
        if (lane == 63) s_scan[wv] = x;
        __syncthreads();
        if (wv == 0) {
            int s2 = lane < TS_WAVES ? s_scan[lane] : 0;
#pragma unroll
            for (int o = 1; o < TS_WAVES; o <<= 1) { const int y = __shfl_up(s2, o, 64); if (lane >= o) s2 += y; }
            if (lane < TS_WAVES) s_scan[lane] = s2;
        }
        __syncthreads();
        int base = (wv ? s_scan[wv - 1] : 0) + x - run;
        for (int c = c0; c < c1; ++c) { const int cnt = cells[c]; cells[c] = base; cell_start[c] = base; base += cnt; }
        if (tid == TS_BLOCK - 1) { cell_start[nc] = s_scan[TS_WAVES - 1]; cells[nc] = s_scan[TS_WAVES - 1]; }
    }
#if defined(TS_AB_STOP) && TS_AB_STOP == 4
    return;
#endif
    __syncthreads();
    // ---- scatter (arrival order): cells[] holds the START of every cell (+ the total behind the last)
#pragma unroll
    for (int u = 0; u < PER; ++u)
        if (mycell[u] >= 0) tmp_list[cells[mycell[u]] + myslot[u]] = u * TS_BLOCK + tid;
#if defined(TS_AB_STOP) && TS_AB_STOP == 5
    return;
#endif
    __syncthreads();
    // ---- stable order inside each cell: rank = number of same-cell points with a smaller original index
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int c = mycell[u];
        if (c < 0) continue;
        const int i = u * TS_BLOCK + tid;
        const int s2 = cells[c], e = cells[c + 1];
        int rank = 0;
        for (int t = s2; t < e; ++t) rank += (tmp_list[t] < i);
        sorted_idx[s2 + rank] = i;
        sorted_pos[s2 + rank] = make_float4(pv[u][0], pv[u][1], pv[u][2], __int_as_float(i));
    }
}

__global__ void __launch_bounds__(TS_BLOCK) k_trans_stage1(TsArgs S, TfArgs A)
{
    extern __shared__ __attribute__((aligned(16))) char ts_lds[];
    if (blockIdx.x == 0) {
        switch (S.per) {
            case 1: ts_build<1>(S, ts_lds); break;
            case 2: ts_build<2>(S, ts_lds); break;
            case 3: ts_build<3>(S, ts_lds); break;
            case 4: ts_build<4>(S, ts_lds); break;
            case 5: ts_build<5>(S, ts_lds); break;
            case 6: ts_build<6>(S, ts_lds); break;
            case 7: case 8: ts_build<8>(S, ts_lds); break;
            case 9: case 10: case 11: case 12: ts_build<12>(S, ts_lds); break;
            default: ts_build<16>(S, ts_lds); break;
        }
        return;
    }
    tf_body<1, true, TS_WAVES>(A, tf_carve<TS_WAVES>(ts_lds, 3), (int)blockIdx.x - 1, (int)gridDim.x - 1);
}

// ================================================================================================
// Round 4 (late): stage 1 for SMALL clouds = an ALL-PAIRS search instead of a cell grid.  For the clouds this model steps (4 913
// particles in the headline configuration) the grid was never about arithmetic: n^2 = 24 M distance tests are ~4 us of packed fp32
// on 256 CUs, while the single-workgroup counting sort (19.6 us) and the position -> cell -> ranges -> candidates chain of the sweep
// (14 + 12 us of the front kernel) are chains of dependent round trips.  Every workgroup integrates the WHOLE cloud into LDS
// (3 x 4 B x n: 59 KB; it writes pos_new / vel_new / feats for its own queries only), then a wave per query walks all candidates
// 128 at a time — two per lane, v_pk_* arithmetic in nf_dist2's operation order, so the test is bit-identical to the grid path's —
// and appends the hits to the pitched rows idx_f / d2_f in ascending index (the oracle's own order).  The same workgroups then run
// their share of the container half (it needs the static box grid only), re-using the LDS.  No fluid grid exists in this mode;
// k_trans_front_rows consumes the rows.  Limit: TB_MAX_POINTS (LDS); larger clouds take the grid path above.
// ================================================================================================
#define TB_MAX_POINTS 8192
typedef float tb_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int wv_of(int tid) { return __builtin_amdgcn_readfirstlane(tid >> 6); }
#define TB_LDS_FLOATS(npad) (3 * (npad) + 6 * ((npad) >> 4) + 6 * ((npad) >> 7) + TF_MAXP * TS_WAVES / 2)

// min / max over the 16 lanes of a DPP row: four row rotations (row_ror 1, 2, 4, 8), every lane ends up with the row's value — VALU
// only (__shfl_xor goes through the LDS crossbar: the butterflies of this reduction cost 12 us that way)
template <bool MAX>
__device__ __forceinline__ float tb_row_minmax(float v)
{
#define TB_ROR(n) { const float t = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x120 | (n), 0xf, 0xf, false)); \
                    v = MAX ? fmaxf(v, t) : fminf(v, t); }
    TB_ROR(1) TB_ROR(2) TB_ROR(4) TB_ROR(8)
#undef TB_ROR
    return v;
}

__device__ __forceinline__ void tb_search(const TsArgs& S, const TfArgs& A, char* lds, int blk, int nblk)
{
    const int n = A.n, npad = (n + 127) & ~127, nrow = npad >> 4, nchunk = npad >> 7;
    float* const sx = (float*)lds;
    float* const sy = sx + npad;
    float* const sz = sy + npad;
    float* const rbox = sz + npad;                                      // [6][nrow]: bounds of every 16 consecutive particles
    float* const cbox = rbox + 6 * nrow;                                // [6][nchunk]: bounds of every chunk of 128
    uint16_t* const hits = (uint16_t*)(cbox + 6 * nchunk) + (size_t)TF_MAXP * wv_of(threadIdx.x);   // [waves][TF_MAXP] neighbour indices (< 8 192)
    const int tid = threadIdx.x, lane = tid & 63, wv = wv_of(tid);
    const int q0 = (int)((unsigned)n * (unsigned)blk / (unsigned)nblk), q1 = (int)((unsigned)n * (unsigned)(blk + 1) / (unsigned)nblk);   // (n <= 8 192, nblk <= #CUs)
    // ---- integrate the whole cloud into LDS (all loads in flight at once: clamped indices); own queries also to memory
    {
        constexpr int PER = TB_MAX_POINTS / TS_BLOCK;
        const float g[3] = {S.gx, S.gy, S.gz};
        float pv[PER][3], vv[PER][3];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            if (u * TS_BLOCK >= npad) continue;                        // (uniform)
            const int ic = min(u * TS_BLOCK + tid, n - 1);
#pragma unroll
            for (int d = 0; d < 3; ++d) { pv[u][d] = S.pos[3 * (size_t)ic + d]; vv[u][d] = S.vel[3 * (size_t)ic + d]; }
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            if (u * TS_BLOCK >= npad) continue;
            const int i = u * TS_BLOCK + tid;
            float o[3], w[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float v = vv[u][d];
                const float vn = v + g[d] * S.dt;                       // same expressions as k_trans_integrate
                o[d] = pv[u][d] + (v + vn) / 2 * S.dt;
                w[d] = vn;
            }
            if (i >= q0 && i < q1) {
#pragma unroll
                for (int d = 0; d < 3; ++d) { S.pos_new[3 * (size_t)i + d] = o[d]; S.vel_new[3 * (size_t)i + d] = w[d]; }
                *(float4*)(S.feats4 + 4 * (size_t)i) = make_float4(1.f, w[0], w[1], w[2]);
            }
            if (u * TS_BLOCK + 64 * wv < npad) {                        // (uniform per wave: a wave's 64 particles = one half chunk)
                const bool live = i < n;                                // the padding never passes the radius test
                sx[i] = live ? o[0] : INFINITY; sy[i] = live ? o[1] : INFINITY; sz[i] = live ? o[2] : INFINITY;
                // bounds of every 16 consecutive particles (a DPP row), combined into the chunks' bounds below.  (All 16 waves of all
                // workgroups do this for the whole cloud: its VALU instructions, four waves to a SIMD, are what the phase costs.)
#ifdef TB_AB_NO_BOUNDS
                if (0)
#endif
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const float lo = tb_row_minmax<false>(live ? o[d] : INFINITY), hi = tb_row_minmax<true>(live ? o[d] : -INFINITY);
                    if ((lane & 15) == 0) {
                        const int r = i >> 4;
                        rbox[d * nrow + r] = lo;
                        rbox[(3 + d) * nrow + r] = hi;
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int t = tid; t < 6 * nchunk; t += TS_BLOCK) {                  // chunk c, column k: 8 rows of 16
        const int k = t / nchunk, c = t - k * nchunk;
        const float* src = rbox + k * nrow + 8 * c;
        float v = src[0];
#pragma unroll
        for (int e = 1; e < 8; ++e) v = k < 3 ? fminf(v, src[e]) : fmaxf(v, src[e]);
        cbox[k * nchunk + c] = v;
    }
    __syncthreads();
    // ---- a wave per query: lane c tests the bounds of chunk c (128 consecutive particles; index order is spatially coherent in
    // the clouds this model steps, so most chunks are out of reach), then the wave walks the chunks in reach.  Hits are parked in the
    // wave's LDS row and leave as whole rows
    __syncthreads();
    const int pitch = A.pitch_f, cap = min(pitch, TF_MAXP);
    const unsigned long long lt = (1ull << lane) - 1ull;
    const float r2 = A.r2;
#ifdef TB_AB_NO_SEARCH
    return;
#endif
    float bb[6];
    {
        const int c = min(lane, nchunk - 1);
#pragma unroll
        for (int k = 0; k < 6; ++k) bb[k] = cbox[k * nchunk + c];
    }
    const int qn = q1 - q0;
    for (int i = q0 + qn * wv / TS_WAVES; i < q0 + qn * (wv + 1) / TS_WAVES; ++i) {
        const float qx = sx[i], qy = sy[i], qz = sz[i];
        int cnt = 0;
#ifdef TB_AB_NO_CULL
        unsigned long long todo = nchunk >= 64 ? ~0ull : (1ull << nchunk) - 1ull;
#else
        // nf_box_dist2's argument: fp32 sub / mul / add are monotone, so the distance to the bounds never exceeds nf_dist2 to a point inside
        unsigned long long todo = __ballot(lane < nchunk && nf_box_dist2(bb, qx, qy, qz) <= r2);
#endif
        while (todo) {
            const int c0 = (__ffsll((long long)todo) - 1) << 7;
            todo &= todo - 1;
            const tb_f2 X = {sx[c0 + lane], sx[c0 + 64 + lane]}, Y = {sy[c0 + lane], sy[c0 + 64 + lane]}, Z = {sz[c0 + lane], sz[c0 + 64 + lane]};
            const tb_f2 dx = qx - X, dy = qy - Y, dz = qz - Z;          // nf_dist2: (q - p), mul + add chain in d = 0, 1, 2 order
            tb_f2 s2 = dx * dx;
            s2 = s2 + dy * dy;
            s2 = s2 + dz * dz;
            unsigned long long ma = __ballot(s2.x <= r2), mb = __ballot(s2.y <= r2);
            if (ma | mb) {
                // radius_search_ignore_query_points=True: a neighbour AT the query's position is not one (d2 == 0 happens once per
                // query — itself — and for underflowing offsets: the rare path looks at the coordinates)
                if (__ballot(s2.x == 0.f || s2.y == 0.f)) {
                    ma = __ballot(s2.x <= r2 && !(X.x == qx && Y.x == qy && Z.x == qz));
                    mb = __ballot(s2.y <= r2 && !(X.y == qx && Y.y == qy && Z.y == qz));
                }
                if ((ma >> lane) & 1) { const int w = cnt + __popcll(ma & lt); if (w < cap) hits[w] = (uint16_t)(c0 + lane); }
                cnt += __popcll(ma);
                if ((mb >> lane) & 1) { const int w = cnt + __popcll(mb & lt); if (w < cap) hits[w] = (uint16_t)(c0 + 64 + lane); }
                cnt += __popcll(mb);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        {
            int32_t* const irow = A.idx_f + (int64_t)i * pitch;
            float* const drow = A.d2_f + (int64_t)i * pitch;
            const int np = min(cnt, cap);
            // (the squared distance is evaluated again for the row: the same expression on the same operands, the same bits)
#pragma unroll
            for (int c = 0; c < 2; ++c)
                if (64 * c + lane < np) {
                    const int j = hits[64 * c + lane];
                    irow[64 * c + lane] = j;
                    drow[64 * c + lane] = nf_dist2(qx, qy, qz, sx[j], sy[j], sz[j]);
                }
        }
        if (lane == 0) {
            A.counts2[i] = cnt;
            A.num_nbrs[i] = (float)cnt;
            if (cnt > pitch) {
                atomicMax(A.overflow2, (unsigned long long)cnt);
                if (A.host_flag) { A.host_flag[0] = cnt; __threadfence_system(); }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
}

// workgroups [0, nsearch): the all-pairs search; the workgroups behind them: the CONTAINER half (cell-grid sweep of the static box grid
// from the state + conv0_obstacle) on 8 of their 16 waves — 70 KB of LDS, so that one of them shares a CU with a search workgroup: the
// container half is a chain of round trips (one particle in eight has container points in reach), the search is arithmetic
#define TB_BOX_WAVES 8
__global__ void __launch_bounds__(TS_BLOCK) k_trans_stage1b(TsArgs S, TfArgs A, int nsearch)
{
    extern __shared__ __attribute__((aligned(16))) char ts_lds[];
    if ((int)blockIdx.x < nsearch) { tb_search(S, A, ts_lds, blockIdx.x, nsearch); return; }
    if (threadIdx.x >= 64 * TB_BOX_WAVES) return;
    tf_body<1, true, TB_BOX_WAVES>(A, tf_carve<TB_BOX_WAVES>(ts_lds, 3), (int)blockIdx.x - nsearch, (int)gridDim.x - nsearch);
}

extern "C" int nf_trans_all_pairs_max_points(void) { return TB_MAX_POINTS; }

static int tb_launch(const TsArgs& S, const TfArgs& A, hipStream_t st)
{
    const int n = A.n, npad = (n + 127) & ~127, ncu = tf_ncu();
    int nsearch = (n + TS_WAVES - 1) / TS_WAVES;                         // a wave per query and round; one workgroup per CU
    if (nsearch > ncu) nsearch = ncu;
#ifdef TB_AB_BOX_IN_STAGE1
    const int iters = (n + ncu * TB_BOX_WAVES - 1) / (ncu * TB_BOX_WAVES);
    const int nbox = tf_box_blocks((n + TB_BOX_WAVES * iters - 1) / (TB_BOX_WAVES * iters), n);
#else
    const int nbox = 0;                                                  // (the container half rides in k_trans_front_rows' launch)
#endif
    const size_t lds_search = (size_t)TB_LDS_FLOATS(npad) * sizeof(float), lds_box = nbox ? TF_LDS_BYTES(3, TB_BOX_WAVES) : 0;
    const size_t lds = lds_search > lds_box ? lds_search : lds_box;
    static bool attr_set[64] = {};
    if (nf_first_use_on_device(attr_set))
        hipFuncSetAttribute((const void*)k_trans_stage1b, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(TF_LDS_BYTES(3, TB_BOX_WAVES) > (size_t)TB_LDS_FLOATS(TB_MAX_POINTS) * 4 ? TF_LDS_BYTES(3, TB_BOX_WAVES) : (size_t)TB_LDS_FLOATS(TB_MAX_POINTS) * 4));
    hipLaunchKernelGGL(k_trans_stage1b, dim3(nsearch + nbox), dim3(TS_BLOCK), lds, st, S, A, nsearch);
    return 0;
}

// nf_trans_step's first two launches (see the comment above); the caller's TfArgs carries both halves' arguments
static int ts_launch(const TsArgs& S0, const TfArgs& A, hipStream_t st)
{
    TsArgs S = S0;
    const int n = S.h.n_points;
    S.per = (n + TS_BLOCK - 1) / TS_BLOCK;
    const size_t lds_build = (size_t)(S.h.n_cells + 1 + n) * sizeof(int), lds_box = TF_LDS_BYTES(3, TS_WAVES);
    const size_t lds = lds_build > lds_box ? lds_build : lds_box;
    static bool attr_set[64] = {};
    if (nf_first_use_on_device(attr_set))
        hipFuncSetAttribute((const void*)k_trans_stage1, hipFuncAttributeMaxDynamicSharedMemorySize, (TS_MAX_LDS_INTS + 1) * (int)sizeof(int));
    // the container half: one workgroup of 16 waves per CU (the LDS of the grid build's workgroup sizes the launch), a particle
    // per wave and round (most waves leave after the sweep).  (The container half raises its overflow word from here; the
    // completion word is the fluid half's: the last launch of the front.)
    const int ncu = tf_ncu(), per = TS_WAVES;
    int iters = (n + ncu * per - 1) / (ncu * per);
    if (iters < 1) iters = 1;
#ifdef TS_AB_NO_BOX
    const int box_wg = 0;
#else
    const int box_wg = tf_box_blocks((n + per * iters - 1) / (per * iters), n);
#endif
    hipLaunchKernelGGL(k_trans_stage1, dim3(1 + box_wg), dim3(TS_BLOCK), lds, st, S, A);
    return 0;
}

int nf_trans_stage12(const nf_trans_step_t* s, const float* pos, const float* vel, float* num_nbrs, int32_t* host_flag3,
                     int step_id, nf_stream_t stream)
{
    NF_CHECK_ARG(s && pos && vel && num_nbrs, "null pointer");
    NF_CHECK_ARG(s->pitch_f >= 1 && s->pitch_f <= TF_MAXP && s->pitch_b >= 1 && s->pitch_b <= TF_MAXP, "pitch must be in [1, nf_trans_front_max_pitch()]");
    NF_CHECK_ARG(!host_flag3 || s->done_counter, "the completion word needs the workgroup counter");
    NF_CHECK_ARG(s->search >= 0 && s->search <= 2, "search: 0 auto, 1 cell grid, 2 all pairs");
    NF_CHECK_ARG(s->search != 2 || s->n <= TB_MAX_POINTS, "the all-pairs search serves clouds up to nf_trans_all_pairs_max_points()");
    const bool all_pairs = s->search == 2 || (s->search == 0 && s->n <= TB_MAX_POINTS);
    TsArgs S;
    memset(&S, 0, sizeof(S));
    if (!all_pairs) {
        size_t tot = 0;
        NF_CHECK_ARG(nf_grid_make_header(s->n, s->radius, s->bbox, &S.h, &tot) == NF_OK, "bad grid parameters");
        NF_CHECK_ARG(s->grid_ws_bytes >= tot, "workspace too small");
        NF_CHECK_ARG(s->n <= TP_BLOCK * TP_MAX_PER_THREAD && S.h.n_cells + s->n <= TS_MAX_LDS_INTS,
                     "cloud or grid too large for the fused step (use the multi-launch path)");
    }
    NF_CHECK_ARG(s->n > 0, "empty cloud");
    S.ws = s->grid_ws; S.pos = pos; S.vel = vel; S.gx = s->gravity[0]; S.gy = s->gravity[1]; S.gz = s->gravity[2]; S.dt = s->dt;
    S.pos_new = s->pos_new; S.vel_new = s->vel_new; S.feats4 = s->feats; S.per = 0; S.cell = s->radius;
    TfArgs A;
    memset(&A, 0, sizeof(A));
    A.grid[0] = s->grid_ws; A.grid[1] = s->box_grid; A.q = s->pos_new; A.pos = pos; A.vel = vel;
    A.gx = S.gx; A.gy = S.gy; A.gz = S.gz; A.dt = S.dt;
    A.feats_f = s->feats; A.feats_b = s->box_feats; A.n = s->n;
    A.r2 = s->radius * s->radius; A.extent = s->extent; A.use_window = s->use_window; A.pitch_f = s->pitch_f; A.pitch_b = s->pitch_b;
    A.relu_out = 1;
    A.counts2 = s->counts2; A.num_nbrs = num_nbrs; A.idx_f = s->idx_f; A.d2_f = s->d2_f; A.roff = s->roff; A.ent = s->ent;
    A.k_fluid = s->k_fluid; A.b_fluid = s->b_fluid; A.k_obst = s->k_obst; A.b_obst = s->b_obst;
    A.dense_w = s->dense0_w; A.dense_b = s->dense0_b; A.a0 = s->a0; A.overflow2 = (unsigned long long*)s->overflow2;
    // (no completion protocol in these launches: nf_trans_step's first layer raises the word when it starts)
    A.host_flag = (volatile int*)host_flag3; A.done_ctr = nullptr; A.step_id = step_id;
    if (all_pairs) {
        tb_launch(S, A, (hipStream_t)stream);
        NF_CHECK_LAUNCH();
        // one fluid workgroup per CU, every one with its equal share of the particles
#ifdef TF_AB_NF
        const int nf = TF_AB_NF;
#else
        const int nf = (s->n + TF_WAVES - 1) / TF_WAVES < tf_ncu() ? (s->n + TF_WAVES - 1) / TF_WAVES : tf_ncu();
#endif
#if defined(TB_AB_BOX_IN_STAGE1) || defined(TS_AB_NO_BOX)
        const int nb = 0;
#else
        const int nb = tf_box_blocks(tf_blocks(s->n, 1), s->n);
#endif
        hipLaunchKernelGGL(k_trans_front_rows, dim3(nf + nb), dim3(64 * TF_WAVES), 0, (hipStream_t)stream, A, nf);
        NF_CHECK_LAUNCH();
        return NF_OK;
    }
    ts_launch(S, A, (hipStream_t)stream);
    NF_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_trans_front, dim3(tf_blocks(s->n), 1), dim3(64 * TF_WAVES), 0, (hipStream_t)stream, A, 1);
    NF_CHECK_LAUNCH();
    return NF_OK;
}
