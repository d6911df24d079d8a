// nf_trans.hip — front half of the transition step (ParticleNet.forward, models/transmodel.py:151-163) for the inference path:
//
//   nf_trans_prepare   ONE workgroup: gravity integration (B1, :100-104) + the fluid cell grid of the integrated
//                      positions (counting sort in LDS, stable in original index) — replaces k_trans_integrate and the
//                      six k_grid_* / scan launches of nf_grid_build(with_firstk_lists = 0)
//   nf_trans_front     fixed-radius search of both clouds, the row-entry lists of the fluid pairs (what the G-free
//                      convolutions of nf_cconv_gf.hip consume) and layer 0 (conv0_obstacle, conv0_fluid, dense0_fluid)
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

template <int CIN>
__device__ __forceinline__ float tf_patch_times_filter(const float* __restrict__ patch, const float* __restrict__ Ks, int co, int half)
{
    // out[co] = sum_m patch[m] * Ks[m][co], m = node * CIN + ci; the two half-waves take the two halves of m
    const int M = 64 * CIN, m0 = half * (M / 2);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 4
    for (int m = m0; m < m0 + M / 2; m += 4) {
        a0 += patch[m] * Ks[m * 32 + co];
        a1 += patch[m + 1] * Ks[(m + 1) * 32 + co];
        a2 += patch[m + 2] * Ks[(m + 2) * 32 + co];
        a3 += patch[m + 3] * Ks[(m + 3) * 32 + co];
    }
    const float a = (a0 + a1) + (a2 + a3);
    return a + __shfl_xor(a, 32, 64);
}

// interpolation data of one pair, exactly k_pair_precompute (nf_cconv.hip): base node (bx, by, bz), fractions, window
struct TfPair { int j, bx, by, bz; float fx, fy, fz, imp; };

__device__ __forceinline__ TfPair tf_pair(const TfStage& st, int t, float qx, float qy, float qz, float scale, float inv_r2, int use_window)
{
    TfPair P;
    P.j = st.j[t];
    float x = (st.u.pos.px[t] - qx) * scale, y = (st.u.pos.py[t] - qy) * scale, z = (st.u.pos.pz[t] - qz) * scale;
    tr_ball_to_cube(x, y, z);
    P.imp = 1.f;
    if (use_window) { const float tt = 1.f - st.d2[t] * inv_r2; P.imp = fminf(fmaxf(tt * tt * tt, 0.f), 1.f); }
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

// LDS regions of the front body (static __shared__ in k_trans_front, carved from the dynamic block in k_trans_stage1)
struct TfLds {
    float* Ks;                          // 64 * CI * 32 floats: the layer-0 filter of this cloud
    TfStage* stage;                     // [TF_WAVES]
    int* rcnt; int* rcur; int* rbase;   // [TF_WAVES][16], [TF_WAVES][16], [TF_WAVES][17]
};
#define TF_LDS_BYTES(CI, NW) ((size_t)64 * (CI) * 32 * 4 + sizeof(TfStage) * (NW) + (size_t)(NW) * (16 + 16 + 17) * 4)

template <int NW>
__device__ __forceinline__ TfLds tf_carve(char* base, int ci)
{
    TfLds L;
    L.Ks = (float*)base; base += (size_t)64 * ci * 32 * 4;
    L.stage = (TfStage*)base; base += sizeof(TfStage) * NW;
    L.rcnt = (int*)base; base += NW * 16 * 4;
    L.rcur = (int*)base; base += NW * 16 * 4;
    L.rbase = (int*)base;
    return L;
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
#ifndef TF_AB_STRIDED
    // a wave's particles are CONSECUTIVE indices (index order is spatially coherent: their cell rows and candidates overlap in
    // the L1 / L2: 46.0 -> 44.2 us against particles strided by the launch width, round 4)
    const int per_wave = (A.n + nblk * NW - 1) / (nblk * NW);
    const int stride = 1;
    int i = (blk * NW + wv) * per_wave;
    const int i_end = min(A.n, i + per_wave);
#else
    const int stride = nblk * NW;
    int i = blk * NW + wv;
    const int i_end = A.n;
#endif

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
    float qx, qy, qz, q1x, q1y, q1z, q2x, q2y, q2z;
    int rs = 0, re = 0;
    bool first_particle = true;
    if (i < i_end) load_q(i, qx, qy, qz); else qx = qy = qz = 0.f;
#ifdef TF_AB_READAHEAD
    load_q(i + stride, q1x, q1y, q1z);
#else
    q1x = q1y = q1z = 0.f;
#endif
    {   // the layer-0 filter of this cloud: requested now, parked in LDS behind the first ranges request
        const float4* ksrc = (const float4*)(WHICH ? A.k_obst : A.k_fluid);
        constexpr int KN4 = 64 * CI * 32 / 4, PER = (KN4 + 64 * NW - 1) / (64 * NW);
        float4 kv[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int t = threadIdx.x + u * 64 * NW;
            kv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < KN4) kv[u] = ksrc[t];
        }
        if (g.sdx > 0 && i < i_end) load_ranges(qx, qy, qz, rs, re);
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int t = threadIdx.x + u * 64 * NW;
            if (t < KN4) ((float4*)L.Ks)[t] = kv[u];
        }
        __syncthreads();
    }
    for (; i < i_end; i += stride) {
        // requests for the particles ahead: the position two ahead, the row ranges one ahead
        int rs1 = 0, re1 = 0;
#ifndef TF_AB_READAHEAD
        // (default since the A/B of round 4: the read-ahead made the kernel SLOWER, 50.1 vs 45.9 us — the requests of the next
        // particles queue in front of the current one's candidates, and their registers cost the sweep its allocation)
        if (!first_particle) { load_q(i, qx, qy, qz); if (g.sdx > 0) load_ranges(qx, qy, qz, rs, re); }
        first_particle = false;
        q2x = q2y = q2z = 0.f;
#else
        load_q(i + 2 * stride, q2x, q2y, q2z);
        if (g.sdx > 0 && i + stride < A.n) load_ranges(q1x, q1y, q1z, rs1, re1);
#endif
        if (lane < 16) { rcnt[lane] = 0; rcur[lane] = 0; }
        int cnt = 0;
#ifdef TF_AB_SKIP_ALL
        if (lane == 0) A.a0[(size_t)i * 96 + WHICH] = (float)rs;
        qx = q1x; qy = q1y; qz = q1z; q1x = q2x; q1y = q2y; q1z = q2z; rs = rs1; re = re1;
        continue;
#endif
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
        qx = q1x; qy = q1y; qz = q1z; q1x = q2x; q1y = q2y; q1z = q2z; rs = rs1; re = re1;      // rotate the read-ahead
        float* orow = A.a0 + (size_t)i * 96;
        if (WHICH && np == 0) {
            // no container point in reach (the large majority of a fluid body): conv0_obstacle = its bias
            if (lane < 32) { const float v = A.b_obst[lane]; orow[lane] = A.relu_out ? fmaxf(v, 0.f) : v; }
            continue;
        }
        // ---- pass 2 (lane = pair, dense): interpolation data of up to two chunks of 64 pairs (kept in registers for the passes
        // below), the pitched neighbour rows, row counts
        TfPair P[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int t = 64 * c + lane;
            P[c].j = 0; P[c].bx = P[c].by = P[c].bz = 0; P[c].fx = P[c].fy = P[c].fz = P[c].imp = 0.f;
            if (t < np) {
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
        // ---- layer-0 patch P[node][ci] = sum_pairs w(pair, node) * feat[pair][ci], in rounds of 32 pairs WITHOUT float atomics
        // (ds_add_f32 with the same-address collisions of this scatter cost 44 of the kernel's 70 us): a pair's 8 (node, weight)
        // items are bucketed by node with integer LDS atomics (the returned slot orders a bucket), then lane = NODE walks its
        // bucket and accumulates in registers.
        float pacc[4] = {0.f, 0.f, 0.f, 0.f};
#ifndef TF_AB_SKIP_PATCH
        for (int t0 = 0; t0 < np; t0 += TF_CHUNK) {
            st.ccount[lane] = 0;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            // this round's pairs sit in lanes (t0 & 63) .. +31 of register set t0 >> 6; lanes 0..31 take them over
            const int srcl = (t0 & 63) + (lane & (TF_CHUNK - 1));
            TfPair Q;
            {
                const TfPair& R = P[0];
                const TfPair& R1 = P[1];
                const bool hi = t0 >= 64;
                Q.j = __shfl(hi ? R1.j : R.j, srcl, 64);
                Q.bx = __shfl(hi ? R1.bx : R.bx, srcl, 64); Q.by = __shfl(hi ? R1.by : R.by, srcl, 64); Q.bz = __shfl(hi ? R1.bz : R.bz, srcl, 64);
                Q.fx = __shfl(hi ? R1.fx : R.fx, srcl, 64); Q.fy = __shfl(hi ? R1.fy : R.fy, srcl, 64); Q.fz = __shfl(hi ? R1.fz : R.fz, srcl, 64);
                Q.imp = __shfl(hi ? R1.imp : R.imp, srcl, 64);
            }
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
                float4 f4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!WHICH) f4 = *(const float4*)(A.feats_f + 4 * (size_t)Q.j);
                else { f4.x = A.feats_b[3 * (size_t)Q.j]; f4.y = A.feats_b[3 * (size_t)Q.j + 1]; f4.z = A.feats_b[3 * (size_t)Q.j + 2]; }
                *(float4*)st.u.feat[lane] = f4;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            const int mycnt = st.ccount[lane];
            int x = mycnt;
#pragma unroll
            for (int o2 = 1; o2 < 64; o2 <<= 1) { const int y = __shfl_up(x, o2, 64); if (lane >= o2) x += y; }
            const int myoff = x - mycnt;
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
            int cmax = mycnt;
#pragma unroll
            for (int o2 = 1; o2 < 64; o2 <<= 1) cmax = max(cmax, __shfl_xor(cmax, o2, 64));
            for (int e = 0; e < cmax; ++e) {
                if (e < mycnt) {
                    const float w = st.iw[myoff + e];
                    const float4 f4 = *(const float4*)st.u.feat[st.it[myoff + e]];
                    pacc[0] = fmaf(w, f4.x, pacc[0]); pacc[1] = fmaf(w, f4.y, pacc[1]);
                    pacc[2] = fmaf(w, f4.z, pacc[2]); pacc[3] = fmaf(w, f4.w, pacc[3]);
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
#endif
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
            if (ci < CI) patch[lane * CI + ci] = pacc[ci];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        if (!WHICH) {
            // ---- pass 3: exclusive scan of the 16 row counts -> roff
            int c = lane < 16 ? rcnt[lane] : 0;
            int x = c;
#pragma unroll
            for (int o2 = 1; o2 < 16; o2 <<= 1) { const int y = __shfl_up(x, o2, 64); if (lane >= o2) x += y; }
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
        // ---- layer 0: patch x filter (+ the Linear branch on the particle's own features)
        const int co = lane & 31, half = lane >> 5;
#ifdef TF_AB_SKIP_GEMV
        if (lane == 0) orow[WHICH] = patch[5];
        continue;
#endif
        if (!WHICH) {
            const float af = tf_patch_times_filter<4>(patch, Ks, co, half);
            if (half == 0) { const float v = af + A.b_fluid[co]; orow[32 + co] = A.relu_out ? fmaxf(v, 0.f) : v; }
            else {
                float s2 = A.dense_b[co];
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) s2 += A.feats_f[(size_t)i * 4 + ci] * A.dense_w[co * 4 + ci];
                orow[64 + co] = A.relu_out ? fmaxf(s2, 0.f) : s2;
            }
        } else {
            const float ao = tf_patch_times_filter<3>(patch, Ks, co, half);
            if (half == 0) { const float v = ao + A.b_obst[co]; orow[co] = A.relu_out ? fmaxf(v, 0.f) : v; }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();          // the next particle of this wave reuses the stage / patch / counters
    }
}

// the completion word: the LAST workgroup of the launch to arrive writes step_id into the pinned host word
__device__ __forceinline__ void tf_arrive(const TfArgs& A, unsigned total)
{
    if (!A.host_flag) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();                // this workgroup's overflow words (if any) before its arrival
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
static int tf_blocks(int n)
{
    const int ncu = tf_ncu(), per = TF_WAVES, iters = (n + ncu * per - 1) / (ncu * per);
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
    hipLaunchKernelGGL(k_trans_front, dim3(tf_blocks(n), 2), dim3(64 * TF_WAVES), 0, (hipStream_t)stream, A, 3);
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
    const int box_wg = (n + per * iters - 1) / (per * iters);
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
    TsArgs S;
    size_t tot = 0;
    NF_CHECK_ARG(nf_grid_make_header(s->n, s->radius, s->bbox, &S.h, &tot) == NF_OK, "bad grid parameters");
    NF_CHECK_ARG(s->grid_ws_bytes >= tot, "workspace too small");
    NF_CHECK_ARG(s->n > 0 && s->n <= TP_BLOCK * TP_MAX_PER_THREAD && S.h.n_cells + s->n <= TS_MAX_LDS_INTS,
                 "cloud or grid too large for the fused step (use the multi-launch path)");
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
    A.host_flag = (volatile int*)host_flag3; A.done_ctr = s->done_counter; A.step_id = step_id;
    ts_launch(S, A, (hipStream_t)stream);
    NF_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_trans_front, dim3(tf_blocks(s->n), 1), dim3(64 * TF_WAVES), 0, (hipStream_t)stream, A, 1);
    NF_CHECK_LAUNCH();
    return NF_OK;
}
