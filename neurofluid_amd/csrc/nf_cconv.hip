// nf_cconv.hip — Lagrangian transition model (models/transmodel.py): integration, continuous
// convolution (Open3D ContinuousConv contract, call sites :86-95, :116-118, :125) and update.
//
// Formulation: "transform, then gather".
//   G[j][cell][co] = sum_ci x[j][ci] * K[cell][ci][co]          (dense fp32-MFMA GEMM, N x Cin x 64*Cout)
//   y[i][co]       = sum_{j in N(i)} sum_{8 corners c} w_ij * t_c(i,j) * G[j][cell_c(i,j)][co]
// which equals Open3D's "gather a (64*Cin) patch per point, then GEMM" up to summation order and
// does the same FLOPs, but turns the neighbour-dependent part into a pure weighted row gather with
// 256-byte coalesced reads.  The per-pair interpolation data (8 cells + 8 weights, window folded in)
// depends only on positions, so it is computed ONCE per step and shared by all fluid->fluid layers
// (the reference rebuilds its hash table and re-derives it in each of 5 convs).
// The dense (Linear) branch of each layer rides along as 1 extra "cell" of the same GEMM.
// Layers with Cin <= 4 (conv0_fluid, conv0_obstacle) evaluate directly with the filter in LDS.
#include "nf_common.h"
#include <math.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------
// B1: integrate / update  (models/transmodel.py:100-104, :144-148)
// ------------------------------------------------------------------------------------------------
__global__ void k_trans_integrate(const float* __restrict__ pos, const float* __restrict__ vel, float gx, float gy,
                                  float gz, float dt, int n, float* __restrict__ pos_new, float* __restrict__ vel_new,
                                  float* __restrict__ feats4)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float g[3] = {gx, gy, gz};
    float vn[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float v = vel[3 * i + d];
        vn[d] = v + g[d] * dt;
        pos_new[3 * i + d] = pos[3 * i + d] + (v + vn[d]) / 2 * dt;
        vel_new[3 * i + d] = vn[d];
    }
    if (feats4) {  // fluid_feats = [1, vel_new]  (:111-114)
        feats4[4 * i] = 1.f; feats4[4 * i + 1] = vn[0]; feats4[4 * i + 2] = vn[1]; feats4[4 * i + 3] = vn[2];
    }
}

__global__ void k_trans_update(const float* __restrict__ pos, const float* __restrict__ pos_new,
                               const float* __restrict__ y3, float scale, float dt, int n, float* __restrict__ pos_c,
                               float* __restrict__ vel_c)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * n) return;
    float pc = pos_new[i] + scale * y3[i];   // pos_correction = (1/128) * ans_convs[-1]  (:141)
    pos_c[i] = pc;
    vel_c[i] = (pc - pos[i]) / dt;
}

extern "C" int nf_trans_integrate(const float* pos, const float* vel, const float gravity[3], float dt, int n,
                                  float* pos_new, float* vel_new, float* feats4, nf_stream_t stream)
{
    NF_CHECK_ARG(pos && vel && gravity && pos_new && vel_new, "null pointer");
    if (n <= 0) return NF_OK;
    hipLaunchKernelGGL(k_trans_integrate, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, pos, vel, gravity[0],
                       gravity[1], gravity[2], dt, n, pos_new, vel_new, feats4);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_trans_update(const float* pos, const float* pos_new, const float* y3, float scale, float dt, int n,
                               float* pos_c, float* vel_c, nf_stream_t stream)
{
    NF_CHECK_ARG(pos && pos_new && y3 && pos_c && vel_c, "null pointer");
    if (n <= 0) return NF_OK;
    hipLaunchKernelGGL(k_trans_update, dim3((3 * n + 255) / 256), dim3(256), 0, (hipStream_t)stream, pos, pos_new, y3,
                       scale, dt, n, pos_c, vel_c);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// ------------------------------------------------------------------------------------------------
// pair interpolation data: ball -> cylinder -> cube (volume preserving), align_corners, trilinear
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ball_to_cube(float& x, float& y, float& z)
{
    // sphere -> cylinder
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
    // cylinder -> cube
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

// one wave per CSR row; lanes stride over the row's entries
__global__ void __launch_bounds__(256) k_pair_precompute(const float* __restrict__ inp_pos, const float* __restrict__ out_pos,
                                                         const int64_t* __restrict__ row_splits,
                                                         const int32_t* __restrict__ nbr, const float* __restrict__ d2,
                                                         int n_out, float extent, int use_window, int negate,
                                                         float* __restrict__ pw, uint8_t* __restrict__ pc)
{
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (row >= n_out) return;
    float ox = out_pos[3 * row], oy = out_pos[3 * row + 1], oz = out_pos[3 * row + 2];
    const float radius = 0.5f * extent, inv_r2 = 1.f / (radius * radius), scale = 2.f / extent;
    for (int64_t p = row_splits[row] + lane; p < row_splits[row + 1]; p += 64) {
        int j = nbr[p];
        float x = (inp_pos[3 * j] - ox) * scale, y = (inp_pos[3 * j + 1] - oy) * scale, z = (inp_pos[3 * j + 2] - oz) * scale;
        if (negate) { x = -x; y = -y; z = -z; }   // the same pair seen from the neighbour (transposed operator)
        ball_to_cube(x, y, z);
        float imp = 1.f;
        if (use_window) {  // _window_poly6(d2 / radius^2) = clamp((1-R)^3, 0, 1)   (models/transmodel.py:73-77)
            float t = 1.f - d2[p] * inv_r2;
            imp = fminf(fmaxf(t * t * t, 0.f), 1.f);
        }
        float c[3] = {(x + 1.f) * 1.5f, (y + 1.f) * 1.5f, (z + 1.f) * 1.5f};  // [-1,1] -> [0,3] (4 cells, align_corners)
        int i0[3];
        float f[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float cc = fminf(fmaxf(c[d], 0.f), 3.f);
            float fl = fminf(floorf(cc), 2.f);
            i0[d] = (int)fl;
            f[d] = cc - fl;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
            float w = (dx ? f[0] : 1.f - f[0]) * (dy ? f[1] : 1.f - f[1]) * (dz ? f[2] : 1.f - f[2]);
            pw[p * 8 + k] = imp * w;
            pc[p * 8 + k] = (uint8_t)(((i0[2] + dz) * 4 + (i0[1] + dy)) * 4 + (i0[0] + dx));
        }
    }
}

extern "C" int nf_cconv_pairs(const float* inp_pos, const float* out_pos, const int64_t* row_splits, const int32_t* nbr,
                              const float* dist2, int n_out, float extent, int use_window, int negate, float* pair_w,
                              uint8_t* pair_cell, nf_stream_t stream)
{
    NF_CHECK_ARG(inp_pos && out_pos && row_splits && pair_w && pair_cell, "null pointer");
    NF_CHECK_ARG(extent > 0.f, "bad extent");
    if (n_out <= 0) return NF_OK;
    hipLaunchKernelGGL(k_pair_precompute, dim3((n_out + 3) / 4), dim3(256), 0, (hipStream_t)stream, inp_pos, out_pos,
                       row_splits, nbr, dist2, n_out, extent, use_window, negate, pair_w, pair_cell);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// ------------------------------------------------------------------------------------------------
// direct continuous convolution for small Cin (<= 4), Cout = 32: filter (64 x Cin x 32) in LDS,
// one wave per output point, lane = (entry parity, co).  Optional Linear branch on the query's own
// features (dense0_fluid, models/transmodel.py:117).
// ------------------------------------------------------------------------------------------------
template <int CIN>
__global__ void __launch_bounds__(256) k_cconv_small(const float* __restrict__ feats, const int64_t* __restrict__ row_splits,
                                                     const int32_t* __restrict__ nbr, const float* __restrict__ pw,
                                                     const uint8_t* __restrict__ pc, const float* __restrict__ kernel,
                                                     const float* __restrict__ bias, int n_out, float* __restrict__ out,
                                                     int ld_out, int col_off, const float* __restrict__ self_feats,
                                                     const float* __restrict__ dense_w, const float* __restrict__ dense_b,
                                                     int dense_col_off)
{
    __shared__ float Ks[64 * CIN * 32];
    for (int t = threadIdx.x; t < 64 * CIN * 32; t += 256) Ks[t] = kernel[t];
    __syncthreads();
    const int lane = threadIdx.x & 63, co = lane & 31, half = lane >> 5;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < n_out; row += gridDim.x * 4) {
        float acc = 0.f;
        for (int64_t p = row_splits[row] + half; p < row_splits[row + 1]; p += 2) {
            int j = nbr[p];
            float fj[CIN];
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) fj[ci] = feats[(size_t)j * CIN + ci];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float w = pw[p * 8 + k];
                const float* kc = Ks + (int)pc[p * 8 + k] * CIN * 32 + co;
                float s = 0.f;
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) s += fj[ci] * kc[ci * 32];
                acc += w * s;
            }
        }
        acc += __shfl_xor(acc, 32, 64);
        if (half == 0) out[(size_t)row * ld_out + col_off + co] = acc + bias[co];
        if (dense_w && half == 1) {
            float s = dense_b[co];
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) s += self_feats[(size_t)row * CIN + ci] * dense_w[co * CIN + ci];
            out[(size_t)row * ld_out + dense_col_off + co] = s;
        }
    }
}

extern "C" int nf_cconv_small(const float* feats, int cin, const int64_t* row_splits, const int32_t* nbr,
                              const float* pair_w, const uint8_t* pair_cell, const float* kernel, const float* bias,
                              int n_out, float* out, int ld_out, int col_off, const float* self_feats,
                              const float* dense_w, const float* dense_b, int dense_col_off, nf_stream_t stream)
{
    NF_CHECK_ARG(feats && row_splits && kernel && bias && out, "null pointer");
    NF_CHECK_ARG(cin == 3 || cin == 4, "cin must be 3 or 4 (Cout is 32)");
    if (n_out <= 0) return NF_OK;
    int blocks = (n_out + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    hipStream_t st = (hipStream_t)stream;
    if (cin == 3)
        hipLaunchKernelGGL(k_cconv_small<3>, dim3(blocks), dim3(256), 0, st, feats, row_splits, nbr, pair_w, pair_cell, kernel,
                           bias, n_out, out, ld_out, col_off, self_feats, dense_w, dense_b, dense_col_off);
    else
        hipLaunchKernelGGL(k_cconv_small<4>, dim3(blocks), dim3(256), 0, st, feats, row_splits, nbr, pair_w, pair_cell, kernel,
                           bias, n_out, out, ld_out, col_off, self_feats, dense_w, dense_b, dense_col_off);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// ------------------------------------------------------------------------------------------------
// G = act(A) * [K_flat | W_dense^T]   fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32)
//   A: (M x Cin) row-major, optional ReLU on load (models/transmodel.py:124 inp_feats = relu(prev))
//   B(k, n): n < 64*Cout -> kernel[(n / Cout) * Cin * Cout + k * Cout + n % Cout]   (filter (4,4,4,Cin,Cout))
//            else        -> dense_w[(n - 64*Cout) * Cin + k]                         (Linear weight [Cout][Cin])
//   G: (M x 65*Cout) row-major.
// 128x128 tile per 4-wave workgroup, 2x2 MFMA tiles per wave, K staged through LDS in 32-deep slabs
// stored k-major so that A/B fragment reads are lane-contiguous (conflict-free).
// ------------------------------------------------------------------------------------------------
#define GT_M 128
#define GT_N 128
#define GT_K 32
#define GT_AP 33      // A slab is m-major with an odd pitch: fragment reads (32 rows, fixed k) and the staged stores are conflict-free
#define GT_BP (GT_N + 4)

// Slab loaders: each thread brings 4 x 16 B of A and 4 x 16 B of B per 32-deep slab into registers (issued before
// the MFMAs of the previous slab, so the L2 latency hides behind them), then parks them in LDS.
template <bool VA>
__device__ __forceinline__ void gemm_load_a(const float* __restrict__ A, int M, int cin, int relu, int m0, int k0, int tid,
                                            float4 (&ra)[4])
{
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int t = tid + 256 * u, m = t >> 3, kq = t & 7;        // 128 rows x 8 quads
        const int gm = m0 + m, gk = k0 + 4 * kq;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gm < M) {
            const float* src = A + (size_t)gm * cin + gk;
            if (VA) { if (gk < cin) v = *(const float4*)src; }
            else {
                if (gk < cin) v.x = src[0];
                if (gk + 1 < cin) v.y = src[1];
                if (gk + 2 < cin) v.z = src[2];
                if (gk + 3 < cin) v.w = src[3];
            }
        }
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        ra[u] = v;
    }
}

template <bool VB>
__device__ __forceinline__ void gemm_load_b(const float* __restrict__ kernel, const float* __restrict__ dense_w, int cin,
                                            int cout, int n0, int k0, int tid, float4 (&rb)[4])
{
    const int ntot = 65 * cout;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int t = tid + 256 * u, k = t >> 5, nq = t & 31;       // 32 k x 32 quads
        const int gk = k0 + k, gn = n0 + 4 * nq;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gk < cin) {
            if (VB && gn + 3 < 64 * cout) {                          // a quad never straddles two filter cells (cout % 4 == 0)
                v = *(const float4*)(kernel + (size_t)(gn / cout) * cin * cout + (size_t)gk * cout + gn % cout);
            } else {
                float e[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int g = gn + c;
                    e[c] = 0.f;
                    if (g < 64 * cout) e[c] = kernel[(size_t)(g / cout) * cin * cout + (size_t)gk * cout + g % cout];
                    else if (g < ntot) e[c] = dense_w[(size_t)(g - 64 * cout) * cin + gk];
                }
                v = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
        rb[u] = v;
    }
}

template <bool VA, bool VB>
__global__ void __launch_bounds__(256) k_cconv_gemm(const float* __restrict__ A, int M, int cin, int cout, int relu,
                                                    const float* __restrict__ kernel, const float* __restrict__ dense_w,
                                                    float* __restrict__ G)
{
    __shared__ float As[GT_M * GT_AP];
    __shared__ float Bs[GT_K][GT_BP];
    const int ntot = 65 * cout;
    const int m0 = blockIdx.y * GT_M, n0 = blockIdx.x * GT_N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float4 ra[4], rb[4];
    gemm_load_a<VA>(A, M, cin, relu, m0, 0, tid, ra);
    gemm_load_b<VB>(kernel, dense_w, cin, cout, n0, 0, tid, rb);
    for (int k0 = 0; k0 < cin; k0 += GT_K) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = tid + 256 * u;
            float* da = As + (t >> 3) * GT_AP + 4 * (t & 7);
            da[0] = ra[u].x; da[1] = ra[u].y; da[2] = ra[u].z; da[3] = ra[u].w;
            *(float4*)&Bs[t >> 5][4 * (t & 31)] = rb[u];
        }
        __syncthreads();
        if (k0 + GT_K < cin) {      // next slab in flight while this one is multiplied
            gemm_load_a<VA>(A, M, cin, relu, m0, k0 + GT_K, tid, ra);
            gemm_load_b<VB>(kernel, dense_w, cin, cout, n0, k0 + GT_K, tid, rb);
        }
#pragma unroll
        for (int kk = 0; kk < GT_K; kk += 2) {
            const int kr = kk + (lane >> 5), c = lane & 31;
            const float a0 = As[(wm + c) * GT_AP + kr], a1 = As[(wm + 32 + c) * GT_AP + kr];
            const float b0 = Bs[kr][wn + c], b1 = Bs[kr][wn + 32 + c];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
    // D layout: lane -> column j = lane & 31, rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int m = m0 + wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                int n = n0 + wn + 32 * b + (lane & 31);
                if (m < M && n < ntot) G[(size_t)m * ntot + n] = acc[a][b][r];
            }
}

extern "C" int nf_cconv_transform(const float* A, int M, int cin, int cout, int relu, const float* kernel,
                                  const float* dense_w, float* G, nf_stream_t stream)
{
    NF_CHECK_ARG(A && kernel && dense_w && G, "null pointer");
    NF_CHECK_ARG(cin >= 1 && cout >= 1, "bad channel counts");
    if (M <= 0) return NF_OK;
    int ntot = 65 * cout;
    dim3 grid((ntot + GT_N - 1) / GT_N, (M + GT_M - 1) / GT_M);
    const bool va = (cin % 4) == 0, vb = (cout % 4) == 0;
    hipStream_t st = (hipStream_t)stream;
    if (va && vb) hipLaunchKernelGGL((k_cconv_gemm<true, true>), grid, dim3(256), 0, st, A, M, cin, cout, relu, kernel, dense_w, G);
    else if (va) hipLaunchKernelGGL((k_cconv_gemm<true, false>), grid, dim3(256), 0, st, A, M, cin, cout, relu, kernel, dense_w, G);
    else if (vb) hipLaunchKernelGGL((k_cconv_gemm<false, true>), grid, dim3(256), 0, st, A, M, cin, cout, relu, kernel, dense_w, G);
    else hipLaunchKernelGGL((k_cconv_gemm<false, false>), grid, dim3(256), 0, st, A, M, cin, cout, relu, kernel, dense_w, G);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// ------------------------------------------------------------------------------------------------
// gather:  y[i][co] = sum_pairs sum_corners pw * G[j][cell*Cout + co] + G[i][64*Cout + co]
//                     + bias_conv[co] + bias_dense[co] (+ residual[i][co])
// one wave per output point; the row's (j, 8 cells, 8 weights) are staged through LDS 64 entries
// at a time, then every lane walks them for its output channel(s).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_cconv_gather(const float* __restrict__ G, int cout,
                                                      const int64_t* __restrict__ row_splits,
                                                      const int32_t* __restrict__ nbr, const float* __restrict__ pw,
                                                      const uint8_t* __restrict__ pc, const float* __restrict__ bias_c,
                                                      const float* __restrict__ bias_d, const float* __restrict__ residual,
                                                      int n_out, float* __restrict__ out)
{
    __shared__ int s_j[4][64];
    __shared__ float s_w[4][64 * 8];
    __shared__ int s_c[4][64 * 8];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ntot = 65 * cout;
    // lanes map to (sub, co): sub-groups split the staged entries when cout < 64
    const int groups = cout >= 64 ? 1 : (64 / cout >= 1 ? 64 / cout : 1);
    const int co = lane % cout, sub = lane / cout;
    const bool lane_on = sub < groups && cout <= 64;
    // XCD-aware row order: workgroup b runs on XCD b % 8, so give every XCD a CONTIGUOUS eighth of the rows — neighbouring
    // rows share most of their (j, cell) lines of G, which then hit in that XCD's L2 (watercube conv1: 44 -> 35 us)
    const int vblk = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    for (int row = vblk * 4 + wv; row < n_out; row += gridDim.x * 4) {
        float acc = 0.f;
        const int64_t s = row_splits[row], e = row_splits[row + 1];
        for (int64_t base = s; base < e; base += 64) {
            int cnt = (int)((e - base) < 64 ? (e - base) : 64);
            if (lane < cnt) {
                int64_t p = base + lane;
                s_j[wv][lane] = nbr[p];
#pragma unroll
                for (int k = 0; k < 8; ++k) { s_w[wv][lane * 8 + k] = pw[p * 8 + k]; s_c[wv][lane * 8 + k] = pc[p * 8 + k]; }
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): LDS writes visible wave-wide
            if (lane_on)
                for (int t = sub; t < cnt; t += groups) {
                    const float* gr = G + (size_t)s_j[wv][t] * ntot + co;
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc += s_w[wv][t * 8 + k] * gr[s_c[wv][t * 8 + k] * cout];
                }
            __builtin_amdgcn_wave_barrier();
        }
        // reduce the sub-groups (lanes with equal co)
        if (groups > 1) {
            // gather partial sums through LDS (cout need not be a power of two, e.g. 3)
            s_w[wv][lane] = lane_on ? acc : 0.f;
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (lane < cout) {
                float t = 0.f;
                for (int g2 = 0; g2 < groups; ++g2) t += s_w[wv][g2 * cout + lane];
                acc = t;
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (lane < cout) {
            float v = acc + G[(size_t)row * ntot + 64 * cout + lane] + bias_c[lane] + bias_d[lane];
            if (residual) v += residual[(size_t)row * cout + lane];
            out[(size_t)row * cout + lane] = v;
        }
    }
}

extern "C" int nf_cconv_gather(const float* G, int cout, const int64_t* row_splits, const int32_t* nbr, const float* pair_w,
                               const uint8_t* pair_cell, const float* bias_conv, const float* bias_dense, const float* residual,
                               int n_out, float* out, nf_stream_t stream)
{
    NF_CHECK_ARG(G && row_splits && bias_conv && bias_dense && out, "null pointer");
    NF_CHECK_ARG(cout >= 1 && cout <= 64, "cout must be in [1,64]");
    if (n_out <= 0) return NF_OK;
    int blocks = (n_out + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    blocks = (blocks + 7) & ~7;      // the kernel's XCD-aware row order needs a multiple of 8
    hipLaunchKernelGGL(k_cconv_gather, dim3(blocks), dim3(256), 0, (hipStream_t)stream, G, cout, row_splits, nbr, pair_w,
                       pair_cell, bias_conv, bias_dense, residual, n_out, out);
    NF_CHECK_LAUNCH();
    return NF_OK;
}


// ================================================================================================
// backward (B8).  Open3D's continuous_conv is differentiated w.r.t. filter and input features only (not
// positions); the weight/feature GEMMs are plain GEMMs done by the caller (library GEMM), the neighbour-
// dependent parts are these kernels.
// ================================================================================================

// dG[j][cell][co] = sum over entries (j <- i) of the TRANSPOSED pair cache: tpw * dy[i][co]   (fluid<->fluid is
// symmetric: i in N(j) <=> j in N(i), so row j of the same CSR lists exactly the rows i that gathered from j; the
// transposed cache holds the pair's interpolation data as seen from i).  dG[j][64][co] = dy[j][co] (Linear branch).
// One wave per row, a private 64 x Cout tile in LDS (ds_add_f32 from one wave only: in order), no global atomics, deterministic.
__global__ void __launch_bounds__(256) k_cconv_gather_t(const float* __restrict__ dy, int cout,
                                                        const int64_t* __restrict__ row_splits,
                                                        const int32_t* __restrict__ nbr, const float* __restrict__ tpw,
                                                        const uint8_t* __restrict__ tpc, int n, float* __restrict__ dG)
{
    // Lane t of the wave holds pair t of the current chunk of 64 (neighbour index, 8 corner weights, 8 corner cells packed in
    // two words); the loop over the pairs reads them with v_readlane into scalar registers.  (Staged through LDS instead,
    // every pair cost three dependent LDS round trips before its first tile update — 1 100 cycles per pair with one wave
    // per SIMD; and ds_add_f32 in place of the read-add-write runs at a fraction of the plain LDS rate: 405 vs 87 us.)
    extern __shared__ float tile_all[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* tile = tile_all + wv * 64 * cout;
    const int ntot = 65 * cout;
    const int wpb = blockDim.x >> 6;
    for (int row = blockIdx.x * wpb + wv; row < n; row += gridDim.x * wpb) {
        for (int t = lane; t < 64 * cout; t += 64) tile[t] = 0.f;
        const int64_t s = row_splits[row], e = row_splits[row + 1];
        for (int64_t base = s; base < e; base += 64) {
            const int cnt = __builtin_amdgcn_readfirstlane((int)((e - base) < 64 ? (e - base) : 64));
            int jl = 0, c03 = 0, c47 = 0;
            float w8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (lane < cnt) {
                const int64_t p = base + lane;
                jl = nbr[p];
                unsigned long long seen = 0ull;
                bool distinct = true;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    w8[k] = tpw[p * 8 + k];
                    const int c = tpc[p * 8 + k];
                    if (k < 4) c03 |= c << (8 * k); else c47 |= c << (8 * (k - 4));
                    distinct &= !((seen >> c) & 1ull);
                    seen |= 1ull << c;
                }
                // the 8 corner cells of a pair are pairwise distinct unless the interpolation was clamped at the filter's
                // border: then (and only then) its updates must run one after the other
                if (distinct) c47 |= 1 << 31;
            }
            for (int co = lane; co < cout; co += 64) {
                // dy rows of 8 pairs at a time, the next 8 in flight behind the LDS work of these
                float g8[8], g8n[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) g8[u] = u < cnt ? dy[(size_t)__builtin_amdgcn_readlane(jl, u) * cout + co] : 0.f;
                for (int t0 = 0; t0 < cnt; t0 += 8) {
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        g8n[u] = t0 + 8 + u < cnt ? dy[(size_t)__builtin_amdgcn_readlane(jl, t0 + 8 + u) * cout + co] : 0.f;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int t = t0 + u;
                        if (t >= cnt) break;
                        const float g = g8[u];
                        const int a03 = __builtin_amdgcn_readlane(c03, t), a47 = __builtin_amdgcn_readlane(c47, t);
                        int a[8];
                        float wk[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            a[k] = (((k < 4 ? a03 >> (8 * k) : a47 >> (8 * (k - 4))) & 63)) * cout + co;
                            wk[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w8[k]), t));
                        }
                        if (a47 < 0) {          // distinct cells: 8 independent reads, 8 FMAs, 8 writes (the same sums as the chain)
                            float v[8];
#pragma unroll
                            for (int k = 0; k < 8; ++k) v[k] = tile[a[k]];
#pragma unroll
                            for (int k = 0; k < 8; ++k) v[k] += wk[k] * g;
#pragma unroll
                            for (int k = 0; k < 8; ++k) tile[a[k]] = v[k];
                        } else {
#pragma unroll
                            for (int k = 0; k < 8; ++k) tile[a[k]] += wk[k] * g;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) g8[u] = g8n[u];
                }
            }
        }
        float* out = dG + (size_t)row * ntot;
        for (int t = lane; t < 64 * cout; t += 64) out[t] = tile[t];
        for (int co = lane; co < cout; co += 64) out[64 * cout + co] = dy[(size_t)row * cout + co];
    }
}

// cout <= 4 (the last layer: 3 channels): lane = CELL.  With lane = channel only 3 lanes of the wave above work; here every
// lane owns one of the 64 cells, finds its weight in a pair's 8 corners by comparison with the broadcast cell ids and keeps
// COUT accumulators in registers — no LDS, ~25 instructions per pair instead of ~150 (31 -> ~10 us per launch).
template <int COUT>
__global__ void __launch_bounds__(64) k_cconv_gather_t_small(const float* __restrict__ dy, const int64_t* __restrict__ row_splits,
                                                             const int32_t* __restrict__ nbr, const float* __restrict__ tpw,
                                                             const uint8_t* __restrict__ tpc, int n, float* __restrict__ dG)
{
    const int lane = threadIdx.x;
    for (int row = blockIdx.x; row < n; row += gridDim.x) {
        const int64_t s = row_splits[row], e = row_splits[row + 1];
        float acc[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
        for (int64_t base = s; base < e; base += 64) {
            const int cnt = __builtin_amdgcn_readfirstlane((int)((e - base) < 64 ? (e - base) : 64));
            int jl = 0, c03 = 0, c47 = 0;
            float w8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, g[COUT];
#pragma unroll
            for (int co = 0; co < COUT; ++co) g[co] = 0.f;
            if (lane < cnt) {
                const int64_t p = base + lane;
                jl = nbr[p];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    w8[k] = tpw[p * 8 + k];
                    const int c = tpc[p * 8 + k];
                    if (k < 4) c03 |= c << (8 * k); else c47 |= c << (8 * (k - 4));
                }
#pragma unroll
                for (int co = 0; co < COUT; ++co) g[co] = dy[(size_t)jl * COUT + co];      // lane t holds pair t's dy row
            }
            for (int t = 0; t < cnt; ++t) {
                const int a03 = __builtin_amdgcn_readlane(c03, t), a47 = __builtin_amdgcn_readlane(c47, t);
                float wl = 0.f;                                   // this cell's weight in pair t (corners in order)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int ck = ((k < 4 ? a03 >> (8 * k) : a47 >> (8 * (k - 4))) & 63);
                    const float wk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w8[k]), t));
                    wl += ck == lane ? wk : 0.f;
                }
#pragma unroll
                for (int co = 0; co < COUT; ++co)
                    acc[co] += wl * __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g[co]), t));
            }
        }
        float* out = dG + (size_t)row * (65 * COUT);
#pragma unroll
        for (int co = 0; co < COUT; ++co) out[lane * COUT + co] = acc[co];
        if (lane < COUT) out[64 * COUT + lane] = dy[(size_t)row * COUT + lane];
    }
}

// cout = 64 on the matrix pipe: for one particle the transposed gather is a small dense product,
//     dG[cell][co] = sum_t W[t][cell] * dy[j_t][co],      W[t][cell] = sum of pair t's corner weights that fall on `cell`
// (8 non-zeros per row of W).  A wave scatters its chunk of up to 64 pairs into W in LDS (lane t owns row t: its own 8
// read-add-writes, nothing shared) and runs 64 cells x 64 channels x pairs on v_mfma_f32_32x32x2_f32: per K-step (two pairs)
// two ds_read_b32 of W, two global loads of dy rows and four MFMAs, instead of ~150 VALU / LDS instructions per pair.
// Sums: every output element adds its pairs in list order (the pairs that do not touch the cell contribute exact zeros);
// duplicate corner cells of a pair (interpolation clamped at the filter's border) are added up before the product.
#define GT_PITCH 65
__global__ void __launch_bounds__(64) k_cconv_gather_t64(const float* __restrict__ dy, const int64_t* __restrict__ row_splits,
                                                         const int32_t* __restrict__ nbr, const float* __restrict__ tpw,
                                                         const uint8_t* __restrict__ tpc, int n, float* __restrict__ dG)
{
    __shared__ float Wt[64 * GT_PITCH];
    const int lane = threadIdx.x, h = lane >> 5, c = lane & 31;
    for (int row = blockIdx.x; row < n; row += gridDim.x) {
        const int64_t s = row_splits[row], e = row_splits[row + 1];
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        for (int64_t base = s; base < e; base += 64) {
            const int cnt = __builtin_amdgcn_readfirstlane((int)((e - base) < 64 ? (e - base) : 64));
            const int nk = (cnt + 1) >> 1;                       // K-steps of two pairs
            // W rows 0 .. 2 nk - 1 <- 0, then lane t adds its pair's 8 corner weights into row t
            for (int t = 0; t < 2 * nk; ++t) Wt[t * GT_PITCH + lane] = 0.f;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            int jl = row;                                         // padded pair: any valid row (its W row is zero, its B operand too)
            if (lane < cnt) {
                const int64_t p = base + lane;
                jl = nbr[p];
#pragma unroll
                for (int k = 0; k < 8; ++k) Wt[lane * GT_PITCH + tpc[p * 8 + k]] += tpw[p * 8 + k];
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            // B operands (dy rows of the two pairs of a K-step, one per half-wave), 4 K-steps ahead
            float b0[4], b1[4];
            auto loadB = [&](int kk, float& x0, float& x1) {
                const int t = 2 * kk + h;
                const int jlo = __builtin_amdgcn_readlane(jl, (2 * kk) & 63), jhi = __builtin_amdgcn_readlane(jl, (2 * kk + 1) & 63);
                const float* src = dy + (size_t)(h ? jhi : jlo) * 64 + c;
                const bool live = kk < nk && t < cnt;
                x0 = live ? src[0] : 0.f;
                x1 = live ? src[32] : 0.f;
            };
#pragma unroll
            for (int u = 0; u < 4; ++u) loadB(u, b0[u], b1[u]);
            for (int k0 = 0; k0 < nk; k0 += 4) {
                float n0[4], n1[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) loadB(k0 + 4 + u, n0[u], n1[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int kk = k0 + u;
                    if (kk < nk) {
                        const float a0 = Wt[(2 * kk + h) * GT_PITCH + c], a1 = Wt[(2 * kk + h) * GT_PITCH + 32 + c];
                        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[u], acc[0][0], 0, 0, 0);
                        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1[u], acc[0][1], 0, 0, 0);
                        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0[u], acc[1][0], 0, 0, 0);
                        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[u], acc[1][1], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) { b0[u] = n0[u]; b1[u] = n1[u]; }
            }
            __builtin_amdgcn_wave_barrier();                      // the next chunk rewrites W
        }
        float* out = dG + (size_t)row * (65 * 64);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    out[(32 * a + (r & 3) + 8 * (r >> 2) + 4 * h) * 64 + 32 * b + c] = acc[a][b][r];
        out[64 * 64 + lane] = dy[(size_t)row * 64 + lane];
    }
}

extern "C" int nf_cconv_gather_bwd(const float* dy, int cout, const int64_t* row_splits, const int32_t* nbr,
                                   const float* pair_w_t, const uint8_t* pair_cell_t, int n, float* dG, nf_stream_t stream)
{
    NF_CHECK_ARG(dy && row_splits && dG, "null pointer");
    NF_CHECK_ARG(cout >= 1 && cout <= 64, "cout must be in [1,64]");
    if (n <= 0) return NF_OK;
    if (cout <= 4) {
        const int blocks_s = n < 16384 ? n : 16384;
        hipStream_t st_ = (hipStream_t)stream;
        switch (cout) {
            case 1: hipLaunchKernelGGL(k_cconv_gather_t_small<1>, dim3(blocks_s), dim3(64), 0, st_, dy, row_splits, nbr, pair_w_t, pair_cell_t, n, dG); break;
            case 2: hipLaunchKernelGGL(k_cconv_gather_t_small<2>, dim3(blocks_s), dim3(64), 0, st_, dy, row_splits, nbr, pair_w_t, pair_cell_t, n, dG); break;
            case 3: hipLaunchKernelGGL(k_cconv_gather_t_small<3>, dim3(blocks_s), dim3(64), 0, st_, dy, row_splits, nbr, pair_w_t, pair_cell_t, n, dG); break;
            default: hipLaunchKernelGGL(k_cconv_gather_t_small<4>, dim3(blocks_s), dim3(64), 0, st_, dy, row_splits, nbr, pair_w_t, pair_cell_t, n, dG); break;
        }
        NF_CHECK_LAUNCH();
        return NF_OK;
    }
    if (cout == 64) {
        int blocks64 = n < 16384 ? n : 16384;
        hipLaunchKernelGGL(k_cconv_gather_t64, dim3(blocks64), dim3(64), 0, (hipStream_t)stream, dy, row_splits, nbr, pair_w_t,
                           pair_cell_t, n, dG);
        NF_CHECK_LAUNCH();
        return NF_OK;
    }
    // one wave per workgroup (16 KB of LDS at cout = 64): 55 us per launch at 4 913 particles x 39 pairs, 59 with four.  The
    // kernel is bound by its instruction count (~150 per pair and wave), not by LDS or memory: see DESIGN section 6
    const int wpb = 1;
    int blocks = (n + wpb - 1) / wpb;
    if (blocks > 16384) blocks = 16384;
    size_t lds = (size_t)wpb * 64 * cout * sizeof(float);
    hipLaunchKernelGGL(k_cconv_gather_t, dim3(blocks), dim3(64 * wpb), lds, (hipStream_t)stream, dy, cout, row_splits, nbr,
                       pair_w_t, pair_cell_t, n, dG);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// filter gradient of the direct small-Cin conv: dK[cell][ci][co] = sum_i sum_pairs(i) sum_corners pw * feat[j][ci] * dy[i][col_off + co]
// Factored as  dK = A^T * dy  with the "patch" matrix A[i][cell*CIN + ci] = sum_pairs sum_corners pw * feat[j][ci]
// (what Open3D builds explicitly before its GEMM):
//   1. k_cconv_small_patch : one wave per output point accumulates its 64*CIN patch in LDS (2 pairs x 8 corners x CIN
//      updates per step, ds_add_f32 inside the wave only) and writes the row of A;
//   2. k_cconv_small_wgemm : dK partials over row slices (thread = patch column, 32 accumulators), deterministic
//      reduction of the slices.
// (The first version accumulated pw * feat * dy for every (pair, corner, ci, co) straight into a block-wide LDS copy of
// dK with atomics and flushed with global atomics: 245 M LDS atomics, 1.1 ms for 4 913 points — the largest kernel of
// the end-to-end training step — and an order-dependent result.  This one: ~30 us, deterministic.)
#define SWG_SLICES 128
template <int CIN>
__global__ void __launch_bounds__(256) k_cconv_small_patch(const float* __restrict__ feats, const int64_t* __restrict__ row_splits,
                                                           const int32_t* __restrict__ nbr, const float* __restrict__ pw,
                                                           const uint8_t* __restrict__ pc, int n_out, float* __restrict__ A)
{
    __shared__ float patch[4][64 * 4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int sub = lane >> 5, k = (lane >> 2) & 7, ci = lane & 3;      // lane -> (pair of the step, corner, channel)
    float* pt = patch[wv];
    for (int row = blockIdx.x * 4 + wv; row < n_out; row += gridDim.x * 4) {
#pragma unroll
        for (int t = lane; t < 64 * CIN; t += 64) pt[t] = 0.f;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        const int64_t e = row_splits[row + 1];
        for (int64_t p = row_splits[row] + sub; p < e; p += 2) {
            if (ci < CIN) atomicAdd(pt + (int)pc[p * 8 + k] * CIN + ci, pw[p * 8 + k] * feats[(size_t)nbr[p] * CIN + ci]);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = lane; t < 64 * CIN; t += 64) A[(size_t)row * (64 * CIN) + t] = pt[t];
        __builtin_amdgcn_wave_barrier();
    }
}

template <int CIN>
__global__ void __launch_bounds__(64 * CIN) k_cconv_small_wgemm(const float* __restrict__ A, const float* __restrict__ dy, int ld_dy,
                                                                int col_off, int n_out, int rows_per_slice,
                                                                float* __restrict__ partial)
{
    __shared__ float gy[32];
    const int m = threadIdx.x;                       // patch column = cell * CIN + ci
    const int r0 = blockIdx.x * rows_per_slice, r1 = min(n_out, r0 + rows_per_slice);
    float acc[32];
#pragma unroll
    for (int co = 0; co < 32; ++co) acc[co] = 0.f;
    for (int row = r0; row < r1; ++row) {
        __syncthreads();
        if (m < 32) gy[m] = dy[(size_t)row * ld_dy + col_off + m];
        __syncthreads();
        const float a = A[(size_t)row * (64 * CIN) + m];
#pragma unroll
        for (int co = 0; co < 32; ++co) acc[co] += a * gy[co];
    }
    float* out = partial + ((size_t)blockIdx.x * (64 * CIN) + m) * 32;
#pragma unroll
    for (int co = 0; co < 32; ++co) out[co] = acc[co];
}

// 64 outputs x 4 slice groups per workgroup: a group adds every fourth slice with four loads in flight, the groups are
// folded through LDS in a fixed order (a thread per output walking all 128 slices one load after the other: 32 us for 4 MB)
__global__ void __launch_bounds__(256) k_cconv_small_wreduce(const float* __restrict__ partial, int total, int nslices,
                                                             float* __restrict__ dK)
{
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < total) {
        int z = g;
        for (; z + 12 < nslices; z += 16) {
            s0 += partial[(size_t)z * total + i];
            s1 += partial[(size_t)(z + 4) * total + i];
            s2 += partial[(size_t)(z + 8) * total + i];
            s3 += partial[(size_t)(z + 12) * total + i];
        }
        for (; z < nslices; z += 4) s0 += partial[(size_t)z * total + i];
    }
    part[g][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && i < total) dK[i] += (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);   // accumulate: the caller zero-initialises
}

extern "C" size_t nf_cconv_small_bwd_filter_workspace_floats(int cin, int n_out)
{
    return (size_t)(n_out > 0 ? n_out : 0) * 64 * cin + (size_t)SWG_SLICES * 64 * cin * 32;
}

extern "C" int nf_cconv_small_bwd_filter(const float* feats, int cin, const int64_t* row_splits, const int32_t* nbr,
                                         const float* pair_w, const uint8_t* pair_cell, const float* dy, int ld_dy,
                                         int col_off, int n_out, float* workspace, float* dkernel, nf_stream_t stream)
{
    NF_CHECK_ARG(feats && row_splits && dy && dkernel && workspace, "null pointer");
    NF_CHECK_ARG(cin == 3 || cin == 4, "cin must be 3 or 4");
    if (n_out <= 0) return NF_OK;
    int blocks = (n_out + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipStream_t st = (hipStream_t)stream;
    float* A = workspace;
    float* partial = workspace + (size_t)n_out * 64 * cin;
    const int rows_per = (n_out + SWG_SLICES - 1) / SWG_SLICES;
    const int total = 64 * cin * 32;
    if (cin == 3) {
        hipLaunchKernelGGL(k_cconv_small_patch<3>, dim3(blocks), dim3(256), 0, st, feats, row_splits, nbr, pair_w, pair_cell, n_out, A);
        hipLaunchKernelGGL(k_cconv_small_wgemm<3>, dim3(SWG_SLICES), dim3(192), 0, st, (const float*)A, dy, ld_dy, col_off, n_out,
                           rows_per, partial);
    } else {
        hipLaunchKernelGGL(k_cconv_small_patch<4>, dim3(blocks), dim3(256), 0, st, feats, row_splits, nbr, pair_w, pair_cell, n_out, A);
        hipLaunchKernelGGL(k_cconv_small_wgemm<4>, dim3(SWG_SLICES), dim3(256), 0, st, (const float*)A, dy, ld_dy, col_off, n_out,
                           rows_per, partial);
    }
    hipLaunchKernelGGL(k_cconv_small_wreduce, dim3((total + 63) / 64), dim3(256), 0, st, (const float*)partial, total, SWG_SLICES,
                       dkernel);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// feature gradient of the direct small-Cin conv (fluid<->fluid, transposed pair cache):
// dfeat[j][ci] = sum_{entries (j <- i)} sum_corners tpw * sum_co K[tpc][ci][co] * dy[i][col_off + co]
template <int CIN>
__global__ void __launch_bounds__(256) k_cconv_small_bwd_feat(const float* __restrict__ kernel,
                                                              const int64_t* __restrict__ row_splits,
                                                              const int32_t* __restrict__ nbr, const float* __restrict__ tpw,
                                                              const uint8_t* __restrict__ tpc, const float* __restrict__ dy,
                                                              int ld_dy, int col_off, int n, float* __restrict__ dfeat)
{
    __shared__ float Ks[64 * CIN * 32];
    for (int t = threadIdx.x; t < 64 * CIN * 32; t += 256) Ks[t] = kernel[t];
    __syncthreads();
    const int lane = threadIdx.x & 63, co = lane & 31, half = lane >> 5;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < n; row += gridDim.x * 4) {
        float acc[CIN];
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) acc[ci] = 0.f;
        for (int64_t p = row_splits[row] + half; p < row_splits[row + 1]; p += 2) {
            float g = dy[(size_t)nbr[p] * ld_dy + col_off + co];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float wg = tpw[p * 8 + k] * g;
                const float* kc = Ks + (int)tpc[p * 8 + k] * CIN * 32 + co;
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) acc[ci] += wg * kc[ci * 32];
            }
        }
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
            float v = acc[ci];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) dfeat[(size_t)row * CIN + ci] = v;
        }
    }
}

extern "C" int nf_cconv_small_bwd_feat(const float* kernel, int cin, const int64_t* row_splits, const int32_t* nbr,
                                       const float* pair_w_t, const uint8_t* pair_cell_t, const float* dy, int ld_dy,
                                       int col_off, int n, float* dfeat, nf_stream_t stream)
{
    NF_CHECK_ARG(kernel && row_splits && dy && dfeat, "null pointer");
    NF_CHECK_ARG(cin == 3 || cin == 4, "cin must be 3 or 4");
    if (n <= 0) return NF_OK;
    int blocks = (n + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    hipStream_t st = (hipStream_t)stream;
    if (cin == 3)
        hipLaunchKernelGGL(k_cconv_small_bwd_feat<3>, dim3(blocks), dim3(256), 0, st, kernel, row_splits, nbr, pair_w_t,
                           pair_cell_t, dy, ld_dy, col_off, n, dfeat);
    else
        hipLaunchKernelGGL(k_cconv_small_bwd_feat<4>, dim3(blocks), dim3(256), 0, st, kernel, row_splits, nbr, pair_w_t,
                           pair_cell_t, dy, ld_dy, col_off, n, dfeat);
    NF_CHECK_LAUNCH();
    return NF_OK;
}
