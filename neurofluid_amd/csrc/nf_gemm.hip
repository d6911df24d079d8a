// nf_gemm.hip — plain strided fp32 GEMM on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32), for the handful of dense
// products of the TRAINING path that are plain GEMMs and went through a vendor BLAS before (round 2's torch.mm calls):
//   renderer backward (models/nerf.py:83-124 under autograd):  dX = dpre_1 W_1 + dpre_5 W_5[:, :cx],  dXdir = dpre_dir W_dir[:, 256:]
//   ContinuousConv backward (models/transmodel.py:125):          dB = relu(x)^T dG,   dx = dG B^T,   dense0 weight / feature terms
//
//   C[m][n] (+)= sum_k opA(A)[m][k] * B[k][n]          A addressed as A[m * sa_m + k * sa_k], B as B[k * sb_k + n * sb_n]
// One of (sa_m, sa_k) and one of (sb_k, sb_n) must be 1 (the contiguous axis decides the LDS layout of the operand so that
// both the staged stores and the MFMA fragment reads are conflict-free); opA = ReLU on load (optional).  128x128 output
// tile per 4-wave workgroup, 32-deep slabs through LDS, next slab's global loads in flight behind the MFMAs; split-K over
// gridDim.z with a deterministic slice reduction.  Exact fp32 (fmaf chain per MFMA); only the summation order differs from a
// reference GEMM.
#include "nf_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GG_M 128
#define GG_N 128
#define GG_K 32
#define GG_PM 33          // pitch of an operand stored major-by-(m or n): odd -> fragment reads of 32 rows at one k are conflict-free
#define GG_PK 132         // pitch of an operand stored k-major

struct GemmArgs {
    const float* A; const float* B; float* C; float* ws;
    int M, N, K;
    long long sa_m, sa_k, sb_k, sb_n, ldc;
    int accumulate, relu_a, k_per_split;
};

// A tile (128 x 32) -> registers.  KC = true: the operand is contiguous along k (thread: row t >> 3, quad t & 7);
// false: contiguous along m (thread: k = t >> 5, quad of rows t & 31).
template <int MB, bool KC, bool VEC>
__device__ __forceinline__ void gg_load_a(const GemmArgs& g, int m0, int k0, int k_end, int tid, float4 (&r)[4])
{
#pragma unroll
    for (int u = 0; u < MB; ++u) {
        const int t = tid + 256 * u;
        float e[4] = {0.f, 0.f, 0.f, 0.f};
        if (KC) {
            const int m = m0 + (t >> 3), k = k0 + 4 * (t & 7);
            if (m < g.M) {
                const float* src = g.A + (long long)m * g.sa_m + k;
                if (VEC && k + 3 < k_end) { const float4 v = *(const float4*)src; e[0] = v.x; e[1] = v.y; e[2] = v.z; e[3] = v.w; }
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (k + j < k_end) e[j] = src[j];
                }
            }
        } else {
            const int k = k0 + t / (8 * MB), m = m0 + 4 * (t % (8 * MB));
            if (k < k_end) {
                const float* src = g.A + (long long)k * g.sa_k + m;
                if (VEC && m + 3 < g.M) { const float4 v = *(const float4*)src; e[0] = v.x; e[1] = v.y; e[2] = v.z; e[3] = v.w; }
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (m + j < g.M) e[j] = src[j];
                }
            }
        }
        if (g.relu_a) {
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = fmaxf(e[j], 0.f);
        }
        r[u] = make_float4(e[0], e[1], e[2], e[3]);
    }
}

// B tile (32 x 128).  NC = true: contiguous along n (thread: k = t >> 5, quad of columns t & 31); false: contiguous along k
// (thread: column t >> 3, quad of k t & 7).
template <bool NC, bool VEC>
__device__ __forceinline__ void gg_load_b(const GemmArgs& g, int n0, int k0, int k_end, int tid, float4 (&r)[4])
{
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int t = tid + 256 * u;
        float e[4] = {0.f, 0.f, 0.f, 0.f};
        if (NC) {
            const int k = k0 + (t >> 5), n = n0 + 4 * (t & 31);
            if (k < k_end) {
                const float* src = g.B + (long long)k * g.sb_k + n;
                if (VEC && n + 3 < g.N) { const float4 v = *(const float4*)src; e[0] = v.x; e[1] = v.y; e[2] = v.z; e[3] = v.w; }
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (n + j < g.N) e[j] = src[j];
                }
            }
        } else {
            const int n = n0 + (t >> 3), k = k0 + 4 * (t & 7);
            if (n < g.N) {
                const float* src = g.B + (long long)n * g.sb_n + k;
                if (VEC && k + 3 < k_end) { const float4 v = *(const float4*)src; e[0] = v.x; e[1] = v.y; e[2] = v.z; e[3] = v.w; }
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (k + j < k_end) e[j] = src[j];
                }
            }
        }
        r[u] = make_float4(e[0], e[1], e[2], e[3]);
    }
}

// MB = 32-row blocks of the output tile (tile = 32 MB x 128): every wave owns 32 columns and all MB row blocks, so an output
// with few rows (the filter gradients of the continuous convolutions: M = Cin = 96 / 64) fits its tile exactly instead of
// padding a 128-row tile, and a tall product with few tiles (dX of the renderer: 6 000 x 198, K = 256) gets twice the
// workgroups from 64-row tiles.  The sums of an output element do not depend on MB (same K order, same split).
template <int MB, bool A_KC, bool B_NC, bool VA, bool VB>
__global__ void __launch_bounds__(256) k_gemm_f32(GemmArgs g)
{
    // A: m-major [32 MB][33] (A_KC) or k-major [32][132]; B: k-major [32][132] (B_NC) or n-major [128][33]
    __shared__ float As[GG_M * GG_PM > GG_K * GG_PK ? GG_M * GG_PM : GG_K * GG_PK];
    __shared__ float Bs[GG_M * GG_PM > GG_K * GG_PK ? GG_M * GG_PM : GG_K * GG_PK];
    const int m0 = blockIdx.y * (32 * MB), n0 = blockIdx.x * GG_N;
    const int kb = blockIdx.z * g.k_per_split, k_end = min(g.K, kb + g.k_per_split);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave * 32;
    f32x16 acc[MB];
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float4 ra[4], rb[4];
    gg_load_a<MB, A_KC, VA>(g, m0, kb, k_end, tid, ra);
    gg_load_b<B_NC, VB>(g, n0, kb, k_end, tid, rb);
    for (int k0 = kb; k0 < k_end; k0 += GG_K) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = tid + 256 * u;
            if (u < MB) {
                if (A_KC) { float* d = As + (t >> 3) * GG_PM + 4 * (t & 7); d[0] = ra[u].x; d[1] = ra[u].y; d[2] = ra[u].z; d[3] = ra[u].w; }
                else *(float4*)(As + (t / (8 * MB)) * GG_PK + 4 * (t % (8 * MB))) = ra[u];
            }
            if (B_NC) *(float4*)(Bs + (t >> 5) * GG_PK + 4 * (t & 31)) = rb[u];
            else { float* d = Bs + (t >> 3) * GG_PM + 4 * (t & 7); d[0] = rb[u].x; d[1] = rb[u].y; d[2] = rb[u].z; d[3] = rb[u].w; }
        }
        __syncthreads();
        if (k0 + GG_K < k_end) {      // next slab in flight while this one is multiplied
            gg_load_a<MB, A_KC, VA>(g, m0, k0 + GG_K, k_end, tid, ra);
            gg_load_b<B_NC, VB>(g, n0, k0 + GG_K, k_end, tid, rb);
        }
#pragma unroll
        for (int kk = 0; kk < GG_K; kk += 2) {
            const int kr = kk + (lane >> 5), c = lane & 31;
            const float b0 = B_NC ? Bs[kr * GG_PK + wn + c] : Bs[(wn + c) * GG_PM + kr];
#pragma unroll
            for (int a = 0; a < MB; ++a) {
                const float av = A_KC ? As[(32 * a + c) * GG_PM + kr] : As[kr * GG_PK + 32 * a + c];
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc[a], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // D layout: lane -> column j = lane & 31, rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const bool split = gridDim.z > 1;
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int n = n0 + wn + (lane & 31);
            if (m < g.M && n < g.N) {
                if (split) g.ws[((size_t)blockIdx.z * g.M + m) * g.N + n] = acc[a][r];
                else {
                    float* dst = g.C + (long long)m * g.ldc + n;
                    *dst = g.accumulate ? *dst + acc[a][r] : acc[a][r];
                }
            }
        }
}

// slices folded in a fixed order (deterministic), four independent loads per step
__global__ void __launch_bounds__(256) k_gemm_reduce(const float* __restrict__ ws, int M, int N, int splits, float* __restrict__ C,
                                                     long long ldc, int accumulate)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)M * N;
    if (i >= total) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int z = 0;
    for (; z + 3 < splits; z += 4) {
        s0 += ws[(size_t)z * total + i];
        s1 += ws[(size_t)(z + 1) * total + i];
        s2 += ws[(size_t)(z + 2) * total + i];
        s3 += ws[(size_t)(z + 3) * total + i];
    }
    for (; z < splits; ++z) s0 += ws[(size_t)z * total + i];
    const float v = (s0 + s1) + (s2 + s3);
    float* dst = C + (long long)(i / N) * ldc + (i % N);
    *dst = accumulate ? *dst + v : v;
}

extern "C" size_t nf_gemm_f32_workspace_floats(int M, int N, int splits)
{
    return splits > 1 ? (size_t)splits * (size_t)(M > 0 ? M : 0) * (size_t)(N > 0 ? N : 0) : 0;
}

extern "C" int nf_gemm_f32(int M, int N, int K, const float* A, int64_t sa_m, int64_t sa_k, int relu_a, const float* B,
                           int64_t sb_k, int64_t sb_n, float* C, int64_t ldc, int accumulate, int splits, float* workspace,
                           nf_stream_t stream)
{
    NF_CHECK_ARG(A && B && C, "null pointer");
    NF_CHECK_ARG(M >= 0 && N >= 0 && K >= 0 && ldc >= N, "bad sizes");
    NF_CHECK_ARG((sa_k == 1 || sa_m == 1) && (sb_n == 1 || sb_k == 1), "each operand needs a unit stride along one axis");
    NF_CHECK_ARG(splits >= 1 && (splits == 1 || workspace), "split-K needs a workspace (nf_gemm_f32_workspace_floats)");
    if (M == 0 || N == 0) return NF_OK;
    hipStream_t st = (hipStream_t)stream;
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.ws = workspace; g.M = M; g.N = N; g.K = K;
    g.sa_m = sa_m; g.sa_k = sa_k; g.sb_k = sb_k; g.sb_n = sb_n; g.ldc = ldc; g.accumulate = accumulate; g.relu_a = relu_a;
    int slabs = (K + GG_K - 1) / GG_K;
    if (splits > slabs) splits = slabs > 0 ? slabs : 1;
    g.k_per_split = ((slabs + splits - 1) / splits) * GG_K;
    if (g.k_per_split <= 0) g.k_per_split = GG_K;
    splits = K > 0 ? (K + g.k_per_split - 1) / g.k_per_split : 1;
    const bool a_kc = sa_k == 1, b_nc = sb_n == 1;
    // 16-B vector loads need an aligned base and a stride of whole quads along the non-contiguous axis
    const bool va = ((uintptr_t)A % 16 == 0) && ((a_kc ? sa_m : sa_k) % 4 == 0);
    const bool vb = ((uintptr_t)B % 16 == 0) && ((b_nc ? sb_k : sb_n) % 4 == 0);
    // rows per tile: an output of at most 128 rows gets a tile that fits it; a taller one 128-row tiles, or 64-row tiles when
    // 128-row tiles would leave most of the chip without a workgroup
    int mb = 4;
    if (M <= 128) mb = (M + 31) / 32;
    else if ((long long)((N + GG_N - 1) / GG_N) * ((M + 127) / 128) * splits < 192) mb = 2;
    dim3 grid((N + GG_N - 1) / GG_N, (M + 32 * mb - 1) / (32 * mb), splits);
#define GG_LAUNCH(AK, BN, VA_, VB_)                                                                                   \
    do {                                                                                                              \
        if (mb == 4) hipLaunchKernelGGL((k_gemm_f32<4, AK, BN, VA_, VB_>), grid, dim3(256), 0, st, g);                 \
        else if (mb == 3) hipLaunchKernelGGL((k_gemm_f32<3, AK, BN, VA_, VB_>), grid, dim3(256), 0, st, g);            \
        else if (mb == 2) hipLaunchKernelGGL((k_gemm_f32<2, AK, BN, VA_, VB_>), grid, dim3(256), 0, st, g);            \
        else hipLaunchKernelGGL((k_gemm_f32<1, AK, BN, VA_, VB_>), grid, dim3(256), 0, st, g);                         \
    } while (0)
#define GG_PICK_V(AK, BN)                                       \
    do {                                                        \
        if (va && vb) GG_LAUNCH(AK, BN, true, true);            \
        else if (va) GG_LAUNCH(AK, BN, true, false);            \
        else if (vb) GG_LAUNCH(AK, BN, false, true);            \
        else GG_LAUNCH(AK, BN, false, false);                   \
    } while (0)
    if (a_kc && b_nc) GG_PICK_V(true, true);
    else if (a_kc) GG_PICK_V(true, false);
    else if (b_nc) GG_PICK_V(false, true);
    else GG_PICK_V(false, false);
#undef GG_PICK_V
#undef GG_LAUNCH
    NF_CHECK_LAUNCH();
    if (splits > 1) {
        const size_t total = (size_t)M * N;
        hipLaunchKernelGGL(k_gemm_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const float*)workspace, M, N,
                           splits, C, (long long)ldc, accumulate);
        NF_CHECK_LAUNCH();
    }
    return NF_OK;
}
