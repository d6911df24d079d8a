// nf_mlp.hip — the NeRF MLP (models/nerf.py:83-124) as ONE persistent fp32-MFMA kernel.
//
// Design:
//  * transposed problem  D[out_feature][sample] = W[out][in] * H[in][sample]  on
//    v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD = the chip's 157 TF f32 matrix peak);
//    a wave owns 32 samples for the WHOLE network.  With that orientation the D fragment of one
//    layer (lane l: sample l&31, features (r&3)+8(r>>2)+4(l>>5)) IS a valid B fragment of the next
//    layer — register r of block b pairs feature f with f+4 across the two half-waves, and the
//    K order of a dot product is free — so hidden activations never leave the register file:
//    128 VGPRs of activations + 128 AGPRs of accumulators per lane, one wave per SIMD.
//  * the A operand (weights) is pre-packed (nf_nerf_pack) so that each lane fetches the operands of
//    4 consecutive MFMAs with one 16-byte load; the 2.7 MB of weights of a net stay L2-resident
//    (4 MiB L2 per XCD) and stream L2 -> VGPR four K-steps ahead of the MFMAs, no LDS, no barriers.
//    Every VMEM issue is placed in the shadow of ONE MFMA (sched_group_barrier): a load costs ~55
//    cycles of the wave's issue slot, two back to back idle the matrix pipe (measured -17 %).
//  * two 128-register accumulator sets alternate between layers; the ReLU of the previous set is
//    applied on the fly, one register per K-step; the bias enters as one extra K-step (A = bias,
//    B = 1, C = 0): there is no per-layer VALU pass.
//  * skip connection (layer 5) and the view branch are split-K accumulations over the feature
//    matrix X (re-read from HBM/L2, 1 KB per row), never materialised concatenations.
//  * sigma (256->1) and rgb (128->3) heads run on the VALU from the fragments, then one 16-B store
//    per sample scatters (r,g,b,sigma) to the dense per-sample array.
#include "nf_common.h"
#include <math.h>
#include <type_traits>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

#include "nf_mlp_layout.h"

extern "C" size_t nf_nerf_packed_floats(int cx, int cd) { return (size_t)mlp_layout(cx, cd).total; }


__global__ void k_mlp_pack(NfMlpLayout L, NfNerfPtrs P, float* __restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L.total) return;
    float v = 0.f;
    if (i >= L.off_bstep[0]) {   // bias K-steps
        if (i >= L.off_bstep_dir) {
            int k = i - L.off_bstep_dir, e = k & 3, lane = (k >> 2) & 63;
            out[i] = lane < 32 ? P.b[9][32 * e + lane] : 0.f;
        } else {
            int l = (i - L.off_bstep[0]) / 512, k = (i - L.off_bstep[0]) % 512;
            int e = k & 3, lane = (k >> 2) & 63, g = k >> 8;
            out[i] = lane < 32 ? P.b[l][32 * (4 * g + e) + lane] : 0.f;
        }
        return;
    }
    // biases
    if (i >= L.off_b[0]) {
        if (i >= L.off_brgb) { int k = i - L.off_brgb; v = k < 3 ? P.b[11][k] : 0.f; }
        else if (i >= L.off_bsig) { int k = i - L.off_bsig; v = k < 1 ? P.b[10][0] : 0.f; }
        else if (i >= L.off_bdir) v = P.b[9][i - L.off_bdir];
        else { int l = (i - L.off_b[0]) / 256; v = P.b[l][(i - L.off_b[0]) % 256]; }
        out[i] = v;
        return;
    }
    if (i >= L.off_wrgb) {  // [c][(b*16+r)][h]
        int k = i - L.off_wrgb, c = k / 128, rem = k % 128, br = rem >> 1, h = rem & 1;
        v = P.w[11][c * 128 + frag_feature(br >> 4, br & 15, h)];
        out[i] = v;
        return;
    }
    if (i >= L.off_wsig) {
        int k = i - L.off_wsig, br = k >> 1, h = k & 1;
        out[i] = P.w[10][frag_feature(br >> 4, br & 15, h)];
        return;
    }
    if (i >= L.off_dir_x) {  // [s][lane][e], 4 output blocks
        int k = i - L.off_dir_x, e = k & 3, lane = (k >> 2) & 63, s = k >> 8;
        int q = s >> 2, r = s & 3, h = lane >> 5;
        int f = 8 * q + 4 * h + r;  // index into the dir-like features
        int o = 32 * e + (lane & 31);
        out[i] = f < L.cd ? P.w[9][(size_t)o * (256 + L.cd) + 256 + f] : 0.f;
        return;
    }
    if (i >= L.off_dir_h) {
        int k = i - L.off_dir_h, e = k & 3, lane = (k >> 2) & 63, s = k >> 8;
        int f = frag_feature(s >> 4, s & 15, lane >> 5);
        int o = 32 * e + (lane & 31);
        out[i] = P.w[9][(size_t)o * (256 + L.cd) + f];
        return;
    }
    // layers 0..8
    for (int l = 8; l >= 0; --l) {
        int in_dim = (l == 0) ? L.cx : (l == 4 ? L.cx + 256 : 256);
        if (L.off_h[l] >= 0 && i >= L.off_h[l]) {  // [s][g][lane][e]
            int k = i - L.off_h[l], e = k & 3, lane = (k >> 2) & 63, g = (k >> 8) & 1, s = k >> 9;
            int f = frag_feature(s >> 4, s & 15, lane >> 5);
            int o = 32 * (4 * g + e) + (lane & 31);
            int col = (l == 4) ? L.cx + f : f;  // layer 5 input = cat[input_xyz, h]
            out[i] = P.w[l][(size_t)o * in_dim + col];
            return;
        }
        if (L.off_x[l] >= 0 && i >= L.off_x[l]) {
            int k = i - L.off_x[l], e = k & 3, lane = (k >> 2) & 63, g = (k >> 8) & 1, s = k >> 9;
            int q = s >> 2, r = s & 3, h = lane >> 5;
            int f = 8 * q + 4 * h + r;
            int o = 32 * (4 * g + e) + (lane & 31);
            out[i] = f < L.cx ? P.w[l][(size_t)o * in_dim + f] : 0.f;
            return;
        }
    }
}

extern "C" int nf_nerf_pack(const nf_nerf_params_t* params, int cx, int cd, float* packed, nf_stream_t stream)
{
    NF_CHECK_ARG(params && packed, "null pointer");
    NF_CHECK_ARG(cx >= 1 && cx <= 256 && cd >= 1 && cd <= 256, "bad channel counts");
    NfNerfPtrs P;
    for (int i = 0; i < 12; ++i) {
        NF_CHECK_ARG(params->w[i] && params->b[i], "null weight/bias pointer");
        P.w[i] = params->w[i]; P.b[i] = params->b[i];
    }
    NfMlpLayout L = mlp_layout(cx, cd);
    hipLaunchKernelGGL(k_mlp_pack, dim3((L.total + 255) / 256), dim3(256), 0, (hipStream_t)stream, L, P, packed);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// ------------------------------------------------------------------------------------------------
// forward kernel
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void step8(const f32x4 w0, const f32x4 w1, const float bv, f32x16 (&acc)[8])
{
    acc[0] = MFMA32(w0[0], bv, acc[0]);
    acc[1] = MFMA32(w0[1], bv, acc[1]);
    acc[2] = MFMA32(w0[2], bv, acc[2]);
    acc[3] = MFMA32(w0[3], bv, acc[3]);
    acc[4] = MFMA32(w1[0], bv, acc[4]);
    acc[5] = MFMA32(w1[1], bv, acc[5]);
    acc[6] = MFMA32(w1[2], bv, acc[6]);
    acc[7] = MFMA32(w1[3], bv, acc[7]);
}

__device__ __forceinline__ void step4(const f32x4 w0, const float bv, f32x16 (&acc)[4])
{
    acc[0] = MFMA32(w0[0], bv, acc[0]);
    acc[1] = MFMA32(w0[1], bv, acc[1]);
    acc[2] = MFMA32(w0[2], bv, acc[2]);
    acc[3] = MFMA32(w0[3], bv, acc[3]);
}

// K-steps whose B operand comes from the feature matrix: group q = 8 features = 4 steps.
// STASH: the feature groups are copied into this wave's private LDS slice as they are consumed (layer 1), so that
// the skip layer reads them back from LDS instead of fetching X from HBM a second time (the tile has left the L2
// by then: without the stash the kernel moved 1.8x its algorithmic bytes).
// X is streamed exactly once: a non-temporal load keeps it from evicting the 2.7 MB weight set that all waves of
// the XCD re-read from the L2 every tile.
template <bool NT>
__device__ __forceinline__ f32x4 load_x(const f32x4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }

template <bool STASH>
__device__ __forceinline__ void kloop_x8(const f32x4* __restrict__ wp /* + lane */, const f32x4* __restrict__ xp /* + lane */,
                                         int nq, f32x16 (&acc)[8], f32x4* __restrict__ stash /* LDS + lane */)
{
    f32x4 xv = load_x<STASH>(xp);
    f32x4 w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = wp[i * 64];
    for (int q = 0; q < nq; ++q) {
        f32x4 xn = xv, wn[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) wn[i] = w[i];
        if (q + 1 < nq) {
            xn = load_x<STASH>(xp + (q + 1) * 64);
#pragma unroll
            for (int i = 0; i < 8; ++i) wn[i] = wp[(q + 1) * 512 + i * 64];
        }
        if (STASH) stash[q * 64] = xv;
        step8(w[0], w[1], xv[0], acc);
        step8(w[2], w[3], xv[1], acc);
        step8(w[4], w[5], xv[2], acc);
        step8(w[6], w[7], xv[3], acc);
        if (q + 1 < nq) {   // one VMEM issue per MFMA shadow (see kloop_src)
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 23, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        xv = xn;
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = wn[i];
    }
}

__device__ __forceinline__ void kloop_x4(const f32x4* __restrict__ wp, const f32x4* __restrict__ xp, int nq,
                                         f32x16 (&acc)[4])
{
    f32x4 xv = load_x<true>(xp);
    f32x4 w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = wp[i * 64];
    for (int q = 0; q < nq; ++q) {
        f32x4 xn = xv, wn[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) wn[i] = w[i];
        if (q + 1 < nq) {
            xn = load_x<true>(xp + (q + 1) * 64);
#pragma unroll
            for (int i = 0; i < 4; ++i) wn[i] = wp[(q + 1) * 256 + i * 64];
        }
        step4(w[0], xv[0], acc);
        step4(w[1], xv[1], acc);
        step4(w[2], xv[2], acc);
        step4(w[3], xv[3], acc);
        if (q + 1 < nq) {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 11, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        xv = xn;
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = wn[i];
    }
}


// row-major save of a fragment (training): feature f of sample `row` -> dst[row*stride + f]
template <int NB>
__device__ __forceinline__ void save_frag(const f32x16 (&v)[NB], float* __restrict__ dst_row /* row base + section */, int h)
{
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            f32x4 o = {v[b][4 * rq], v[b][4 * rq + 1], v[b][4 * rq + 2], v[b][4 * rq + 3]};
            *(f32x4*)(dst_row + 32 * b + 8 * rq + 4 * h) = o;
        }
}

// ------------------------------------------------------------------------------------------------
// forward kernel: double-buffered accumulators.  Layer l accumulates into one 128-register set while the
// ReLU of the previous layer's set is applied ON THE FLY, one register per K-step, hidden behind the MFMAs;
// the bias enters as one extra K-step (A = bias, B = 1, C = 0), so there is no per-layer VALU pass at all.
// Weights are fetched 4 K-steps ahead (5-slot register ring).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bias_step8(const f32x4* __restrict__ p /* + lane */, f32x16 (&acc)[8])
{
    const f32x4 w0 = p[0], w1 = p[64];
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    acc[0] = MFMA32(w0[0], 1.f, z); acc[1] = MFMA32(w0[1], 1.f, z); acc[2] = MFMA32(w0[2], 1.f, z); acc[3] = MFMA32(w0[3], 1.f, z);
    acc[4] = MFMA32(w1[0], 1.f, z); acc[5] = MFMA32(w1[1], 1.f, z); acc[6] = MFMA32(w1[2], 1.f, z); acc[7] = MFMA32(w1[3], 1.f, z);
}

__device__ __forceinline__ void bias_step4(const f32x4* __restrict__ p, f32x16 (&acc)[4])
{
    const f32x4 w0 = p[0];
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    acc[0] = MFMA32(w0[0], 1.f, z); acc[1] = MFMA32(w0[1], 1.f, z); acc[2] = MFMA32(w0[2], 1.f, z); acc[3] = MFMA32(w0[3], 1.f, z);
}

// dst += W * act(src), act = ReLU (RELU) or identity; NB = 8 or 4 output blocks.
// SAVE: the activated values of `src` are stored row-major at save_row (training).
template <bool RELU, int NB, bool SAVE>
__device__ __forceinline__ void kloop_src(const f32x4* __restrict__ p /* + lane */, const f32x16 (&src)[8],
                                          f32x16 (&dst)[NB], float* __restrict__ save_row, int h, bool row_ok)
{
    constexpr int G = NB / 4;          // float4 loads per step
    constexpr int D = 4;               // prefetch distance in steps
    f32x4 ring[D + 1][G];
#pragma unroll
    for (int s = 0; s < D; ++s)
#pragma unroll
        for (int g = 0; g < G; ++g) ring[s][g] = p[(s * G + g) * 64];
    f32x16 cur, nxt;
#pragma unroll
    for (int r = 0; r < 16; ++r) cur[r] = RELU ? fmaxf(src[0][r], 0.f) : src[0][r];
#pragma unroll
    for (int b = 0; b < 8; ++b) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int s = b * 16 + r;
            if (s + D < 128) {
#pragma unroll
                for (int g = 0; g < G; ++g) ring[(s + D) % (D + 1)][g] = p[((s + D) * G + g) * 64];
            }
            if (b + 1 < 8) nxt[r] = RELU ? fmaxf(src[b + 1][r], 0.f) : src[b + 1][r];   // next block's operand, hidden
            const float bv = cur[r];
            const f32x4 w0 = ring[s % (D + 1)][0];
            dst[0] = MFMA32(w0[0], bv, dst[0]); dst[1] = MFMA32(w0[1], bv, dst[1]);
            dst[2] = MFMA32(w0[2], bv, dst[2]); dst[3] = MFMA32(w0[3], bv, dst[3]);
            if (NB == 8) {
                const f32x4 w1 = ring[s % (D + 1)][G - 1];
                dst[NB - 4] = MFMA32(w1[0], bv, dst[NB - 4]); dst[NB - 3] = MFMA32(w1[1], bv, dst[NB - 3]);
                dst[NB - 2] = MFMA32(w1[2], bv, dst[NB - 2]); dst[NB - 1] = MFMA32(w1[3], bv, dst[NB - 1]);
            }
            // A VMEM issue costs ~55 cycles of the wave's issue slot: one per MFMA shadow (64 cycles), never two
            // back to back (measured: adjacent load pairs idle the matrix pipe ~45 cycles per K-step).
            if (s + D < 128) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NB == 8 ? 3 : 3, 0);
                if (NB == 8) {
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (SAVE && row_ok) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                f32x4 o = {cur[4 * rq], cur[4 * rq + 1], cur[4 * rq + 2], cur[4 * rq + 3]};
                *(f32x4*)(save_row + 32 * b + 8 * rq + 4 * h) = o;
            }
        }
        cur = nxt;
    }
}

template <bool SAVE>
__device__ __forceinline__ void mlp_layer(const NfMlpLayout& L, const f32x4* __restrict__ P4, int l, int lane, int h,
                                          const f32x4* __restrict__ xs /* LDS stash + lane */, const f32x16 (&src)[8],
                                          f32x16 (&dst)[8], float* __restrict__ arow, bool row_ok)
{
    bias_step8(P4 + (L.off_bstep[l] >> 2) + lane, dst);
    if (L.off_x[l] >= 0) kloop_x8<false>(P4 + (L.off_x[l] >> 2) + lane, xs, L.qx, dst, nullptr);
    // the ReLU'd src IS the saved activation h_l (slot l-1)
    kloop_src<true, 8, SAVE>(P4 + (L.off_h[l] >> 2) + lane, src, dst, SAVE ? arow + (l - 1) * 256 : nullptr, h, row_ok);
}

template <bool SAVE>
__global__ void __launch_bounds__(256) k_mlp_fwd(NfMlpLayout L, const float* __restrict__ packed,
                                                  const float* __restrict__ X, const int* __restrict__ n_rows, int max_rows,
                                                  const int* __restrict__ row_sample, float4* __restrict__ rgbsigma,
                                                  float* __restrict__ acts)
{
    extern __shared__ f32x4 xstash[];      // [4 waves][qx][64 lanes] x 16 B: each wave's own copy of its tile's position groups
    const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
    const int gwave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const int nrows = min(*n_rows, max_rows);
    const int ntiles = (nrows + 31) >> 5;
    const int Q = L.qx + L.qd;
    f32x4* xs = xstash + (threadIdx.x >> 6) * (L.qx * 64) + lane;

    for (int tile = gwave; tile < ntiles; tile += nwaves) {
        const float* __restrict__ pk = packed + opaque_zero();
        const f32x4* P4 = (const f32x4*)pk;
        const f32x4* xt = (const f32x4*)X + (size_t)tile * Q * 64 + lane;
        const int row = tile * 32 + j;
        const bool row_ok = row < nrows;
        float* arow = SAVE ? acts + (size_t)(row_ok ? row : 0) * NF_ACT_STRIDE : nullptr;
        f32x16 accA[8], accB[8];

        // layer 0 = xyz_encoding_1: bias step + feature-matrix K-steps
        bias_step8(P4 + (L.off_bstep[0] >> 2) + lane, accA);
        kloop_x8<true>(P4 + (L.off_x[0] >> 2) + lane, xt, L.qx, accA, xs);
#pragma unroll 1
        for (int l = 1; l < 9; l += 2) {
            mlp_layer<SAVE>(L, P4, l, lane, h, xs, accA, accB, arow, row_ok);
            if (l + 1 == 8) {  // sigma head reads h8 = relu(accB) before xyz_encoding_final consumes it
                // (computed again inside the next layer's loop; 256 VALU ops per tile)
            }
            mlp_layer<SAVE>(L, P4, l + 1, lane, h, xs, accB, accA, arow, row_ok);
        }
        // after the loop: accA = xyz_encoding_final output (no activation), accB = pre-activation of layer 8 (h8 = relu)
        float sigma;
        {
            const float* ws_ = pk + L.off_wsig;
            float part = 0.f;
#pragma unroll
            for (int b = 0; b < 8; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float w0 = ws_[(b * 16 + r) * 2], w1 = ws_[(b * 16 + r) * 2 + 1];
                    part += fmaxf(accB[b][r], 0.f) * (h ? w1 : w0);
                }
            sigma = part + __shfl_xor(part, 32, 64) + pk[L.off_bsig];
        }
        // view branch: hd = relu(W_dir [final | dir feats] + b); final is saved as activation slot 8
        f32x16 hd[4];
        bias_step4(P4 + (L.off_bstep_dir >> 2) + lane, hd);
        kloop_x4(P4 + (L.off_dir_x >> 2) + lane, xt + L.qx * 64, L.qd, hd);
        kloop_src<false, 4, SAVE>(P4 + (L.off_dir_h >> 2) + lane, accA, hd, SAVE ? arow + 8 * 256 : nullptr, h, row_ok);
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) hd[b][r] = fmaxf(hd[b][r], 0.f);
        if (SAVE && row_ok) save_frag<4>(hd, arow + 9 * 256, h);

        const float* wr = pk + L.off_wrgb;
        float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = hd[b][r];
                int k = (b * 16 + r) * 2;
                c0 += v * (h ? wr[k + 1] : wr[k]);
                c1 += v * (h ? wr[128 + k + 1] : wr[128 + k]);
                c2 += v * (h ? wr[256 + k + 1] : wr[256 + k]);
            }
        c0 += __shfl_xor(c0, 32, 64); c1 += __shfl_xor(c1, 32, 64); c2 += __shfl_xor(c2, 32, 64);
        c0 += pk[L.off_brgb]; c1 += pk[L.off_brgb + 1]; c2 += pk[L.off_brgb + 2];
        if (h == 0 && row_ok) {
            float4 o;
            o.x = 1.f / (1.f + expf(-c0)); o.y = 1.f / (1.f + expf(-c1)); o.z = 1.f / (1.f + expf(-c2)); o.w = sigma;
            rgbsigma[row_sample[row]] = o;
        }
    }
}

extern "C" int nf_nerf_mlp_fwd(const float* packed, int cx, int cd, const float* X, const int32_t* n_rows, int max_rows,
                               const int32_t* row_sample, float* rgbsigma, float* acts, nf_stream_t stream)
{
    NF_CHECK_ARG(packed && X && n_rows && row_sample && rgbsigma, "null pointer");
    NF_CHECK_ARG(cx >= 1 && cx <= 256 && cd >= 1 && cd <= 256, "bad channel counts");
    if (max_rows <= 0) return NF_OK;
    NfMlpLayout L = mlp_layout(cx, cd);
    int tiles = (max_rows + 31) / 32;
    int blocks = (tiles + 3) / 4;
    if (blocks > 256) blocks = 256;  // one 4-wave workgroup per CU, persistent over tiles
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)4 * L.qx * 64 * sizeof(f32x4);      // the X stash: 100 KB at qx = 25 (one workgroup per CU)
    NF_CHECK_ARG(lds <= 160 * 1024, "feature row too wide for the LDS stash");
    static bool attr_set[64] = {};
    if (nf_first_use_on_device(attr_set)) {
        hipFuncSetAttribute((const void*)k_mlp_fwd<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)k_mlp_fwd<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    if (acts)
        hipLaunchKernelGGL(k_mlp_fwd<true>, dim3(blocks), dim3(256), lds, st, L, packed, X, n_rows, max_rows, row_sample,
                           (float4*)rgbsigma, acts);
    else
        hipLaunchKernelGGL(k_mlp_fwd<false>, dim3(blocks), dim3(256), lds, st, L, packed, X, n_rows, max_rows, row_sample,
                           (float4*)rgbsigma, acts);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// ================================================================================================
// backward (data gradient) — same register-resident structure, transposed weights
// ================================================================================================
extern "C" size_t nf_nerf_packed_bwd_floats(void) { return (size_t)mlp_layout_t().total; }

__global__ void k_mlp_pack_bwd(NfMlpLayoutT T, int cx, int cd, NfNerfPtrs P, float* __restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T.total) return;
    if (i >= T.off_dx0) {           // the feature-gradient parts: K = the layer's output units, outputs = feature columns (zero beyond cx / cd)
        const int part = i >= T.off_dxd ? 2 : (i >= T.off_dx4 ? 1 : 0);
        int k = i - (part == 2 ? T.off_dxd : (part == 1 ? T.off_dx4 : T.off_dx0)), e = k & 3, lane = (k >> 2) & 63, g = (k >> 8) & 1, s = k >> 9;
        int o = frag_feature(s >> 4, s & 15, lane >> 5);          // the layer's output unit (K index): < 256, dir: < 128
        int in = 32 * (4 * g + e) + (lane & 31);                   // feature column
        float v = 0.f;
        if (part == 0) { if (in < cx) v = P.w[0][(size_t)o * cx + in]; }
        else if (part == 1) { if (in < cx) v = P.w[4][(size_t)o * (cx + 256) + in]; }
        else if (in < cd) v = P.w[9][(size_t)o * (256 + cd) + 256 + in];
        out[i] = v;
        return;
    }
    if (i >= T.off_dir) {
        int k = i - T.off_dir, e = k & 3, lane = (k >> 2) & 63, g = (k >> 8) & 1, s = k >> 9;
        int o = frag_feature(s >> 4, s & 15, lane >> 5);          // dir hidden unit (K index), < 128
        int in = 32 * (4 * g + e) + (lane & 31);                   // final feature (output of this GEMM)
        out[i] = P.w[9][(size_t)o * (256 + cd) + in];
        return;
    }
    for (int l = 8; l >= 1; --l)
        if (i >= T.off_h[l]) {
            int k = i - T.off_h[l], e = k & 3, lane = (k >> 2) & 63, g = (k >> 8) & 1, s = k >> 9;
            int o = frag_feature(s >> 4, s & 15, lane >> 5);
            int in = 32 * (4 * g + e) + (lane & 31);
            int in_dim = (l == 4) ? cx + 256 : 256;
            int col = (l == 4) ? cx + in : in;
            out[i] = P.w[l][(size_t)o * in_dim + col];
            return;
        }
}

extern "C" int nf_nerf_pack_bwd(const nf_nerf_params_t* params, int cx, int cd, float* packed_t, nf_stream_t stream)
{
    NF_CHECK_ARG(params && packed_t, "null pointer");
    NfNerfPtrs P;
    for (int i = 0; i < 12; ++i) { P.w[i] = params->w[i]; P.b[i] = params->b[i]; }
    NfMlpLayoutT T = mlp_layout_t();
    hipLaunchKernelGGL(k_mlp_pack_bwd, dim3((T.total + 255) / 256), dim3(256), 0, (hipStream_t)stream, T, cx, cd, P, packed_t);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

template <int NB>
__device__ __forceinline__ void zero_frag(f32x16 (&acc)[NB])
{
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
}

// K-steps over a 4-block (128-feature) fragment, 8 output blocks
__device__ __forceinline__ void kloop_act8_from4(const f32x4* __restrict__ p, const f32x16 (&act)[4], f32x16 (&acc)[8])
{
    f32x4 a0 = p[0], a1 = p[64];
    f32x4 b0 = p[128], b1 = p[192];
#pragma unroll
    for (int s = 0; s < 64; s += 2) {
        f32x4 c0 = a0, c1 = a1, d0 = b0, d1 = b1;
        if (s + 2 < 64) { c0 = p[(s + 2) * 128]; c1 = p[(s + 2) * 128 + 64]; }
        step8(a0, a1, act[s >> 4][s & 15], acc);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 3 < 64) { d0 = p[(s + 3) * 128]; d1 = p[(s + 3) * 128 + 64]; }
        step8(b0, b1, act[(s + 1) >> 4][(s + 1) & 15], acc);
        __builtin_amdgcn_sched_barrier(0);
        a0 = c0; a1 = c1; b0 = d0; b1 = d1;
    }
}

// K-steps over a 4-block (128-feature) source fragment, 8 output blocks; same load discipline as kloop_src
__device__ __forceinline__ void kloop_src4(const f32x4* __restrict__ p, const f32x16 (&src)[4], f32x16 (&dst)[8])
{
    constexpr int D = 4;
    f32x4 ring[D + 1][2];
#pragma unroll
    for (int s = 0; s < D; ++s) { ring[s][0] = p[(s * 2) * 64]; ring[s][1] = p[(s * 2 + 1) * 64]; }
#pragma unroll
    for (int s = 0; s < 64; ++s) {
        if (s + D < 64) { ring[(s + D) % (D + 1)][0] = p[((s + D) * 2) * 64]; ring[(s + D) % (D + 1)][1] = p[((s + D) * 2 + 1) * 64]; }
        step8(ring[s % (D + 1)][0], ring[s % (D + 1)][1], src[s >> 4][s & 15], dst);
        if (s + D < 64) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Backward K-loop: dst = W^T * T(src), where T is applied block by block on the fly and T(src) — the
// pre-activation gradient of the layer below — is stored row-major as it is produced:
//   MODE 0: T = identity                          (xyz_encoding_final has no activation)
//   MODE 1: T = [h > 0] * src                     (ReLU mask from the saved activation row `hrow`)
//   MODE 2: T = [h > 0] * (src + dsig * w_sigma)  (h8 also feeds the sigma head)
template <int MODE>
__device__ __forceinline__ void kloop_bwd(const f32x4* __restrict__ p /* + lane */, const f32x16 (&src)[8], f32x16 (&dst)[8],
                                          const float* __restrict__ hrow, const float* __restrict__ wsig, float dsig,
                                          float* __restrict__ save_row, int h, bool row_ok)
{
    constexpr int D = 4;
    f32x4 ring[D + 1][2];
#pragma unroll
    for (int s = 0; s < D; ++s) { ring[s][0] = p[(s * 2) * 64]; ring[s][1] = p[(s * 2 + 1) * 64]; }
    f32x4 hv[4], hn[4];
    if (MODE != 0) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) hv[rq] = *(const f32x4*)(hrow + 8 * rq + 4 * h);
    }
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        f32x16 cur;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = src[b][r];
            if (MODE == 2) v += dsig * (h ? wsig[(b * 16 + r) * 2 + 1] : wsig[(b * 16 + r) * 2]);
            if (MODE != 0) v = hv[r >> 2][r & 3] > 0.f ? v : 0.f;
            cur[r] = v;
        }
        if (MODE != 0 && b + 1 < 8) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) hn[rq] = *(const f32x4*)(hrow + 32 * (b + 1) + 8 * rq + 4 * h);
        }
        if (row_ok) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                f32x4 o = {cur[4 * rq], cur[4 * rq + 1], cur[4 * rq + 2], cur[4 * rq + 3]};
                *(f32x4*)(save_row + 32 * b + 8 * rq + 4 * h) = o;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int s = b * 16 + r;
            if (s + D < 128) { ring[(s + D) % (D + 1)][0] = p[((s + D) * 2) * 64]; ring[(s + D) % (D + 1)][1] = p[((s + D) * 2 + 1) * 64]; }
            step8(ring[s % (D + 1)][0], ring[s % (D + 1)][1], cur[r], dst);
            if (s + D < 128) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE != 0) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) hv[rq] = hn[rq];
        }
    }
}

__global__ void __launch_bounds__(256) k_mlp_bwd(NfMlpLayout L, NfMlpLayoutT T, const float* __restrict__ packed,
                                                 const float* __restrict__ packed_t, const float* __restrict__ acts,
                                                 const int* __restrict__ n_rows, int max_rows,
                                                 const int* __restrict__ row_sample, const float4* __restrict__ rgbsigma,
                                                 const float4* __restrict__ d_rgbsigma, float* __restrict__ dpre)
{
    const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
    const int gwave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const int nrows = min(*n_rows, max_rows);
    const int ntiles = (nrows + 31) >> 5;
    for (int tile = gwave; tile < ntiles; tile += nwaves) {
        const int z0 = opaque_zero();
        const float* __restrict__ pk = packed + z0;
        const f32x4* PT4 = (const f32x4*)(packed_t + z0);
        const int row = tile * 32 + j;
        const bool valid = row < nrows;
        const float* arow = acts + (size_t)(valid ? row : 0) * NF_ACT_STRIDE;
        float* drow = dpre + (size_t)(valid ? row : 0) * NF_DPRE_STRIDE;
        float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f), g4 = o4;
        if (valid) {
            int sample = row_sample[row];
            o4 = rgbsigma[sample];
            g4 = d_rgbsigma[sample];
        }
        // rgb = sigmoid(z): dz = g * y (1 - y)
        const float dz0 = g4.x * o4.x * (1.f - o4.x), dz1 = g4.y * o4.y * (1.f - o4.y), dz2 = g4.z * o4.z * (1.f - o4.z);
        const float dsig = g4.w;
        if (valid && h == 0) *(float4*)(drow + 2432) = make_float4(dz0, dz1, dz2, dsig);

        // d(dir hidden) = W_rgb^T dz, masked by relu
        f32x16 dd[4];
        const float* wr = pk + L.off_wrgb;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                f32x4 hv = *(const f32x4*)(arow + 9 * 256 + 32 * b + 8 * rq + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int r = 4 * rq + e, k = (b * 16 + r) * 2;
                    float v = dz0 * (h ? wr[k + 1] : wr[k]) + dz1 * (h ? wr[128 + k + 1] : wr[128 + k]) +
                              dz2 * (h ? wr[256 + k + 1] : wr[256 + k]);
                    dd[b][r] = hv[e] > 0.f ? v : 0.f;
                }
            }
        if (valid) save_frag<4>(dd, drow + 9 * 256, h);

        // d(final) = W_dir[:, :256]^T dpre_dir   (raw; it is stored as dpre slot 8 by the next loop, MODE 0)
        f32x16 accA[8], accB[8];
        zero_frag<8>(accA);
        kloop_src4(PT4 + (T.off_dir >> 2) + lane, dd, accA);
        const float* ws_ = pk + L.off_wsig;
        // g = 8: d_h8 = W_final^T dpre_final
        zero_frag<8>(accB);
        kloop_bwd<0>(PT4 + (T.off_h[8] >> 2) + lane, accA, accB, nullptr, ws_, dsig, drow + 8 * 256, h, valid);
        // g = 7: slot 7 = [h8 > 0] (d_h8 + dsig w_sigma);  d_h7 = W_8^T slot 7
        zero_frag<8>(accA);
        kloop_bwd<2>(PT4 + (T.off_h[7] >> 2) + lane, accB, accA, arow + 7 * 256, ws_, dsig, drow + 7 * 256, h, valid);
#pragma unroll 1
        for (int g = 6; g >= 2; g -= 2) {
            zero_frag<8>(accB);
            kloop_bwd<1>(PT4 + (T.off_h[g] >> 2) + lane, accA, accB, arow + g * 256, ws_, dsig, drow + g * 256, h, valid);
            zero_frag<8>(accA);
            kloop_bwd<1>(PT4 + (T.off_h[g - 1] >> 2) + lane, accB, accA, arow + (g - 1) * 256, ws_, dsig,
                         drow + (g - 1) * 256, h, valid);
        }
        // accA = d_h1 (raw): slot 0 = [h1 > 0] d_h1
        if (valid) {
#pragma unroll
            for (int b = 0; b < 8; ++b)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    f32x4 hv = *(const f32x4*)(arow + 32 * b + 8 * rq + 4 * h);
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = hv[e] > 0.f ? accA[b][4 * rq + e] : 0.f;
                    *(f32x4*)(drow + 32 * b + 8 * rq + 4 * h) = o;
                }
        }
    }
}

extern "C" int nf_nerf_mlp_bwd(const float* packed, const float* packed_t, int cx, int cd, const float* acts,
                               const int32_t* n_rows, int max_rows, const int32_t* row_sample, const float* rgbsigma,
                               const float* d_rgbsigma, float* dpre, nf_stream_t stream)
{
    NF_CHECK_ARG(packed && packed_t && acts && n_rows && row_sample && rgbsigma && d_rgbsigma && dpre, "null pointer");
    if (max_rows <= 0) return NF_OK;
    NfMlpLayout L = mlp_layout(cx, cd);
    NfMlpLayoutT T = mlp_layout_t();
    int tiles = (max_rows + 31) / 32;
    // one tile per wave, as many workgroups as tiles need: the dispatcher then balances this launch against whatever else
    // is in flight (the other pass's backward on a second stream) instead of 256 persistent workgroups pinning every CU
    int blocks = (tiles + 3) / 4;
    hipLaunchKernelGGL(k_mlp_bwd, dim3(blocks), dim3(256), 0, (hipStream_t)stream, L, T, packed, packed_t, acts, n_rows,
                       max_rows, row_sample, (const float4*)rgbsigma, (const float4*)d_rgbsigma, dpre);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// ================================================================================================
// weight gradients: all 15 GEMMs  dW[m][n] = sum_rows dpre[row][a_col+m] * B[row][b_col+n]  of one NeRF in ONE
// batched launch on fp32 MFMA: a 128 x 128 output tile per WAVE (k_wgrad2 below), the rows split into slices, per-slice
// partial tiles written to P[slice][...] and summed by k_wgrad_reduce (deterministic, no atomics).
// Rounds 1-2 staged 32-row slabs of both operands through LDS for a 128 x 128 tile per 4-wave workgroup (two barriers per
// slab, three workgroups per CU): 0.55-0.58 of the matrix peak on 72 000 rows; double-buffered slabs with one barrier were
// slower (two workgroups per CU).  Neither HBM nor L2 bound it (the same time with every operand row aliased into 2.5 MB).
// ================================================================================================
#define WG_MAX_GEMMS 16
struct NfWgradGemm { int a_col, b_src, b_col, M, N, c_off, ldc, c_col, tile0, tiles_n, colsum; };
struct NfWgradPlan { int ngemm, ntiles, total; NfWgradGemm g[WG_MAX_GEMMS]; };

static NfWgradPlan wgrad_plan(int cx, int cd)
{
    NfWgradPlan P;
    int n = 0, off = 0, tiles = 0, last_a = -1;
    auto add = [&](int a_col, int b_src, int b_col, int M, int N, int c_off, int ldc, int c_col) {
        NfWgradGemm& g = P.g[n++];
        g.a_col = a_col; g.b_src = b_src; g.b_col = b_col; g.M = M; g.N = N; g.c_off = c_off; g.ldc = ldc; g.c_col = c_col;
        g.colsum = a_col != last_a;      // the first GEMM over a block of dpre columns also sums them (bias gradients)
        last_a = a_col;
        g.tile0 = tiles; g.tiles_n = (N + 127) / 128;
        tiles += ((M + 127) / 128) * g.tiles_n;
    };
    // output blob: the 12 weight tensors in nf_nerf_params_t order, each [out][in] row-major
    for (int l = 0; l < 8; ++l) {
        int in_dim = (l == 0) ? cx : (l == 4 ? cx + 256 : 256);
        if (l == 0) add(0, 1, 0, 256, cx, off, in_dim, 0);
        else if (l == 4) { add(l * 256, 1, 0, 256, cx, off, in_dim, 0); add(l * 256, 0, 3 * 256, 256, 256, off, in_dim, cx); }
        else add(l * 256, 0, (l - 1) * 256, 256, 256, off, in_dim, 0);
        off += 256 * in_dim;
    }
    add(8 * 256, 0, 7 * 256, 256, 256, off, 256, 0); off += 256 * 256;                       // xyz_encoding_final
    add(9 * 256, 0, 8 * 256, 128, 256, off, 256 + cd, 0);                                    // dir_encoding: [final | dir]
    add(9 * 256, 1, 8 * ((cx + 7) / 8), 128, cd, off, 256 + cd, 256); off += 128 * (256 + cd);   // dir features start at group qx
    add(2435, 0, 7 * 256, 1, 256, off, 256, 0); off += 256;                                  // sigma
    add(2432, 0, 9 * 256, 3, 128, off, 128, 0); off += 3 * 128;                              // rgb
    P.ngemm = n; P.ntiles = tiles; P.total = off;
    return P;
}

extern "C" size_t nf_nerf_wgrad_floats(int cx, int cd) { return (size_t)wgrad_plan(cx, cd).total; }

// ---- round 3: a 128 x 128 tile per WAVE, operands straight from global memory into MFMA fragments (no LDS, no barrier).
// Both operands are row-major in the reduction index (row = sample), so one 16-B load per lane covers, for lanes 0..31, 128
// consecutive columns of row k and, for lanes 32..63, of row k + 1 — exactly the K pair of one v_mfma_f32_32x32x2_f32 step.
// Lane i holds columns 4i .. 4i+3: component c of the A quad is a valid A fragment whose output ROW i stands for column 4i + c,
// component c' of the B quad a B fragment whose output COLUMN j stands for column 4j + c' (which column an MFMA row / column
// means is free, it is only a matter of where the result is stored).  Two loads therefore feed 16 MFMAs (1 024 cycles of
// the matrix pipe): 256 accumulator registers, one wave per SIMD, a 4-step register ring hides the load latency.  The four
// waves of a workgroup take four consecutive tiles of the plan — the 2 x 2 tiles of a 256 x 256 layer for the same row
// slice — so every operand block is wanted by two waves of ONE CU at about the same time.  The LDS-staged kernel above issued
// 2.6 vector instructions and one LDS access per MFMA and met two barriers per 32 rows: 0.58 of the matrix peak on 72 000
// rows against the bare loop's 0.95.
#define WG2_D 5
__global__ void __launch_bounds__(256) k_wgrad2(NfWgradPlan P, const float* __restrict__ dpre, const float* __restrict__ acts,
                                                const float* __restrict__ xtiles, int Q, int n_rows, int rows_per_slice,
                                                int nslices, float* __restrict__ partial, const int* __restrict__ n_rows_dev)
{
    __shared__ float4 wg2_ring[4][WG2_D + 1][2][64];          // per wave: WG2_D + 1 slots x (A quad, B quad) x 64 lanes = 12 KB
    const int lane = threadIdx.x & 63;
    const int v = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (n_rows_dev) {       // nf_nerf_wgrad_dev: the row count lives on the device (graph replay); the slicing nf_nerf_wgrad does on the host
        n_rows = min(__builtin_amdgcn_readfirstlane(*n_rows_dev), n_rows);         // (n_rows = the caller's capacity)
        if (n_rows <= 0) return;
        const int want = nslices;
        rows_per_slice = ((n_rows + want - 1) / want + 31) / 32 * 32;
        nslices = (n_rows + rows_per_slice - 1) / rows_per_slice;
    }
    if (v >= P.ntiles * nslices) return;
    const int by = v / P.ntiles, bx = v - by * P.ntiles;
    int gi = 0;
#pragma unroll 1
    for (int i = 1; i < P.ngemm; ++i) if (bx >= P.g[i].tile0) gi = i;
    const NfWgradGemm G = P.g[gi];
    const int t = bx - G.tile0;
    const int m0 = (t / G.tiles_n) * 128, n0 = (t % G.tiles_n) * 128;
    const int i32 = lane & 31, h = lane >> 5;
    const int r0 = by * rows_per_slice, r1 = min(n_rows, r0 + rows_per_slice);      // r0 is a multiple of 32
    const int nsteps = (r1 - r0 + 1) >> 1;
    // A: the quad of columns a_al + m0 + 4 i; the sigma row starts 3 columns into its quad (a_shift), rows of dW that do not
    // exist are never stored; lanes beyond the live span read the tile's first quad instead of running off the row
    const int a_shift = G.a_col & 3, a_al = G.a_col - a_shift;
    const int a_span = min(128, G.M - m0 + a_shift);
    const int f0 = G.b_col + n0 + 4 * i32;                     // first of this lane's four B columns
    const bool b_ok = G.b_src ? f0 < 8 * Q : 4 * i32 < G.N - n0;
    const int fb = b_ok ? f0 : G.b_col + n0;
    // addresses = a wave-uniform base (the slice's first row) + a 32-bit lane offset: row rel of the slice (clamped to its last
    // row, so that the ring's read-ahead and an odd last row stay inside the arrays) times the row pitch + the lane's column
    const char* const sA = (const char*)(dpre + (size_t)r0 * NF_DPRE_STRIDE);
    const char* const sB = G.b_src ? (const char*)(xtiles + (size_t)(r0 >> 5) * Q * 256) : (const char*)(acts + (size_t)r0 * NF_ACT_STRIDE);
    const unsigned cA = 4u * (unsigned)(a_al + m0 + (4 * i32 < a_span ? 4 * i32 : 0));
    const unsigned cB = G.b_src ? 4u * (unsigned)(((fb >> 3) * 2 + ((fb >> 2) & 1)) * 128) : 4u * (unsigned)fb;
    // A step reads the row pair (2 s, 2 s + 1) of the slice: the pair's start goes into the SCALAR base of the load (clamped to
    // the last pair that holds a live row, for the ring's read-ahead), the lane offset — row parity x pitch + the lane's column —
    // is a constant.  If the arrays end on an odd row, the upper half of the last pair reads the lower row instead (its
    // values are masked out anyway): nothing is read beyond row n_rows - 1.
    const int lim_even = (n_rows - 1 - r0) & ~1;
    const bool odd_end = ((n_rows - r0) & 1) != 0;
    const unsigned pitchB = G.b_src ? 16u : (unsigned)(NF_ACT_STRIDE * 4);
    const unsigned voA = cA + (unsigned)h * (unsigned)(NF_DPRE_STRIDE * 4), voB = cB + (unsigned)h * pitchB;
    const unsigned voA_last = odd_end ? cA : voA, voB_last = odd_end ? cB : voB;
    const unsigned xq4 = (unsigned)Q * 1024u;
    // live A components: all four for whole tiles; the sigma / rgb rows use 1 / 3 of them and skip the other MFMAs
    int cmask = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) if (c >= a_shift && c - a_shift < G.M - m0) cmask |= 1 << c;
    if (G.M - m0 + a_shift > 4) cmask = 15;
    f32x16 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const bool do_colsum = G.colsum && n0 == 0;
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
    auto k_loop = [&](auto full_tag, auto xb_tag, auto cs_tag) __attribute__((always_inline)) {
        constexpr bool CS = decltype(cs_tag)::value;              // this tile also sums its dpre columns (bias gradients)
        constexpr bool FULL = decltype(full_tag)::value;          // whole tiles: 16 MFMAs per step, no test inside the loop
        constexpr bool XB = decltype(xb_tag)::value;              // B from the X tiles
        // Read-ahead through a wave-private LDS ring filled by LDS-DMA (global_load_lds_dwordx4: each lane's 16 bytes land at
        // slot + 16 lane, the order ds_read_b128 reads them back).  Step s + WG2_D is requested while step s runs; the ring has
        // WG2_D + 1 slots and the loop is unrolled by as many steps, so every slot address is a constant.  Nothing of the ring
        // lives in registers across iterations except the operands of the NEXT step (read from LDS behind this step's MFMAs):
        // a register ring of loop-carried loads came back from the compiler as copies behind vmcnt(0), i.e. without read-ahead.
        // The compiler does not track LDS-DMA: the counted s_waitcnt are explicit.
        typedef __attribute__((address_space(1))) const void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        float4* const myring = &wg2_ring[threadIdx.x >> 6][0][0][0];
        auto dma_a = [&](int s, int slot) __attribute__((always_inline)) {
            const int r2 = min(2 * s, lim_even);                                      // scalar
            const char* pa = sA + (size_t)r2 * (size_t)(NF_DPRE_STRIDE * 4) + (r2 == lim_even ? voA_last : voA);
            __builtin_amdgcn_global_load_lds((gptr_t)pa, (lptr_t)(myring + (slot * 2 + 0) * 64), 16, 0, 0);
        };
        auto dma_b = [&](int s, int slot) __attribute__((always_inline)) {
            const int r2 = min(2 * s, lim_even);
            const char* pb = XB ? sB + ((size_t)(r2 >> 5) * xq4 + (size_t)(r2 & 31) * 16u)
                                : sB + (size_t)r2 * (size_t)(NF_ACT_STRIDE * 4);
            pb += r2 == lim_even ? voB_last : voB;
            __builtin_amdgcn_global_load_lds((gptr_t)pb, (lptr_t)(myring + (slot * 2 + 1) * 64), 16, 0, 0);
        };
        auto dma = [&](int s, int slot) __attribute__((always_inline)) { dma_a(s, slot); dma_b(s, slot); };
#pragma unroll
        for (int d = 0; d < WG2_D; ++d) dma(d, d);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (WG2_D - 1)) : "memory");
        // (the ring is read with inline-asm ds_read_b128: a read the compiler can see makes it wait vmcnt(0) for the DMA writes
        // it believes may alias — all of them, the read-ahead included)
        const unsigned ring_lane = (unsigned)(size_t)(lptr_t)myring + 16u * (unsigned)lane;
        f32x4 a, b;
        asm volatile("ds_read_b128 %0, %1" : "=v"(a) : "v"(ring_lane));
        asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(b) : "v"(ring_lane));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)::"memory");
        auto group = [&](int s0, auto mask_tag) __attribute__((always_inline)) {
            constexpr bool MASK = decltype(mask_tag)::value;      // the slice's last group: rows past its end contribute nothing
#pragma unroll
            for (int d = 0; d < WG2_D + 1; ++d) {
                const int s = s0 + d;
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (WG2_D - 2)) : "memory");       // step s + 1 has landed
                f32x4 an, bn;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(an) : "v"(ring_lane), "n"((((d + 1) % (WG2_D + 1)) * 2 + 0) * 1024));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bn) : "v"(ring_lane), "n"((((d + 1) % (WG2_D + 1)) * 2 + 1) * 1024));
                if (MASK && r0 + 2 * s + h >= r1) a = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (CS) { cs.x += a[0]; cs.y += a[1]; cs.z += a[2]; cs.w += a[3]; }
                const float av[4] = {a[0], a[1], a[2], a[3]}, bv[4] = {b[0], b[1], b[2], b[3]};
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (FULL || (cmask >> c & 1)) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[c][e] = MFMA32(av[c], bv[e], acc[c][e]);
                    }
                    if (c == 0 || c == 2) {       // the step's two read-ahead requests, each behind four MFMAs of its own: a VMEM issue
                                                  // holds the wave's issue slot about as long as one MFMA runs (slot of step s - 1: read out)
                        __builtin_amdgcn_sched_barrier(0);
                        if (c == 0) dma_a(s + WG2_D, (d + WG2_D) % (WG2_D + 1));
                        else dma_b(s + WG2_D, (d + WG2_D) % (WG2_D + 1));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(an), "+v"(bn)::"memory");
                a = an; b = bn;
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        const int nfull = (r1 - r0) / (2 * (WG2_D + 1)) * (WG2_D + 1);       // steps in whole, unmasked groups
#pragma unroll 1
        for (int s0 = 0; s0 < nfull; s0 += WG2_D + 1) group(s0, std::false_type{});
        if (nfull < nsteps) group(nfull, std::true_type{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    if (cmask == 15) {
        if (G.b_src) { if (do_colsum) k_loop(std::true_type{}, std::true_type{}, std::true_type{}); else k_loop(std::true_type{}, std::true_type{}, std::false_type{}); }
        else { if (do_colsum) k_loop(std::true_type{}, std::false_type{}, std::true_type{}); else k_loop(std::true_type{}, std::false_type{}, std::false_type{}); }
    } else k_loop(std::false_type{}, std::false_type{}, std::true_type{});
    float* const slice = partial + (size_t)by * (P.total + NF_DPRE_STRIDE);
    if (do_colsum) {        // bias gradients: the two row parities of a column quad sit in lanes i and i + 32
        cs.x += __shfl_xor(cs.x, 32, 64); cs.y += __shfl_xor(cs.y, 32, 64);
        cs.z += __shfl_xor(cs.z, 32, 64); cs.w += __shfl_xor(cs.w, 32, 64);
        if (h == 0) {
            const float cv[4] = {cs.x, cs.y, cs.z, cs.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int m = m0 + 4 * i32 + c - a_shift;
                if (m >= m0 && m < G.M && 4 * i32 < a_span) slice[P.total + G.a_col + m] = cv[c];
            }
        }
    }
    // D of MFMA (c, e): register r of lane (j = lane & 31, h) is output row i = (r & 3) + 8 (r >> 2) + 4 h, column j, i.e.
    // dW[m0 + 4 i + c - a_shift][n0 + 4 j + e]: the four e of a (c, r) are 16 consecutive bytes of one dW row
    float* const out = slice + G.c_off + G.c_col;
    const bool vec = (G.ldc & 3) == 0 && ((G.c_off + G.c_col + n0) & 3) == 0 && G.N - n0 >= 128;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (!(cmask >> c & 1)) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + 4 * ((r & 3) + 8 * (r >> 2) + 4 * h) + c - a_shift;
            if (m < m0 || m >= G.M) continue;
            float* o = out + (size_t)m * G.ldc + n0 + 4 * i32;
            if (vec) *(float4*)o = make_float4(acc[c][0][r], acc[c][1][r], acc[c][2][r], acc[c][3][r]);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (n0 + 4 * i32 + e < G.N) o[e] = acc[c][e][r];
            }
        }
    }
}

// ---- round 6: the four waves of a workgroup SHARE their operands (k_wgrad3).  k_wgrad2 runs at 0.59 of the matrix peak on the 63 000
// rows of a training step's fine pass and its time does not depend on how the rows are sliced (20 / 21 / 22 / 44 slices: 930 / 912 /
// 905 / 915 us): it is bound by the bytes it asks for.  Every wave streams its own A and B block (2 KB per two rows), although the
// 2 x 2 tiles of a layer need only two A and two B blocks between them: 3.0 GB of requests per launch, every one of them reaching the
// L2 (hit rate 0.18: the twin request arrives while the first is still in flight) and 2.4 GB the memory side, for 1.2 GB of operands.
// Here a workgroup is a JOB: up to four 128 x 128 tiles that share A / B blocks (a 256 x 256 layer: 2 + 2 blocks; the view branch:
// one A block, the two halves of `final` and the direction features; the heads: the dsigma / drgb quad and h8's halves, hd).  Wave w
// brings block w of the job into a workgroup-wide LDS ring with ONE LDS-DMA per step (1 KB; k_wgrad2: two), every wave reads its A and
// its B block from the ring: half the requests, each operand byte fetched once per job.  One s_barrier per step (16 MFMAs) in the middle
// of the step's MFMAs: behind it every wave's piece of step s + 1 has landed (each wave waits for its own DMA before the barrier) and the
// slot that the next DMA overwrites (ring of WG3_P + 1 steps, requests WG3_P steps ahead) was read before the barrier of the step before.
// Same per-tile arithmetic as k_wgrad2 — same MFMAs in the same order over the same slices — so the sums are bit-identical.
#ifndef WG3_P
#define WG3_P 6
#endif
#define WG3_D (WG3_P + 1)
#define WG_MAX_JOBS 16
struct NfWgradJob { int nloads, ntiles; int load[4]; /* gemm | kind << 8 | block << 16 (block = m0 or n0 in units of 128) */
                    int tile[4];  /* gemm | (m0 / 128) << 8 | (n0 / 128) << 12 | slot of A << 16 | slot of B << 20 */ };
struct NfWgradJobs { int njobs; NfWgradJob j[WG_MAX_JOBS]; };

static NfWgradJobs wgrad_jobs(const NfWgradPlan& P)
{
    NfWgradJobs J;
    J.njobs = 0;
    int open_al = -1, open_span = 0;         // the A block of the job that may still take tiles (one m-tile GEMMs only)
    for (int gi = 0; gi < P.ngemm; ++gi) {
        const NfWgradGemm& g = P.g[gi];
        const int mt = (g.M + 127) / 128, nt = g.tiles_n;
        const int a_shift = g.a_col & 3, a_al = g.a_col - a_shift, a_span = g.M + a_shift < 128 ? g.M + a_shift : 128;
        if (mt == 2) {
            NfWgradJob& j = J.j[J.njobs++];
            j.nloads = 2 + nt; j.ntiles = 2 * nt;
            j.load[0] = gi | 0 << 8 | 0 << 16; j.load[1] = gi | 0 << 8 | 1 << 16;
            for (int b = 0; b < nt; ++b) j.load[2 + b] = gi | 1 << 8 | b << 16;
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < nt; ++b) j.tile[a * nt + b] = gi | a << 8 | b << 12 | a << 16 | (2 + b) << 20;
            open_al = -1;
            continue;
        }
        bool fits = false;
        if (open_al == a_al && (open_span == a_span || (open_span <= 4 && a_span <= 4))) {
            const NfWgradJob& o = J.j[J.njobs - 1];
            fits = o.nloads + nt <= 4 && o.ntiles + nt <= 4;
        }
        if (!fits) {
            NfWgradJob& j = J.j[J.njobs++];
            j.nloads = 1; j.ntiles = 0;
            j.load[0] = gi | 0 << 8 | 0 << 16;
            open_al = a_al; open_span = a_span;
        }
        NfWgradJob& j = J.j[J.njobs - 1];
        for (int b = 0; b < nt; ++b) {
            j.tile[j.ntiles++] = gi | 0 << 8 | b << 12 | 0 << 16 | j.nloads << 20;
            j.load[j.nloads++] = gi | 1 << 8 | b << 16;
        }
    }
    return J;
}

__global__ void __launch_bounds__(256) k_wgrad3(NfWgradPlan P, NfWgradJobs J, const float* __restrict__ dpre, const float* __restrict__ acts,
                                                const float* __restrict__ xtiles, int Q, int n_rows, int rows_per_slice,
                                                int nslices, float* __restrict__ partial, const int* __restrict__ n_rows_dev)
{
    __shared__ float4 wg3_ring[WG3_D][4][64];          // WG3_D steps x 4 blocks x 64 lanes x 16 B = 28 KB
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (n_rows_dev) {       // nf_nerf_wgrad_dev: the row count lives on the device (graph replay); the slicing nf_nerf_wgrad does on the host
        n_rows = min(__builtin_amdgcn_readfirstlane(*n_rows_dev), n_rows);         // (n_rows = the caller's capacity)
        if (n_rows <= 0) return;
        const int want = nslices;
        rows_per_slice = ((n_rows + want - 1) / want + 31) / 32 * 32;
        nslices = (n_rows + rows_per_slice - 1) / rows_per_slice;
    }
    const int by = blockIdx.x / J.njobs, ji = blockIdx.x - by * J.njobs;
    if (by >= nslices) return;
    const NfWgradJob& JB = J.j[ji];
    const int i32 = lane & 31, h = lane >> 5;
    const int r0 = by * rows_per_slice, r1 = min(n_rows, r0 + rows_per_slice);      // r0 is a multiple of 32
    const int nsteps = (r1 - r0 + 1) >> 1;
    const int lim_even = (n_rows - 1 - r0) & ~1;
    const bool odd_end = ((n_rows - r0) & 1) != 0;
    const unsigned xq4 = (unsigned)Q * 1024u;
    // ---- this wave's LOAD: block w of the job (address arithmetic of k_wgrad2's dma_a / dma_b); a wave beyond the job's blocks
    // repeats block 0 into its own (unread) ring slot, so that the loop has no branch on it
    const char* sL;
    unsigned voL, voL_last;
    unsigned pitchS, xqS, x16S;            // byte offset of row pair r2: r2 * pitchS + (r2 >> 5) * xqS + (r2 & 31) * x16S
    {
        const int ld = JB.load[w < JB.nloads ? w : 0];
        const NfWgradGemm& GL = P.g[ld & 255];
        const int blk = (ld >> 16) * 128;
        if (((ld >> 8) & 255) == 0) {           // an A block: the quad of dpre columns a_al + m0 + 4 i
            const int a_shift = GL.a_col & 3, a_al = GL.a_col - a_shift;
            const int a_span = min(128, GL.M - blk + a_shift);
            const unsigned cA = 4u * (unsigned)(a_al + blk + (4 * i32 < a_span ? 4 * i32 : 0));
            sL = (const char*)(dpre + (size_t)r0 * NF_DPRE_STRIDE);
            pitchS = (unsigned)(NF_DPRE_STRIDE * 4); xqS = 0; x16S = 0;
            voL = cA + (unsigned)h * pitchS;
            voL_last = odd_end ? cA : voL;
        } else {                                // a B block: activations (row-major) or the feature tiles
            const int f0 = GL.b_col + blk + 4 * i32;
            const bool b_ok = GL.b_src ? f0 < 8 * Q : 4 * i32 < GL.N - blk;
            const int fb = b_ok ? f0 : GL.b_col + blk;
            const bool xL = GL.b_src != 0;
            sL = xL ? (const char*)(xtiles + (size_t)(r0 >> 5) * Q * 256) : (const char*)(acts + (size_t)r0 * NF_ACT_STRIDE);
            const unsigned cB = xL ? 4u * (unsigned)(((fb >> 3) * 2 + ((fb >> 2) & 1)) * 128) : 4u * (unsigned)fb;
            const unsigned pitchL = xL ? 16u : (unsigned)(NF_ACT_STRIDE * 4);
            pitchS = xL ? 0u : pitchL; xqS = xL ? xq4 : 0u; x16S = xL ? 16u : 0u;
            voL = cB + (unsigned)h * pitchL;
            voL_last = odd_end ? cB : voL;
        }
    }
    // ---- this wave's TILE
    const bool has_tile = w < JB.ntiles;
    const int td = has_tile ? JB.tile[w] : JB.tile[0];
    const NfWgradGemm G = P.g[td & 255];
    const int m0 = ((td >> 8) & 15) * 128, n0 = ((td >> 12) & 15) * 128;
    const int slotA = (td >> 16) & 15, slotB = (td >> 20) & 15;
    const int a_shift = G.a_col & 3;
    const int a_span = min(128, G.M - m0 + a_shift);
    int cmask = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) if (c >= a_shift && c - a_shift < G.M - m0) cmask |= 1 << c;
    if (G.M - m0 + a_shift > 4) cmask = 15;
    if (!has_tile) cmask = 0;
    f32x16 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const bool do_colsum = has_tile && G.colsum && n0 == 0;
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    float4* const ring = &wg3_ring[0][0][0];
    const unsigned ring_lane = (unsigned)(size_t)(lptr_t)ring + 16u * (unsigned)lane;
    const unsigned rdA = ring_lane + 1024u * (unsigned)slotA, rdB = ring_lane + 1024u * (unsigned)slotB;
    auto dma = [&](int s, int slot) __attribute__((always_inline)) {
        const int r2 = min(2 * s, lim_even);                                      // scalar
        const size_t off = (size_t)((unsigned)r2 * pitchS) + (size_t)((unsigned)(r2 >> 5) * xqS) + (size_t)((unsigned)(r2 & 31) * x16S);
        const char* pa = sL + off + (r2 == lim_even ? voL_last : voL);
        __builtin_amdgcn_global_load_lds((gptr_t)pa, (lptr_t)(ring + (slot * 4 + w) * 64), 16, 0, 0);
    };
    auto k_loop = [&](auto full_tag, auto cs_tag) __attribute__((always_inline)) {
        constexpr bool CS = decltype(cs_tag)::value;              // this tile also sums its dpre columns (bias gradients)
        constexpr bool FULL = decltype(full_tag)::value;          // whole tiles: 16 MFMAs per step, no test inside the loop
#pragma unroll
        for (int d = 0; d < WG3_P; ++d) dma(d, d);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WG3_P - 1) : "memory");          // step 0 has landed (this wave's piece)
        __builtin_amdgcn_s_barrier();                                                // ... and everybody's
        f32x4 a, b;
        asm volatile("ds_read_b128 %0, %1" : "=v"(a) : "v"(rdA));
        asm volatile("ds_read_b128 %0, %1" : "=v"(b) : "v"(rdB));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)::"memory");
        auto group = [&](int s0, auto mask_tag) __attribute__((always_inline)) {
            constexpr bool MASK = decltype(mask_tag)::value;      // the slice's last group: rows past its end contribute nothing
#pragma unroll
            for (int d = 0; d < WG3_D; ++d) {
                const int s = s0 + d;
                f32x4 an, bn;
                if (MASK && r0 + 2 * s + h >= r1) a = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (CS) { cs.x += a[0]; cs.y += a[1]; cs.z += a[2]; cs.w += a[3]; }
                const float av[4] = {a[0], a[1], a[2], a[3]}, bv[4] = {b[0], b[1], b[2], b[3]};
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (FULL || (cmask >> c & 1)) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[c][e] = MFMA32(av[c], bv[e], acc[c][e]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#ifndef NF_W3_NODMA                 // (dev ablations: garbage results, one thing removed from the instruction stream)
                    if (c == 0) dma(s + WG3_P, (d + WG3_P) % WG3_D);       // the slot of step s - 1: read before the barrier of step s - 1
#endif
                    if (c == 1) {
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WG3_P - 1) : "memory");       // this wave's piece of step s + 1 has landed
#ifndef NF_W3_NOBAR
                        __builtin_amdgcn_s_barrier();
#endif
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(an) : "v"(rdA), "n"(((d + 1) % WG3_D) * 4096));
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bn) : "v"(rdB), "n"(((d + 1) % WG3_D) * 4096));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(an), "+v"(bn)::"memory");
                a = an; b = bn;
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // every wave of the job walks the same number of steps (the barriers): whole groups, then one masked group
        const int nfull = (r1 - r0) / (2 * WG3_D) * WG3_D;       // steps in whole, unmasked groups
#pragma unroll 1
        for (int s0 = 0; s0 < nfull; s0 += WG3_D) group(s0, std::false_type{});
        if (nfull < nsteps) group(nfull, std::true_type{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    if (cmask == 15) { if (do_colsum) k_loop(std::true_type{}, std::true_type{}); else k_loop(std::true_type{}, std::false_type{}); }
    else k_loop(std::false_type{}, std::true_type{});
    if (!has_tile) return;
    float* const slice = partial + (size_t)by * (P.total + NF_DPRE_STRIDE);
    if (do_colsum) {        // bias gradients: the two row parities of a column quad sit in lanes i and i + 32
        cs.x += __shfl_xor(cs.x, 32, 64); cs.y += __shfl_xor(cs.y, 32, 64);
        cs.z += __shfl_xor(cs.z, 32, 64); cs.w += __shfl_xor(cs.w, 32, 64);
        if (h == 0) {
            const float cv[4] = {cs.x, cs.y, cs.z, cs.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int m = m0 + 4 * i32 + c - a_shift;
                if (m >= m0 && m < G.M && 4 * i32 < a_span) slice[P.total + G.a_col + m] = cv[c];
            }
        }
    }
    float* const out = slice + G.c_off + G.c_col;
    const bool vec = (G.ldc & 3) == 0 && ((G.c_off + G.c_col + n0) & 3) == 0 && G.N - n0 >= 128;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (!(cmask >> c & 1)) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + 4 * ((r & 3) + 8 * (r >> 2) + 4 * h) + c - a_shift;
            if (m < m0 || m >= G.M) continue;
            float* o = out + (size_t)m * G.ldc + n0 + 4 * i32;
            if (vec) *(float4*)o = make_float4(acc[c][0][r], acc[c][1][r], acc[c][2][r], acc[c][3][r]);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (n0 + 4 * i32 + e < G.N) o[e] = acc[c][e][r];
            }
        }
    }
}

// partial[slice][total + NF_DPRE_STRIDE] -> dweights[total] | dbias[NF_DPRE_STRIDE]; one float4 per thread, the slices in
// groups of 4 independent loads (total and NF_DPRE_STRIDE are multiples of 4 floats)
__global__ void k_wgrad_reduce(const float* __restrict__ partial, int total, int nslices, float* __restrict__ out,
                               float* __restrict__ dbias, const int* __restrict__ n_rows_dev, int max_rows)
{
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int stride = total + NF_DPRE_STRIDE;
    if (i >= stride) return;
    if (n_rows_dev) {       // the slices k_wgrad2 wrote for this row count (0 rows: zeros)
        const int n = min(*n_rows_dev, max_rows);
        if (n <= 0) nslices = 0;
        else {
            const int rp = ((n + nslices - 1) / nslices + 31) / 32 * 32;
            nslices = (n + rp - 1) / rp;
        }
    }
    const float* p = partial + i;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int z = 0;
    for (; z + 4 <= nslices; z += 4) {
        const float4 a = *(const float4*)(p + (size_t)z * stride), b = *(const float4*)(p + (size_t)(z + 1) * stride);
        const float4 c = *(const float4*)(p + (size_t)(z + 2) * stride), d = *(const float4*)(p + (size_t)(z + 3) * stride);
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
        s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
        s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w;
    }
    for (; z < nslices; ++z) {
        const float4 a = *(const float4*)(p + (size_t)z * stride);
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    if (i < total) *(float4*)(out + i) = s;
    else if (dbias) *(float4*)(dbias + (i - total)) = s;
}

extern "C" size_t nf_nerf_wgrad_workspace_floats(int cx, int cd, int nslices)
{
    return ((size_t)wgrad_plan(cx, cd).total + NF_DPRE_STRIDE) * nslices;
}

extern "C" int nf_nerf_wgrad(const float* dpre, const float* acts, const float* X, int cx, int cd, int n_rows,
                             int nslices, float* workspace, float* dweights, float* dbias, nf_stream_t stream)
{
    NF_CHECK_ARG(dpre && acts && X && workspace && dweights, "null pointer");
    NF_CHECK_ARG(nslices >= 1 && nslices <= 65535, "bad slice count");
    NfWgradPlan P = wgrad_plan(cx, cd);
    hipStream_t st = (hipStream_t)stream;
    if (n_rows <= 0) {
        hipMemsetAsync(dweights, 0, sizeof(float) * P.total, st);
        if (dbias) hipMemsetAsync(dbias, 0, sizeof(float) * NF_DPRE_STRIDE, st);
        return NF_OK;
    }
    int rows_per = (n_rows + nslices - 1) / nslices;
    rows_per = (rows_per + 31) / 32 * 32;          // slices start on a 32-row boundary (the X tiles' granule)
    int ns = (n_rows + rows_per - 1) / rows_per;
#ifdef NF_WGRAD2
    hipLaunchKernelGGL(k_wgrad2, dim3((P.ntiles * ns + 3) / 4), dim3(256), 0, st, P, dpre, acts, X, (cx + 7) / 8 + (cd + 7) / 8, n_rows,
                       rows_per, ns, workspace, (const int*)nullptr);
#else
    const NfWgradJobs J = wgrad_jobs(P);
    hipLaunchKernelGGL(k_wgrad3, dim3(J.njobs * ns), dim3(256), 0, st, P, J, dpre, acts, X, (cx + 7) / 8 + (cd + 7) / 8, n_rows,
                       rows_per, ns, workspace, (const int*)nullptr);
#endif
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(((P.total + NF_DPRE_STRIDE) / 4 + 255) / 256), dim3(256), 0, st,
                       (const float*)workspace, P.total, ns, dweights, dbias, (const int*)nullptr, 0);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// The same launch pair with the row count in DEVICE memory (*n_rows, clamped to max_rows): what a replayed training step needs (a HIP
// graph's launch arguments are frozen; the active rows change from step to step).  The grid is sized for nslices slices; the
// kernels derive the slicing of nf_nerf_wgrad(..., n_rows = *n_rows, nslices, ...) themselves, so the sums are the same bit for bit.
extern "C" int nf_nerf_wgrad_dev(const float* dpre, const float* acts, const float* X, int cx, int cd, const int32_t* n_rows, int max_rows,
                                 int nslices, float* workspace, float* dweights, float* dbias, nf_stream_t stream)
{
    NF_CHECK_ARG(dpre && acts && X && workspace && dweights && n_rows, "null pointer");
    NF_CHECK_ARG(nslices >= 1 && nslices <= 65535 && max_rows >= 1, "bad slice count / capacity");
    NfWgradPlan P = wgrad_plan(cx, cd);
    hipStream_t st = (hipStream_t)stream;
#ifdef NF_WGRAD2
    hipLaunchKernelGGL(k_wgrad2, dim3((P.ntiles * nslices + 3) / 4), dim3(256), 0, st, P, dpre, acts, X, (cx + 7) / 8 + (cd + 7) / 8, max_rows,
                       0, nslices, workspace, (const int*)n_rows);
#else
    const NfWgradJobs J = wgrad_jobs(P);
    hipLaunchKernelGGL(k_wgrad3, dim3(J.njobs * nslices), dim3(256), 0, st, P, J, dpre, acts, X, (cx + 7) / 8 + (cd + 7) / 8, max_rows,
                       0, nslices, workspace, (const int*)n_rows);
#endif
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(((P.total + NF_DPRE_STRIDE) / 4 + 255) / 256), dim3(256), 0, st,
                       (const float*)workspace, P.total, nslices, dweights, dbias, (const int*)n_rows, max_rows);
    NF_CHECK_LAUNCH();
    return NF_OK;
}
