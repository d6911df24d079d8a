// nf_mlp_n.hip — fp32-MFMA NeRF MLP forward for SMALL launches (the training steps): one 32-sample tile per WORKGROUP,
// the 8 output blocks of a layer split over its 4 waves (2 blocks each; the 4-block view branch: 1 each).
//
// k_mlp_fwd (nf_mlp.hip) keeps a tile in ONE wave's registers for all layers: 10 416 dependent-in-order MFMAs = 0.35 ms per
// tile, one wave per SIMD — so a launch costs ceil(tiles / 1024) rounds of 0.35 ms whatever the fill of the last round, and a
// 4 096-ray training step (600 + 2 100 tiles) or a 1 024-ray end-to-end step (160 + 470 tiles) spends most of its MLP time
// on SIMDs that have nothing to do.  Here a tile occupies the four SIMDs of a CU for a quarter of that time: the rounds are
// 256 tiles of ~0.09 ms.  The price is that activations no longer stay in registers: every layer's 64 features per wave go
// to an LDS image act[feature][sample] (double-buffered, one barrier per layer) from which all four waves read their B
// operands back, one ds_read_b32 per K-step (lane-linear, conflict-free both ways).
//
// Same arithmetic as k_mlp_fwd, bit for bit: same MFMA, same K order (bias step first, X part, hidden part), the sigma / rgb
// heads summed by one wave in the same order from the LDS image.
//
// Weight blob (round 3): nf_nerf_pack_n / nf_nerf_pack_bwd_n re-arrange the standard blobs (same size, same part offsets) so that
// ONE 16-B load per lane holds a wave's operands of TWO K-steps (its two blocks x two steps; the one-block view branch: four
// steps): [pair][wave][lane][step-in-pair * 2 + block-in-wave].  With the standard layout a wave used 8 of the 16 bytes of a
// K-step's quad, i.e. one vector-memory instruction (and its 64-bit address arithmetic) per TWO MFMAs — the single wave of a
// SIMD spends about as long issuing that as one MFMA runs: a workgroup alone on its CU kept the matrix pipe 53 % busy.
#include "nf_mlp_layout.h"
#include <math.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

#define NN_ACT (256 * 32)          // floats of one activation image
// Weight operands are fetched with buffer loads: resource = the blob (scalar registers), scalar offset = part + K-step pair,
// vector offset = the lane's constant 16-B slot.  A global load from a per-lane 64-bit pointer paid two vector adds per load
// (the pair stride of 4 KB does not fit the instruction's immediate) — on the ALUs the fp32 MFMA runs on.
typedef unsigned n_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 n_wload(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0));
}
#ifndef NH_DEPTH
#define NH_DEPTH 4               // read-ahead of the hidden parts' K loops, in K-step pairs
#endif

struct NCtx {
    int lane, h, j, w, g, c0;      // wave w owns blocks 2w, 2w + 1 = half g = w >> 1, components c0, c0 + 1 of its 16-B operands
};

// bias K-step (A = bias, B = 1, C = 0) of an 8-block layer: [2][64][4]
__device__ __forceinline__ void n_bias2(const NCtx& c, const f32x4* __restrict__ p, f32x16 (&acc)[2])
{
    const f32x2 wv = *(const f32x2*)((const float*)(p + c.g * 64 + c.lane) + c.c0);
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    acc[0] = MFMA32(wv[0], 1.f, z);
    acc[1] = MFMA32(wv[1], 1.f, z);
}

// K-steps whose B operand comes from the feature matrix (25 groups x 4 steps), read from global X both times (layer 0 and the
// skip layer ~30 us later: 32 KB per tile that the L2 still holds; no LDS stash, so two workgroups fit a CU and each SIMD
// has a second tile's wave to issue from while the first one waits at a layer barrier).
template <int nq>
__device__ __forceinline__ void n_xpart2(const NCtx& c, __amdgpu_buffer_rsrc_t wr_, int poff /* part offset in bytes, N layout */,
                                         const f32x4* __restrict__ xt /* + lane */, f32x16 (&acc)[2])
{
    const unsigned voff = (unsigned)(c.w * 64 + c.lane) * 16u;          // pair stride: 4 KB; a group of 4 K-steps = 2 pairs
    // a group is only 4 K-steps x 2 MFMAs = 512 cycles here: X (HBM the first time, L2 the second) is requested XD groups
    // ahead, the weights (L2) two groups ahead
    constexpr int XD = 6;
    f32x4 xr[XD];
    f32x4 wr[2][2];
#pragma unroll
    for (int q = 0; q < XD; ++q) xr[q] = q < nq ? xt[q * 64] : xt[0];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i) wr[q][i] = n_wload(wr_, voff, poff + (q * 2 + i) * 4096);
#pragma unroll
    for (int q = 0; q < nq; ++q) {
        const f32x4 xv = xr[0];
        f32x4 w[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) w[i] = wr[0][i];
#pragma unroll
        for (int k = 0; k + 1 < XD; ++k) xr[k] = xr[k + 1];
#pragma unroll
        for (int i = 0; i < 2; ++i) wr[0][i] = wr[1][i];
        if (q + XD < nq) xr[XD - 1] = xt[(q + XD) * 64];
        if (q + 2 < nq) {
#pragma unroll
            for (int i = 0; i < 2; ++i) wr[1][i] = n_wload(wr_, voff, poff + ((q + 2) * 2 + i) * 4096);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            acc[0] = MFMA32(w[i][0], xv[2 * i], acc[0]);
            acc[1] = MFMA32(w[i][1], xv[2 * i], acc[1]);
            acc[0] = MFMA32(w[i][2], xv[2 * i + 1], acc[0]);
            acc[1] = MFMA32(w[i][3], xv[2 * i + 1], acc[1]);
        }
    }
}

// hidden part: acc += W * act, act read from the LDS image (already activated); 128 K-steps = 8 source blocks x 16
template <int NS = 128>
__device__ __forceinline__ void n_hpart2(const NCtx& c, __amdgpu_buffer_rsrc_t wr_, int poff /* part offset in bytes, N layout */,
                                         const float* __restrict__ act, f32x16 (&acc)[2])
{
    constexpr int NP = NS / 2, D = NH_DEPTH;          // K-step pairs; read-ahead in pairs
    const unsigned voff = (unsigned)(c.w * 64 + c.lane) * 16u;
    const float* ap = act + (4 * c.h) * 32 + c.j;           // feature frag_feature(b, r, h) = 32 b + (r & 3) + 8 (r >> 2) + 4 h
    f32x4 ring[D + 1];
    float b0[D + 1], b1[D + 1];
#define NH_F(S) ((32 * ((S) >> 4) + ((S) & 3) + 8 * (((S) & 15) >> 2)) * 32)
#pragma unroll
    for (int s = 0; s < D; ++s) {
        ring[s] = n_wload(wr_, voff, poff + s * 4096);
        b0[s] = ap[NH_F(2 * s)];
        b1[s] = ap[NH_F(2 * s + 1)];
    }
#pragma unroll
    for (int s = 0; s < NP; ++s) {
        if (s + D < NP) {
            const int t = s + D;
            ring[t % (D + 1)] = n_wload(wr_, voff, poff + t * 4096);
            b0[t % (D + 1)] = ap[NH_F(2 * t)];
            b1[t % (D + 1)] = ap[NH_F(2 * t + 1)];
        }
        const f32x4 wv = ring[s % (D + 1)];
        const float x0 = b0[s % (D + 1)], x1 = b1[s % (D + 1)];
        acc[0] = MFMA32(wv[0], x0, acc[0]);
        acc[1] = MFMA32(wv[1], x0, acc[1]);
        acc[0] = MFMA32(wv[2], x1, acc[0]);
        acc[1] = MFMA32(wv[3], x1, acc[1]);
        if (s + D < NP) {      // one VMEM and the LDS reads of a pair, each in an MFMA shadow
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#undef NH_F
}

// the wave's two blocks -> LDS image (RELU or raw) and, when training, the saved-activation row (row-major, as k_mlp_fwd)
template <bool RELU, bool SAVE>
__device__ __forceinline__ void n_store2(const NCtx& c, const f32x16 (&acc)[2], float* __restrict__ act, float* __restrict__ save_row,
                                         bool row_ok)
{
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int b = 2 * c.w + i;
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            v[r] = RELU ? fmaxf(acc[i][r], 0.f) : acc[i][r];
            act[(32 * b + (r & 3) + 8 * (r >> 2) + 4 * c.h) * 32 + c.j] = v[r];
        }
        if (SAVE && row_ok) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                f32x4 o = {v[4 * rq], v[4 * rq + 1], v[4 * rq + 2], v[4 * rq + 3]};
                *(f32x4*)(save_row + 32 * b + 8 * rq + 4 * c.h) = o;
            }
        }
    }
}

template <bool SAVE, int QX, int QD>
__global__ void __launch_bounds__(256) k_mlp_fwd_n(NfMlpLayout L, const float* __restrict__ packed, const float* __restrict__ X,
                                                   const int* __restrict__ n_rows, int max_rows,
                                                   const int* __restrict__ row_sample, float4* __restrict__ rgbsigma,
                                                   float* __restrict__ acts)
{
    extern __shared__ float nlds[];        // act image A, act image B
    float* actA = nlds;
    float* actB = nlds + NN_ACT;
    NCtx c;
    c.lane = threadIdx.x & 63; c.h = c.lane >> 5; c.j = c.lane & 31; c.w = threadIdx.x >> 6; c.g = c.w >> 1; c.c0 = 2 * (c.w & 1);
    const int nrows = min(*n_rows, max_rows);
    const int ntiles = (nrows + 31) >> 5;
    constexpr int Q = QX + QD;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int z0 = opaque_zero();
        const float* __restrict__ pk = packed + z0;
        const f32x4* P4 = (const f32x4*)pk;
        const __amdgpu_buffer_rsrc_t wr_ = __builtin_amdgcn_make_buffer_rsrc((void*)packed, 0, L.total * 4, 0x27000);
        const f32x4* xt = (const f32x4*)X + (size_t)tile * Q * 64 + c.lane;
        const int row = tile * 32 + c.j;
        const bool row_ok = row < nrows;
        float* arow = SAVE ? acts + (size_t)(row_ok ? row : 0) * NF_ACT_STRIDE : nullptr;
        f32x16 acc[2];
        f32x4 xdir[QD];         // the view-direction feature groups, requested now, used ~100 us later
#pragma unroll
        for (int q = 0; q < QD; ++q) xdir[q] = xt[(QX + q) * 64];

        // layer 0 = xyz_encoding_1 -> h1 in A
        n_bias2(c, P4 + (L.off_bstep[0] >> 2), acc);
        n_xpart2<QX>(c, wr_, L.off_x[0] * 4 + z0, xt, acc);
        n_store2<true, SAVE>(c, acc, actA, arow, row_ok);
        __syncthreads();
        float* cur = actA;
        float* nxt = actB;
        float sigma = 0.f;
#pragma unroll 1
        for (int l = 1; l < 9; ++l) {
            n_bias2(c, P4 + (L.off_bstep[l] >> 2), acc);
            if (L.off_x[l] >= 0) n_xpart2<QX>(c, wr_, L.off_x[l] * 4 + z0, xt, acc);
            n_hpart2(c, wr_, L.off_h[l] * 4 + z0, cur, acc);
            if (l == 8) {
                // `cur` holds h8: the sigma head, by wave 0, in k_mlp_fwd's order (its lanes sum their own 128 features)
                if (c.w == 0) {
                    const float* ws_ = pk + L.off_wsig;
                    float part = 0.f;
#pragma unroll
                    for (int b = 0; b < 8; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float w0 = ws_[(b * 16 + r) * 2], w1 = ws_[(b * 16 + r) * 2 + 1];
                            part += cur[(32 * b + (r & 3) + 8 * (r >> 2) + 4 * c.h) * 32 + c.j] * (c.h ? w1 : w0);
                        }
                    sigma = part + __shfl_xor(part, 32, 64) + pk[L.off_bsig];
                }
                n_store2<false, SAVE>(c, acc, nxt, SAVE ? arow + 8 * 256 : nullptr, row_ok);      // xyz_encoding_final: no activation
            } else {
                n_store2<true, SAVE>(c, acc, nxt, SAVE ? arow + l * 256 : nullptr, row_ok);        // h_{l+1}
            }
            __syncthreads();
            float* t = cur; cur = nxt; nxt = t;
        }
        // view branch: hd = relu(W_dir [final | dir feats] + b), one block per wave; `cur` holds final
        f32x16 hd;
        {
            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const f32x4* pb = P4 + (L.off_bstep_dir >> 2) + c.lane;
            hd = MFMA32(((const float*)pb)[c.w], 1.f, z);
            const unsigned voff = (unsigned)(c.w * 64 + c.lane) * 16u;               // N layout: [group of 4 steps][wave][64] x 16 B
#pragma unroll
            for (int q = 0; q < QD; ++q) {
                const f32x4 wv = n_wload(wr_, voff, L.off_dir_x * 4 + z0 + q * 4096);
#pragma unroll
                for (int i = 0; i < 4; ++i) hd = MFMA32(wv[i], xdir[q][i], hd);
            }
            const float* ap = cur + (4 * c.h) * 32 + c.j;
#pragma unroll 4
            for (int q = 0; q < 32; ++q) {
                const f32x4 wv = n_wload(wr_, voff, L.off_dir_h * 4 + z0 + q * 4096);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int s = 4 * q + i;
                    hd = MFMA32(wv[i], ap[(32 * (s >> 4) + (s & 3) + 8 * ((s & 15) >> 2)) * 32], hd);
                }
            }
        }
        {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                v[r] = fmaxf(hd[r], 0.f);
                nxt[(32 * c.w + (r & 3) + 8 * (r >> 2) + 4 * c.h) * 32 + c.j] = v[r];
            }
            if (SAVE && row_ok) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    f32x4 o = {v[4 * rq], v[4 * rq + 1], v[4 * rq + 2], v[4 * rq + 3]};
                    *(f32x4*)(arow + 9 * 256 + 32 * c.w + 8 * rq + 4 * c.h) = o;
                }
            }
        }
        __syncthreads();
        if (c.w == 0) {
            const float* wr = pk + L.off_wrgb;
            float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = nxt[(32 * b + (r & 3) + 8 * (r >> 2) + 4 * c.h) * 32 + c.j];
                    const int k = (b * 16 + r) * 2;
                    c0 += v * (c.h ? wr[k + 1] : wr[k]);
                    c1 += v * (c.h ? wr[128 + k + 1] : wr[128 + k]);
                    c2 += v * (c.h ? wr[256 + k + 1] : wr[256 + k]);
                }
            c0 += __shfl_xor(c0, 32, 64); c1 += __shfl_xor(c1, 32, 64); c2 += __shfl_xor(c2, 32, 64);
            c0 += pk[L.off_brgb]; c1 += pk[L.off_brgb + 1]; c2 += pk[L.off_brgb + 2];
            if (c.h == 0 && row_ok) {
                float4 o;
                o.x = 1.f / (1.f + expf(-c0)); o.y = 1.f / (1.f + expf(-c1)); o.z = 1.f / (1.f + expf(-c2)); o.w = sigma;
                rgbsigma[row_sample[row]] = o;
            }
        }
        __syncthreads();        // the images are rewritten by the next tile
    }
}

// ------------------------------------------------------------------------------------------------
// N layout of a weight blob (see the header): regions of 8-block K-steps ([step][2][64][4], 512 floats) become
// [pair][wave][lane][2 sp + i] (block = 2 wave + i, step = 2 pair + sp); regions of 4-block K-steps ([step][64][4], 256 floats)
// become [group of 4][wave = block][lane][step in group]; everything else is copied.  Parts hold an even (resp. multiple-of-4)
// number of steps and follow each other without gaps, so one formula per region covers them all and part offsets stay valid.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_mlp_pack_n(const float* __restrict__ in, float* __restrict__ out, int total, int end8, int end4)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    int o = idx;
    if (idx < end8) {
        const int S = idx >> 9, wi = idx & 511, g = wi >> 8, lane = (wi >> 2) & 63, e = wi & 3;
        const int blk = 4 * g + e, w = blk >> 1, i = blk & 1;
        o = (((S >> 1) * 4 + w) * 64 + lane) * 4 + 2 * (S & 1) + i;
    } else if (idx < end4) {
        const int r = idx - end8, T = r >> 8, wi = r & 255, lane = wi >> 2, w = wi & 3;
        o = end8 + (((T >> 2) * 4 + w) * 64 + lane) * 4 + (T & 3);
    }
    out[o] = in[idx];
}

extern "C" int nf_nerf_pack_n(const float* packed, int cx, int cd, float* packed_n, nf_stream_t stream)
{
    NF_CHECK_ARG(packed && packed_n && packed != packed_n, "null pointer / in-place");
    NfMlpLayout L = mlp_layout(cx, cd);
    NF_CHECK_ARG((L.qx * 4) % 2 == 0 && (L.qd * 4) % 4 == 0, "odd part length");
    hipLaunchKernelGGL(k_mlp_pack_n, dim3((L.total + 255) / 256), dim3(256), 0, (hipStream_t)stream, packed, packed_n, L.total, L.off_dir_h,
                       L.off_wsig);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_nerf_pack_bwd_n(const float* packed_t, float* packed_tn, nf_stream_t stream)
{
    NF_CHECK_ARG(packed_t && packed_tn && packed_t != packed_tn, "null pointer / in-place");
    NfMlpLayoutT T = mlp_layout_t();
    hipLaunchKernelGGL(k_mlp_pack_n, dim3((T.total + 255) / 256), dim3(256), 0, (hipStream_t)stream, packed_t, packed_tn, T.total, T.total, T.total);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// ================================================================================================
// backward (data gradient) for small launches: k_mlp_bwd (nf_mlp.hip) cut the same way — a tile per workgroup, every wave
// two of the eight 32-feature blocks of each layer's input gradient, the pre-activation gradients exchanged through the two
// LDS images.  Per layer g = 8 .. 1:   slot g = T_g(raw gradient)  -> dpre row (global) + LDS image;  barrier;
//                                       raw gradient of the layer below = W_g^T . image   (128 K-steps x 2 MFMAs per wave)
// with T_8 = identity (xyz_encoding_final has no activation), T_7 = [h8 > 0] (. + dsigma w_sigma), T_g = [h_(g+1) > 0] otherwise;
// slot 9 (the view branch's hidden units, from the rgb head) and slot 0 come before and after the loop.  Same transposed
// blob (nf_nerf_pack_bwd), same K order from a zero accumulator: the results equal k_mlp_bwd's bit for bit.
// Why: one wave per tile for ~0.45 ms makes a launch cost ceil(tiles / 1024) rounds whatever the fill of the last one — the
// fine pass of a 4 x 1024-ray training step (2 250 tiles) paid three rounds for 2.2, its coarse pass (220 tiles) a whole round.
// ================================================================================================
template <int MODE>
__device__ __forceinline__ void n_bwd_slot(const NCtx& c, const f32x16 (&raw)[2], const float* __restrict__ hrow, const float* __restrict__ wsig,
                                           float dsig, float* __restrict__ img, float* __restrict__ save_row, bool row_ok)
{
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int b = 2 * c.w + i;
        float v[16];
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            f32x4 hv = {1.f, 1.f, 1.f, 1.f};
            if (MODE != 0) hv = *(const f32x4*)(hrow + 32 * b + 8 * rq + 4 * c.h);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * rq + e;
                float x = raw[i][r];
                if (MODE == 2) x += dsig * (c.h ? wsig[(b * 16 + r) * 2 + 1] : wsig[(b * 16 + r) * 2]);
                if (MODE != 0) x = hv[e] > 0.f ? x : 0.f;
                v[r] = x;
                if (img) img[(32 * b + (r & 3) + 8 * (r >> 2) + 4 * c.h) * 32 + c.j] = x;
            }
        }
        if (row_ok) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                f32x4 o = {v[4 * rq], v[4 * rq + 1], v[4 * rq + 2], v[4 * rq + 3]};
                *(f32x4*)(save_row + 32 * b + 8 * rq + 4 * c.h) = o;
            }
        }
    }
}

__device__ __forceinline__ void n_zero2(f32x16 (&acc)[2])
{
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
}

__global__ void __launch_bounds__(256) k_mlp_bwd_n(NfMlpLayout L, NfMlpLayoutT T, const float* __restrict__ packed,
                                                   const float* __restrict__ packed_t, const float* __restrict__ acts,
                                                   const int* __restrict__ n_rows, int max_rows, const int* __restrict__ row_sample,
                                                   const float4* __restrict__ rgbsigma, const float4* __restrict__ d_rgbsigma,
                                                   float* __restrict__ dpre)
{
    extern __shared__ float nlds[];
    float* cur = nlds;
    float* nxt = nlds + NN_ACT;
    NCtx c;
    c.lane = threadIdx.x & 63; c.h = c.lane >> 5; c.j = c.lane & 31; c.w = threadIdx.x >> 6; c.g = c.w >> 1; c.c0 = 2 * (c.w & 1);
    const int nrows = min(*n_rows, max_rows);
    const int ntiles = (nrows + 31) >> 5;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int z0 = opaque_zero();
        const float* __restrict__ pk = packed + z0;
        const __amdgpu_buffer_rsrc_t wt_ = __builtin_amdgcn_make_buffer_rsrc((void*)packed_t, 0, T.total * 4, 0x27000);
        const int row = tile * 32 + c.j;
        const bool valid = row < nrows;
        const float* arow = acts + (size_t)(valid ? row : 0) * NF_ACT_STRIDE;
        float* drow = dpre + (size_t)(valid ? row : 0) * NF_DPRE_STRIDE;
        float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f), g4 = o4;
        if (valid) {
            const int sample = row_sample[row];
            o4 = rgbsigma[sample];
            g4 = d_rgbsigma[sample];
        }
        // rgb = sigmoid(z): dz = g * y (1 - y)
        const float dz0 = g4.x * o4.x * (1.f - o4.x), dz1 = g4.y * o4.y * (1.f - o4.y), dz2 = g4.z * o4.z * (1.f - o4.z);
        const float dsig = g4.w;
        if (valid && c.h == 0 && c.w == 0) *(float4*)(drow + 2432) = make_float4(dz0, dz1, dz2, dsig);
        // slot 9: d(view-branch hidden) = [hd > 0] W_rgb^T dz, block w by wave w
        {
            const float* wr = pk + L.off_wrgb;
            const int b = c.w;
            float v[16];
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const f32x4 hv = *(const f32x4*)(arow + 9 * 256 + 32 * b + 8 * rq + 4 * c.h);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * rq + e, k = (b * 16 + r) * 2;
                    const float x = dz0 * (c.h ? wr[k + 1] : wr[k]) + dz1 * (c.h ? wr[128 + k + 1] : wr[128 + k]) +
                                    dz2 * (c.h ? wr[256 + k + 1] : wr[256 + k]);
                    v[r] = hv[e] > 0.f ? x : 0.f;
                    cur[(32 * b + (r & 3) + 8 * (r >> 2) + 4 * c.h) * 32 + c.j] = v[r];
                }
            }
            if (valid) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    f32x4 o = {v[4 * rq], v[4 * rq + 1], v[4 * rq + 2], v[4 * rq + 3]};
                    *(f32x4*)(drow + 9 * 256 + 32 * b + 8 * rq + 4 * c.h) = o;
                }
            }
        }
        __syncthreads();
        // raw d(final) = W_dir[:, :256]^T slot 9: 64 K-steps over the 128 hidden units
        f32x16 acc[2];
        n_zero2(acc);
        n_hpart2<64>(c, wt_, T.off_dir * 4 + z0, cur, acc);
        const float* ws_ = pk + L.off_wsig;
#pragma unroll 1
        for (int g = 8; g >= 1; --g) {
            if (g == 8) n_bwd_slot<0>(c, acc, nullptr, ws_, dsig, nxt, drow + 8 * 256, valid);
            else if (g == 7) n_bwd_slot<2>(c, acc, arow + 7 * 256, ws_, dsig, nxt, drow + 7 * 256, valid);
            else n_bwd_slot<1>(c, acc, arow + g * 256, ws_, dsig, nxt, drow + g * 256, valid);
            __syncthreads();
            float* t = cur; cur = nxt; nxt = t;
            n_zero2(acc);
            n_hpart2<128>(c, wt_, T.off_h[g] * 4 + z0, cur, acc);
        }
        // slot 0 = [h1 > 0] d_h1
        n_bwd_slot<1>(c, acc, arow, ws_, dsig, nullptr, drow, valid);
        __syncthreads();        // the images are rewritten by the next tile
    }
}

extern "C" int nf_nerf_mlp_bwd_n(const float* packed, const float* packed_t, int cx, int cd, const float* acts,
                                 const int32_t* n_rows, int max_rows, const int32_t* row_sample, const float* rgbsigma,
                                 const float* d_rgbsigma, float* dpre, nf_stream_t stream)
{
    NF_CHECK_ARG(packed && packed_t && acts && n_rows && row_sample && rgbsigma && d_rgbsigma && dpre, "null pointer");
    NF_CHECK_ARG(cx >= 1 && cx <= 256 && cd >= 1 && cd <= 256, "bad channel counts");
    if (max_rows <= 0) return NF_OK;
    NfMlpLayout L = mlp_layout(cx, cd);
    NfMlpLayoutT T = mlp_layout_t();
    const int tiles = (max_rows + 31) / 32;
    const size_t lds = (size_t)(2 * NN_ACT) * sizeof(float);       // 64 KB: two workgroups per CU
    static bool attr_set[64] = {};
    if (nf_first_use_on_device(attr_set))
        hipFuncSetAttribute((const void*)k_mlp_bwd_n, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k_mlp_bwd_n, dim3(tiles), dim3(256), lds, (hipStream_t)stream, L, T, packed, packed_t, acts, n_rows, max_rows,
                       row_sample, (const float4*)rgbsigma, (const float4*)d_rgbsigma, dpre);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_nerf_mlp_fwd_n(const float* packed, int cx, int cd, const float* X, const int32_t* n_rows, int max_rows,
                                 const int32_t* row_sample, float* rgbsigma, float* acts, nf_stream_t stream)
{
    NF_CHECK_ARG(packed && X && n_rows && row_sample && rgbsigma, "null pointer");
    NF_CHECK_ARG(cx >= 1 && cx <= 256 && cd >= 1 && cd <= 256, "bad channel counts");
    if (max_rows <= 0) return NF_OK;
    NfMlpLayout L = mlp_layout(cx, cd);
    NF_CHECK_ARG(L.qx == 25 && L.qd == 7, "built for the default 198 + 54 feature row (other encodings: nf_nerf_mlp_fwd)");
    const int tiles = (max_rows + 31) / 32;
    const size_t lds = (size_t)(2 * NN_ACT) * sizeof(float);       // 64 KB: two workgroups per CU
    static bool attr_set[64] = {};
    if (nf_first_use_on_device(attr_set)) {
        hipFuncSetAttribute((const void*)k_mlp_fwd_n<true, 25, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)k_mlp_fwd_n<false, 25, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    hipStream_t st = (hipStream_t)stream;
    if (acts)
        hipLaunchKernelGGL((k_mlp_fwd_n<true, 25, 7>), dim3(tiles), dim3(256), lds, st, L, packed, X, n_rows, max_rows, row_sample,
                           (float4*)rgbsigma, acts);
    else
        hipLaunchKernelGGL((k_mlp_fwd_n<false, 25, 7>), dim3(tiles), dim3(256), lds, st, L, packed, X, n_rows, max_rows, row_sample,
                           (float4*)rgbsigma, acts);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

