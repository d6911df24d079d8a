// nf_mlp_h2.hip — fp16-MFMA NeRF MLP forward, version 3: TWO 32-sample tiles per wave, out-block-major.
//
// What round 1's kernel (nf_mlp_h.hip, K-major, one tile per wave) paid per v_mfma_f32_32x32x16_f16:
// one ds_read_b128 (A operand), ~2 VALU for the B operand (8 accvgpr_read + 4 cvt_pk + 4 pk_max per 8 MFMAs ... per
// K-step, but the K-step's B feeds only 8 MFMAs of ONE tile), and a quarter of a ring refill; rocprofv3 counted
// 4.0 VALU per MFMA and MFMA busy 0.46 of the wave cycles.  Here:
//   * a wave owns 64 samples = two tiles.  Every A operand (32 output features x 16 inputs, 1 KB, from the LDS weight
//     ring) feeds TWO MFMAs, one per tile: half the LDS reads, half the ring traffic and half the barriers per MFMA;
//   * the loop nest is out-block-major: for each block of 32 output features, all K-steps run back to back into ONE
//     accumulator per tile.  The activations of a layer are therefore complete block by block, and each finished
//     block is rounded to PACKED fp16 (v_cvt_pk_f16_f32 + v_pk_max_f16 for the ReLU) exactly once, in the shadow of the
//     next block's MFMAs — one conversion per PRODUCED register instead of one per consumed one.  The packed
//     activations of the previous layer (64 registers per tile) are the B operands of all 8 blocks of the next layer;
//   * live registers: 2 banks x 2 tiles x 64 packed activations + 2 x 2 accumulators (64) + operand prefetch: one
//     wave per SIMD, ~400 of the 512 unified registers;
//   * sigma and rgb heads ride the matrix pipe as 1-block "layers" (rows 0 / 0..2 of a 32-row block) instead of a
//     VALU dot product over accumulators that no longer exist in fp32.
// Weight stream: a flat sequence of 1 KB A blocks in consumption order, shared by the 4 waves of a workgroup through
// a 48 KB LDS ring (3 chunks of 16 blocks; rendezvous 4 blocks before a chunk ends, so A operands are prefetched 3
// steps ahead across chunk boundaries).  The X operands of layer 1 are loaded once from HBM (fp16 layout written by
// nf_render_features(x_fp16 = 1)), kept in registers for layer 1 and parked in an LDS stash (26 KB per wave) for the
// skip layer.  Biases: one K-step with a hi/lo fp16 split, as in nf_mlp_h.hip.  fp32 accumulate throughout.
#include "nf_mlp_layout.h"
#include <math.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

#define H2_CHUNK 16          // A blocks (1 KB each) per ring chunk
#define H2_RING 48           // ring = 3 chunks
#ifndef H2_PF
#define H2_PF 3              // A-operand prefetch distance (steps)
#endif
#ifndef H2_BND
#define H2_BND 12            // the chunk rendezvous runs at the start of step (chunk * 16 + 12)
#endif
#define H2_XS 13             // X K-steps of the position-like part (200 padded features -> 13 x 16)

enum { H2K_BIAS = 0, H2K_X = 1, H2K_H = 2 };
struct H2Desc { int layer, blk, kind, idx; };   // layer 0..8 trunk (blk 8 of layer 8 = sigma head), 9 = view branch, 10 = rgb head

__host__ __device__ constexpr int h2_layer_blocks(int l) { return l < 8 ? 8 : (l == 8 ? 9 : (l == 9 ? 4 : 1)); }
__host__ __device__ constexpr int h2_block_steps(int l) { return l == 0 ? 14 : (l == 4 ? 30 : (l <= 8 ? 17 : (l == 9 ? 21 : 9))); }

__host__ __device__ constexpr int h2_total_steps()
{
    int n = 0;
    for (int l = 0; l <= 10; ++l) n += h2_layer_blocks(l) * h2_block_steps(l);
    return n;
}

__host__ __device__ constexpr int h2_padded_steps() { return (h2_total_steps() + H2_RING - 1) / H2_RING * H2_RING; }

__host__ __device__ inline H2Desc h2_desc(int i)
{
    H2Desc d;
    int l = 0;
    for (; l <= 10; ++l) {
        int n = h2_layer_blocks(l) * h2_block_steps(l);
        if (i < n) break;
        i -= n;
    }
    d.layer = l;
    d.blk = i / h2_block_steps(l);
    int s = i % h2_block_steps(l);
    if (s == 0) { d.kind = H2K_BIAS; d.idx = 0; }
    else if (l == 0) { d.kind = H2K_X; d.idx = s - 1; }
    else if (l == 4) { if (s <= H2_XS) { d.kind = H2K_X; d.idx = s - 1; } else { d.kind = H2K_H; d.idx = s - 1 - H2_XS; } }
    else if (l == 9) { if (s <= 4) { d.kind = H2K_X; d.idx = 12 + (s - 1); } else { d.kind = H2K_H; d.idx = s - 5; } }
    else { d.kind = H2K_H; d.idx = s - 1; }
    return d;
}

extern "C" size_t nf_nerf_packed_h2_bytes(void) { return (size_t)h2_padded_steps() * 1024; }

// one workgroup per A block of the stream
__global__ void k_mlp_pack_h2(int cx, int cd, NfNerfPtrs P, _Float16* __restrict__ out)
{
    const H2Desc st = h2_desc(blockIdx.x);
    _Float16* dst = out + (size_t)blockIdx.x * 512;
    const int L = st.layer;
    const bool sigma_blk = (L == 8 && st.blk == 8), rgb_blk = (L == 10);
    for (int t = threadIdx.x; t < 512; t += blockDim.x) {
        const int lane = t >> 3, e = t & 7, h = lane >> 5, jj = lane & 31;
        const int o = 32 * st.blk + jj;
        // weight row of this lane: (matrix, row, row length) or none
        const float* wrow = nullptr;
        float bias = 0.f;
        int in_dim = 0, hcol0 = 0;          // hcol0: column of hidden feature 0 in the row
        if (sigma_blk) { if (jj == 0) { wrow = P.w[10]; bias = P.b[10][0]; } in_dim = 256; }
        else if (rgb_blk) { if (jj < 3) { wrow = P.w[11] + (size_t)jj * 128; bias = P.b[11][jj]; } in_dim = 128; }
        else {
            in_dim = (L == 0) ? cx : (L == 4 ? cx + 256 : (L == 9 ? 256 + cd : 256));
            wrow = P.w[L] + (size_t)o * in_dim;
            bias = P.b[L][o];
            hcol0 = (L == 4) ? cx : 0;
        }
        float v = 0.f;
        if (wrow) {
            if (st.kind == H2K_BIAS) {
                if (h == 0 && e < 2) { _Float16 hi = (_Float16)bias; v = (e == 0) ? (float)hi : (bias - (float)hi); }
            } else if (st.kind == H2K_X) {
                // X operand layout of nf_render_features(x_fp16 = 1): K-step t, lane half h, element e -> padded feature
                const int f = 16 * st.idx + (e < 4 ? 4 * h + e : 8 + 4 * h + (e - 4));
                const int qx8 = ((cx + 7) / 8) * 8;
                if (L == 9) { const int fd = f - qx8; if (fd >= 0 && fd < cd) v = wrow[256 + fd]; }
                else if (f < cx) v = wrow[f];
            } else {
                const int f = frag_feature(st.idx >> 1, 8 * (st.idx & 1) + e, h);
                v = wrow[hcol0 + f];
            }
        }
        dst[t] = (_Float16)v;
    }
}

__global__ void k_zero_u4_h2(u32x4* p, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = u32x4{0u, 0u, 0u, 0u};
}

extern "C" int nf_nerf_pack_h2(const nf_nerf_params_t* params, int cx, int cd, void* stream_h2, nf_stream_t stream)
{
    NF_CHECK_ARG(params && stream_h2, "null pointer");
    NF_CHECK_ARG((cx + 7) / 8 == 25 && (cd + 7) / 8 == 7, "the fp16 path is built for the default 198+54 feature row");
    NfNerfPtrs P;
    for (int i = 0; i < 12; ++i) { P.w[i] = params->w[i]; P.b[i] = params->b[i]; }
    hipStream_t st = (hipStream_t)stream;
    const size_t nu4 = (size_t)h2_padded_steps() * 64;
    hipLaunchKernelGGL(k_zero_u4_h2, dim3((unsigned)((nu4 + 255) / 256)), dim3(256), 0, st, (u32x4*)stream_h2, nu4);
    hipLaunchKernelGGL(k_mlp_pack_h2, dim3(h2_total_steps()), dim3(256), 0, st, cx, cd, P, (_Float16*)stream_h2);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// ------------------------------------------------------------------------------------------------
// kernel
// ------------------------------------------------------------------------------------------------
struct H2Ctx {
    const u32x4* stream;       // global weight stream
    u32x4* ring;               // LDS base
    int nchunks, chunk_next;   // chunk_next: the chunk the NEXT global fetch brings in (runtime, cyclic)
    int lane, wave;
    u32x4 stage[4];            // this wave's quarter of the chunk in flight
    u32x4 ab[4];               // rotating A operands: ab[s & 3] belongs to step s
    u32x4 bias_b, bias_b2;     // B operand of the bias K-step (ones at k = 0, 1), twice (see h2_block)
};

__device__ __forceinline__ void h2_fetch(H2Ctx& c)
{
    const u32x4* src = c.stream + (size_t)c.chunk_next * (H2_CHUNK * 64) + c.wave * 256 + c.lane;
#pragma unroll
    for (int i = 0; i < 4; ++i) c.stage[i] = src[i * 64];
    c.chunk_next = (c.chunk_next + 1 == c.nchunks) ? 0 : c.chunk_next + 1;
}

__device__ __forceinline__ void h2_publish(H2Ctx& c, int third)
{
    u32x4* dst = c.ring + third * (H2_CHUNK * 64) + c.wave * 256 + c.lane;
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i * 64] = c.stage[i];
}

// Rendezvous of chunk k, at the start of step 16 k + 12: every wave is past chunk k - 1, whose third receives chunk
// k + 2; chunk k + 1 (written one rendezvous ago) becomes visible, 4 steps before its first block is consumed.
__device__ __forceinline__ void h2_boundary(H2Ctx& c, int slot)
{
    __syncthreads();
    h2_publish(c, (slot / H2_CHUNK + 2) % 3);
    h2_fetch(c);
}

// One step: the A block of this step times the B operands of the two tiles.  NDS = further ds_read in the region
// (stash prefetch), NVALU = VALU instructions of the conversion piece the caller emitted for this region.
template <int NDS, int NVALU>
__device__ __forceinline__ void h2_step(H2Ctx& c, int& slot, const u32x4 bA, const u32x4 bB, f32x16& aA, f32x16& aB, bool zero)
{
    const bool boundary = (slot % H2_CHUNK) == H2_BND;
    if (boundary) h2_boundary(c, slot);
    c.ab[(slot + H2_PF) & 3] = c.ring[((slot + H2_PF) % H2_RING) * 64 + c.lane];
    const h8 av = __builtin_bit_cast(h8, c.ab[slot & 3]);
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    aA = MFMA16(av, __builtin_bit_cast(h8, bA), zero ? z : aA);
    aB = MFMA16(av, __builtin_bit_cast(h8, bB), zero ? z : aB);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1 + NDS, 0);
    if (boundary) {
        __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
    }
    if (NVALU) __builtin_amdgcn_sched_group_barrier(0x002, (NVALU + 1) / 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if (boundary) {
        __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
    }
    if (NVALU) __builtin_amdgcn_sched_group_barrier(0x002, NVALU / 2, 0);
    asm volatile("" : "+v"(aA));      // MFMAs are pure: pin them to their step (VGPR form: see the build flags)
    asm volatile("" : "+v"(aB));
    slot++;
    __builtin_amdgcn_sched_barrier(0);
}

// padding step of the stream (no arithmetic): ring bookkeeping only
__device__ __forceinline__ void h2_skip(H2Ctx& c, int& slot)
{
    if ((slot % H2_CHUNK) == H2_BND) h2_boundary(c, slot);
    c.ab[(slot + H2_PF) & 3] = c.ring[((slot + H2_PF) % H2_RING) * 64 + c.lane];
    slot++;
    __builtin_amdgcn_sched_barrier(0);
}

template <bool RELU>
__device__ __forceinline__ unsigned h2_pk(float a, float b)
{
    h2v p = {(_Float16)a, (_Float16)b};
    if (RELU) {          // ReLU after rounding == rounding after ReLU; one packed max instead of two fp32 ones
        const h2v zz = {(_Float16)0.f, (_Float16)0.f};
        p = __builtin_elementwise_max(p, zz);
    }
    return __builtin_bit_cast(unsigned, p);
}

// conversion piece p (0..7) of a finished block: tile p >> 2, quarter q = p & 3: accumulator floats 4q .. 4q+3 ->
// packed registers 2q, 2q+1 of the block = elements of K-steps 2 blk (q < 2) / 2 blk + 1 (q >= 2) of the output bank
template <int CVT>   // 1: ReLU, 2: identity
__device__ __forceinline__ void h2_cvt_piece(int p, const f32x16 (&prev)[2], u32x4 (&outA)[16], u32x4 (&outB)[16], int blk)
{
    const int tile = p >> 2, q = p & 3;
    const f32x16& a = prev[tile];
#ifdef H2_AB_NOCVT          /* ceiling probe (garbage results): the finished block's registers taken as they are, no conversion VALU */
    const unsigned r0 = __float_as_uint(a[4 * q]), r1 = __float_as_uint(a[4 * q + 2]);
#else
    const unsigned r0 = h2_pk<CVT == 1>(a[4 * q], a[4 * q + 1]), r1 = h2_pk<CVT == 1>(a[4 * q + 2], a[4 * q + 3]);
#endif
    u32x4& dst = tile ? outB[2 * blk + (q >> 1)] : outA[2 * blk + (q >> 1)];
    dst[2 * (q & 1)] = r0;
    dst[2 * (q & 1) + 1] = r1;
}

__device__ __forceinline__ u32x4 h2_bias_b(int h)
{
    const _Float16 one = (_Float16)(h == 0 ? 1.f : 0.f), zz = (_Float16)0.f;
    const h8 r = {one, one, zz, zz, zz, zz, zz, zz};
    return __builtin_bit_cast(u32x4, r);
}

// One out-block for both tiles.
//   XMODE 0: no X part; 1: X operands in registers (xA/xB[0..NX-1]); 2: X operands from the LDS stash
//   NH: hidden K-steps read from the input bank (inA/inB)
//   CVT: conversion of the PREVIOUS block's accumulators (prev) into the output bank at block index pblk, interleaved
//        with this block's steps 1..8 (0: nothing pending)
template <int XMODE, int NX, int NH, int CVT>
__device__ __forceinline__ void h2_block(H2Ctx& c, int& slot, f32x16 (&acc)[2], const u32x4 (&inA)[16], const u32x4 (&inB)[16],
                                         const u32x4* xA, const u32x4* xB, const u32x4* __restrict__ stash,
                                         const f32x16 (&prev)[2], u32x4 (&outA)[16], u32x4 (&outB)[16], int pblk)
{
    // the two bias MFMAs have identical operands; c.bias_b2 is an opaque copy of the B operand, so that they are not
    // merged into one MFMA + a 16-register copy (8 v_mov_b64 cost more issue slots than the MFMA)
    const u32x4 bb = c.bias_b, bb2 = c.bias_b2;
    u32x4 xr[2][4];
    if (XMODE == 2) {
#pragma unroll
        for (int t = 0; t < 2; ++t) { xr[0][t] = stash[t * 64]; xr[1][t] = stash[(H2_XS + t) * 64]; }
        h2_step<4, 0>(c, slot, bb, bb2, acc[0], acc[1], true);
    } else {
        h2_step<0, 0>(c, slot, bb, bb2, acc[0], acc[1], true);
    }
    // The eight conversion pieces of the previous block ride in this block's first 8 / H2_PPS steps, H2_PPS pieces (4 VALU each) per
    // step (round 5: a VALU burst between two MFMAs costs the matrix pipe about the same whether it holds two instructions or four —
    // measured on the fp32 kernel, gen_mlp_a.py RELU_BATCH — so fewer, larger bursts)
#ifndef H2_PPS
#define H2_PPS 1
#endif
    constexpr int CS = 8 / H2_PPS, NV = 4 * H2_PPS;
    int s = 1;      // step inside the block (static)
#pragma unroll
    for (int t = 0; t < NX; ++t, ++s) {
        if (CVT && s <= CS) {
#pragma unroll
            for (int u = 0; u < H2_PPS; ++u) h2_cvt_piece<CVT>((s - 1) * H2_PPS + u, prev, outA, outB, pblk);
        }
        if (XMODE == 2) {
            if (t + 2 < NX) { xr[0][(t + 2) & 3] = stash[(t + 2) * 64]; xr[1][(t + 2) & 3] = stash[(H2_XS + t + 2) * 64]; }
            if (CVT && s <= CS) { if (t + 2 < NX) h2_step<2, NV>(c, slot, xr[0][t & 3], xr[1][t & 3], acc[0], acc[1], false);
                                  else h2_step<0, NV>(c, slot, xr[0][t & 3], xr[1][t & 3], acc[0], acc[1], false); }
            else { if (t + 2 < NX) h2_step<2, 0>(c, slot, xr[0][t & 3], xr[1][t & 3], acc[0], acc[1], false);
                   else h2_step<0, 0>(c, slot, xr[0][t & 3], xr[1][t & 3], acc[0], acc[1], false); }
        } else {
            if (CVT && s <= CS) h2_step<0, NV>(c, slot, xA[t], xB[t], acc[0], acc[1], false);
            else h2_step<0, 0>(c, slot, xA[t], xB[t], acc[0], acc[1], false);
        }
    }
#pragma unroll
    for (int k = 0; k < NH; ++k, ++s) {
        if (CVT && s <= CS) {
#pragma unroll
            for (int u = 0; u < H2_PPS; ++u) h2_cvt_piece<CVT>((s - 1) * H2_PPS + u, prev, outA, outB, pblk);
            h2_step<0, NV>(c, slot, inA[k], inB[k], acc[0], acc[1], false);
        } else {
            h2_step<0, 0>(c, slot, inA[k], inB[k], acc[0], acc[1], false);
        }
    }
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) k_mlp_fwd_h2(const u32x4* __restrict__ stream_h, int nslots,
                                                    const u32x4* __restrict__ Xh, const int* __restrict__ n_rows, int max_rows,
                                                    const int* __restrict__ row_sample, float4* __restrict__ rgbsigma)
{
    extern __shared__ u32x4 lds[];    // [ring 48 KB][stash 4 waves x 2 tiles x 13 steps x 1 KB]
    const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31, wave = threadIdx.x >> 6;
    const int nrows = min(*n_rows, max_rows);
    const int ntiles = (nrows + 31) >> 5;
    const int npairs = (ntiles + 1) >> 1;
    const int ngroups = (npairs + 3) >> 2;
    u32x4* stash = lds + H2_RING * 64 + wave * (2 * H2_XS * 64) + lane;
    H2Ctx c;
    c.stream = stream_h; c.ring = lds;
    c.nchunks = nslots / H2_CHUNK; c.chunk_next = 0; c.lane = lane; c.wave = wave;
    c.bias_b = h2_bias_b(h); c.bias_b2 = c.bias_b;
    asm volatile("" : "+v"(c.bias_b2));
    // prologue: chunks 0 and 1 into thirds 0 and 1, chunk 2 in flight, A operands of steps 0..2
    h2_fetch(c); h2_publish(c, 0);
    h2_fetch(c); h2_publish(c, 1);
    h2_fetch(c);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < H2_PF; ++s) c.ab[s] = c.ring[s * 64 + lane];

    // X operands of layer 1 / the skip layer live in this wave's LDS stash (2 tiles x 13 K-steps x 1 KB).  The first
    // pair's are staged here; every further pair's are staged by the PREVIOUS pair's pass, in 13 batches of 2 KB spread
    // over the blocks of layers 5..8 (the stash is free once the skip layer has run): global -> 8 registers in one
    // block, ds_write in the next, so neither the HBM latency nor the stores ever sit in front of an MFMA.
    const int last_pair = npairs > 0 ? npairs - 1 : 0;
    if (ngroups > (int)blockIdx.x) {
        const u32x4* pa_ = Xh + (size_t)(2 * min((int)blockIdx.x * 4 + wave, last_pair)) * 16 * 64 + lane;
#pragma unroll
        for (int t = 0; t < H2_XS; ++t) {
            const u32x4 va = __builtin_nontemporal_load(pa_ + t * 64), vb = __builtin_nontemporal_load(pa_ + (16 + t) * 64);
            stash[t * 64] = va;
            stash[(H2_XS + t) * 64] = vb;
        }
    }

    for (int tg = blockIdx.x; tg < ngroups; tg += gridDim.x) {
        int pair = tg * 4 + wave;
        const bool owner = pair < npairs;
        if (!owner) pair = npairs - 1;            // idle waves recompute the last pair (keeps the barriers matched)
        const u32x4* xtA = Xh + (size_t)(2 * pair) * 16 * 64 + lane;        // Xh[tile][16 K-steps][64 lanes]
        const u32x4* xtB = xtA + 16 * 64;
        // next pair of this wave (clamped: the loads of the last pass re-read valid rows and are simply not used)
        const u32x4* xnA = Xh + (size_t)(2 * min((tg + (int)gridDim.x) * 4 + wave, last_pair)) * 16 * 64 + lane;
        int slot = 0;                             // static step counter: every pair consumes exactly nslots (a multiple of the ring)

        u32x4 bank[2][2][16];                     // [which][tile][K-step]: packed fp16 activations
        f32x16 acc[2][2];                         // [buffer][tile]
        u32x4 xf0, xf1;                           // X batch in flight (next pair)

        // Every block call is written out with literal block indices (macros, no loops): the running step counter slot
        // must fold to a constant at every step (ring slot, rendezvous and prefetch decisions are all static), and the
        // loop unroller gives up on bodies of this size before that folding has happened.
#define H2_REP7(M) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
        // ---- layer 0: X (stash) -> bank[1]
        h2_block<2, H2_XS, 0, 0>(c, slot, acc[0], bank[0][0], bank[0][1], nullptr, nullptr, stash, acc[1], bank[1][0], bank[1][1], 0);
#define H2_L0(b) h2_block<2, H2_XS, 0, 1>(c, slot, acc[(b) & 1], bank[0][0], bank[0][1], nullptr, nullptr, stash, acc[((b) - 1) & 1], bank[1][0], bank[1][1], (b) - 1);
        H2_REP7(H2_L0)
        // ---- hidden layers: block 0 of a layer converts block 7 of the previous one (into its own input bank).
        // XFB >= 0: global block index of the layer's block 0 in the X staging schedule (even block: load batch, odd: store)
#define H2_XFER(gb)                                                                                                              \
        if ((gb) >= 0 && (gb) / 2 < H2_XS) {                                                                                     \
            if (((gb) & 1) == 0) { xf0 = __builtin_nontemporal_load(xnA + ((gb) / 2) * 64);                                      \
                                   xf1 = __builtin_nontemporal_load(xnA + (16 + (gb) / 2) * 64); }                               \
            else { stash[((gb) / 2) * 64] = xf0; stash[(H2_XS + (gb) / 2) * 64] = xf1; }                                         \
        }
#define H2_HB(IN, OUT, CV, b, XFB) H2_XFER((XFB) < 0 ? -1 : (XFB) + (b))                                                         \
        h2_block<0, 0, 16, CV>(c, slot, acc[(b) & 1], bank[IN][0], bank[IN][1], nullptr, nullptr, nullptr, acc[((b) - 1) & 1], bank[OUT][0], bank[OUT][1], (b) - 1);
#define H2_HIDDEN_LAYER(IN, OUT, CV, XFB)                                                                                        \
        H2_XFER(XFB)                                                                                                             \
        h2_block<0, 0, 16, 1>(c, slot, acc[0], bank[IN][0], bank[IN][1], nullptr, nullptr, nullptr, acc[1], bank[IN][0], bank[IN][1], 7); \
        H2_HB(IN, OUT, CV, 1, XFB) H2_HB(IN, OUT, CV, 2, XFB) H2_HB(IN, OUT, CV, 3, XFB) H2_HB(IN, OUT, CV, 4, XFB)               \
        H2_HB(IN, OUT, CV, 5, XFB) H2_HB(IN, OUT, CV, 6, XFB) H2_HB(IN, OUT, CV, 7, XFB)
        H2_HIDDEN_LAYER(1, 0, 1, -1)      // layer 1: bank[1] -> bank[0]
        H2_HIDDEN_LAYER(0, 1, 1, -1)      // layer 2
        H2_HIDDEN_LAYER(1, 0, 1, -1)      // layer 3
        // layer 4 (skip): X from the stash + bank[0] -> bank[1]
        h2_block<2, H2_XS, 16, 1>(c, slot, acc[0], bank[0][0], bank[0][1], nullptr, nullptr, stash, acc[1], bank[0][0], bank[0][1], 7);
#define H2_L4(b) h2_block<2, H2_XS, 16, 1>(c, slot, acc[(b) & 1], bank[0][0], bank[0][1], nullptr, nullptr, stash, acc[((b) - 1) & 1], bank[1][0], bank[1][1], (b) - 1);
        H2_REP7(H2_L4)
        H2_HIDDEN_LAYER(1, 0, 1, 0)       // layer 5  (+ X batches 0..3 of the next pair)
        H2_HIDDEN_LAYER(0, 1, 1, 8)       // layer 6  (+ batches 4..7)
        H2_HIDDEN_LAYER(1, 0, 1, 16)      // layer 7: -> bank[0] = relu(h8), the input of xyz_encoding_final AND of sigma (+ 8..11)
        H2_HIDDEN_LAYER(0, 1, 2, 24)      // layer 8 (xyz_encoding_final, no activation) -> bank[1]  (+ batch 12)
        // sigma block on the same input (bank[0]); converts block 7 of the final layer on the way
        h2_block<0, 0, 16, 2>(c, slot, acc[0], bank[0][0], bank[0][1], nullptr, nullptr, nullptr, acc[1], bank[1][0], bank[1][1], 7);
        // ---- view branch: [final (bank[1], identity) | dir features] -> 128 hidden (ReLU) -> bank[0] K-steps 0..7
        u32x4 dA[4], dB[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { dA[t] = __builtin_nontemporal_load(xtA + (12 + t) * 64); dB[t] = __builtin_nontemporal_load(xtB + (12 + t) * 64); }
        // view block 0 runs in acc[1]; sigma (acc[0], register 0 of the lanes h == 0) is read before acc[0] is reused
        h2_block<1, 4, 16, 0>(c, slot, acc[1], bank[1][0], bank[1][1], dA, dB, nullptr, acc[0], bank[0][0], bank[0][1], 0);
        const float sigA = acc[0][0][0], sigB = acc[0][1][0];
#define H2_VB(b) h2_block<1, 4, 16, 1>(c, slot, acc[((b) + 1) & 1], bank[1][0], bank[1][1], dA, dB, nullptr, acc[(b) & 1], bank[0][0], bank[0][1], (b) - 1);
        H2_VB(1) H2_VB(2) H2_VB(3)
        // ---- rgb head: 8 hidden K-steps over relu(view hidden); converts view block 3 on the way
        h2_block<0, 0, 8, 1>(c, slot, acc[1], bank[0][0], bank[0][1], nullptr, nullptr, nullptr, acc[0], bank[0][0], bank[0][1], 3);
        // the stream is padded to a multiple of the ring: walk the padding steps (uniform)
        constexpr int H2_PAD = h2_padded_steps() - h2_total_steps();
#pragma unroll
        for (int i = 0; i < H2_PAD; ++i) h2_skip(c, slot);

        if (h == 0 && owner) {
            const f32x16 rA = acc[1][0], rB = acc[1][1];
            const int rowA = pair * 64 + j, rowB = rowA + 32;
            if (rowA < nrows) {
                float4 o;
                o.x = 1.f / (1.f + expf(-rA[0])); o.y = 1.f / (1.f + expf(-rA[1])); o.z = 1.f / (1.f + expf(-rA[2])); o.w = sigA;
                rgbsigma[row_sample[rowA]] = o;
            }
            if (rowB < nrows) {
                float4 o;
                o.x = 1.f / (1.f + expf(-rB[0])); o.y = 1.f / (1.f + expf(-rB[1])); o.z = 1.f / (1.f + expf(-rB[2])); o.w = sigB;
                rgbsigma[row_sample[rowB]] = o;
            }
        }
    }
}

extern "C" int nf_nerf_mlp_fwd_h2(const void* stream_h2, int cx, int cd, const void* X, const int32_t* n_rows, int max_rows,
                                  const int32_t* row_sample, float* rgbsigma, nf_stream_t stream)
{
    NF_CHECK_ARG(stream_h2 && X && n_rows && row_sample && rgbsigma, "null pointer");
    if (max_rows <= 0) return NF_OK;
    NF_CHECK_ARG((cx + 7) / 8 == 25 && (cd + 7) / 8 == 7, "the fp16 path is built for the default 198+54 feature row");
    const int nslots = h2_padded_steps();
    const int pairs = ((max_rows + 31) / 32 + 1) / 2;
    int blocks = (pairs + 3) / 4;
    if (blocks > 256) blocks = 256;
    const size_t lds = (size_t)H2_RING * 1024 + (size_t)4 * 2 * H2_XS * 1024;
    static bool attr_set[64] = {};
    if (nf_first_use_on_device(attr_set)) {
        hipFuncSetAttribute((const void*)k_mlp_fwd_h2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL(k_mlp_fwd_h2, dim3(blocks), dim3(256), lds, (hipStream_t)stream, (const u32x4*)stream_h2, nslots,
                       (const u32x4*)X, n_rows, max_rows, row_sample, (float4*)rgbsigma);
    NF_CHECK_LAUNCH();
    return NF_OK;
}
