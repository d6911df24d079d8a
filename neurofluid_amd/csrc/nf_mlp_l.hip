// nf_mlp_l.hip — fp32-MFMA NeRF MLP forward (inference) with the weight stream shared through an LDS ring.
//
// Same arithmetic, orientation and register-resident chaining as k_mlp_fwd (nf_mlp.hip): D[out_feature][sample] on
// v_mfma_f32_32x32x2_f32, the D fragment of a layer is the B operand of the next one, ReLU on the fly.  What changes is
// where the A operands (weights) come from.  In k_mlp_fwd every wave streams the whole 2.7 MB weight set from the L2
// itself: one global_load_dwordx4 per 4 MFMAs, and a VMEM issue costs the single wave of a SIMD ~55 cycles of its issue
// slot (measured: a bare 8-MFMA loop sustains 155 TFLOP/s, 136 with that load mix).  Here the 4 waves of a workgroup
// walk the stream together: each wave fetches a QUARTER of every 16 KB chunk (global -> VGPR -> ds_write_b128), all
// four read the operands back with ds_read_b128 — 4x fewer VMEM issues per MFMA, the rest on the cheaper LDS port.
//
// Stream = flat sequence of 2 KB slots (one K-step x 8 output blocks x 64 lanes x 4 B x ... = [2][64 lanes] x 16 B;
// the 4-block view branch packs two K-steps per slot), grouped in chunks of 8 slots, every part of a layer padded to
// whole chunks so that slot positions inside a part are compile-time constants:
//     layer 0          : X part (100 K-steps + 4 pad)                      + bias slot + 7 pad
//     layers 1-3, 5-8  :                           H part (128 K-steps)     + bias slot + 7 pad
//     layer 4 (skip)   : X part (100 + 4 pad)    + H part (128)             + bias slot + 7 pad
//     view branch      : X part (28 half-steps = 14 slots + 2 pad) + H part (128 half-steps = 64 slots) + bias + 7 pad
// The bias enters as the LAST K-step of a layer (A = bias, B = 1), the first K-step of a layer starts from C = 0.
//
// Ring protocol (3 chunks of 16 KB = 48 KB LDS, next to the 100 KB X stash): during the first 4 slots of chunk k every
// wave fetches its quarter of chunk k+2, one global load per K-step; the boundary of chunk k runs at the START of its
// LAST slot: s_barrier (everybody is done with chunk k-1, chunk k+1 is visible), publish the staged chunk k+2 into the
// third that chunk k-1 occupied.  The A operands of every slot are prefetched one K-step ahead, across chunk
// boundaries too, so the rendezvous is never followed by a dependent LDS read.
//
// Built for the default encodings (198 + 54 features: 25 + 7 feature groups); other configurations use k_mlp_fwd.
#include "nf_mlp_layout.h"
#include <math.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

#define LS_F4 128          // f32x4 per slot (2 KB)
#define LS_CHUNK 8         // slots per chunk
#define LS_CHUNK_F4 (LS_CHUNK * LS_F4)
#define LS_STAGE (LS_CHUNK_F4 / 4 / 64)      // f32x4 per lane per chunk per wave = 4
#define LS_QX 25
#define LS_QD 7

// ------------------------------------------------------------------------------------------------
// stream description: parts in consumption order
// ------------------------------------------------------------------------------------------------
struct LPart { int src; int units; int unit_floats; int slots; };   // src: float offset in the packed blob; slots incl. padding
#define LS_MAX_PARTS 32
struct LStream { int nparts, nslots; LPart p[LS_MAX_PARTS]; };

static LStream l_stream(const NfMlpLayout& L)
{
    LStream S;
    int n = 0, slots = 0;
    auto add = [&](int src, int units, int unit_floats) {
        int raw = unit_floats == 512 ? units : (units + 1) / 2;
        int padded = (raw + LS_CHUNK - 1) / LS_CHUNK * LS_CHUNK;
        S.p[n++] = LPart{src, units, unit_floats, padded};
        slots += padded;
    };
    for (int l = 0; l < 9; ++l) {
        if (L.off_x[l] >= 0) add(L.off_x[l], L.qx * 4, 512);
        if (L.off_h[l] >= 0) add(L.off_h[l], 128, 512);
        add(L.off_bstep[l], 1, 512);
    }
    add(L.off_dir_x, L.qd * 4, 256);
    add(L.off_dir_h, 128, 256);
    add(L.off_bstep_dir, 1, 256);
    S.nparts = n; S.nslots = slots;
    return S;
}

extern "C" size_t nf_nerf_stream_floats(int cx, int cd)
{
    NfMlpLayout L = mlp_layout(cx, cd);
    return (size_t)l_stream(L).nslots * 512;
}

// one block per slot: copy (or zero) 512 floats
__global__ void __launch_bounds__(128) k_mlp_stream_pack(LStream S, const float* __restrict__ packed, float* __restrict__ out)
{
    int slot = blockIdx.x, pi = 0, base = 0;
    while (pi < S.nparts - 1 && slot >= base + S.p[pi].slots) { base += S.p[pi].slots; ++pi; }
    const LPart P = S.p[pi];
    const int s = slot - base;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const int t = threadIdx.x;                      // 128 threads x 16 B = one slot
    if (P.unit_floats == 512) {
        if (s < P.units) v = *(const f32x4*)(packed + P.src + (size_t)s * 512 + t * 4);
    } else {                                        // two 1 KB half-steps per slot
        const int u = 2 * s + (t >> 6);
        if (u < P.units) v = *(const f32x4*)(packed + P.src + (size_t)u * 256 + (t & 63) * 4);
    }
    *(f32x4*)(out + (size_t)slot * 512 + t * 4) = v;
}

extern "C" int nf_nerf_pack_stream(const float* packed, int cx, int cd, float* stream_out, nf_stream_t stream)
{
    NF_CHECK_ARG(packed && stream_out, "null pointer");
    NfMlpLayout L = mlp_layout(cx, cd);
    NF_CHECK_ARG(L.qx == LS_QX && L.qd == LS_QD, "the LDS-ring path is built for the default 198+54 feature row");
    LStream S = l_stream(L);
    hipLaunchKernelGGL(k_mlp_stream_pack, dim3(S.nslots), dim3(128), 0, (hipStream_t)stream, S, packed, stream_out);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// ------------------------------------------------------------------------------------------------
// kernel
// ------------------------------------------------------------------------------------------------
struct LCtx {
    const f32x4* stream;       // global weight stream
    f32x4 *t0, *t1, *t2;       // LDS thirds: chunk being consumed, next chunk, the one the next boundary publishes into
    int nchunks, chunk_next;   // chunk_next: the chunk the NEXT global fetch brings in (cyclic over the stream)
    int lane, wave;
    int p;                     // slot position inside the current chunk (compile-time after unrolling)
    f32x4 stage[LS_STAGE];     // this wave's quarter of the chunk in flight
    f32x4 a0, a1, n0, n1;      // A operands of the current slot / of the next slot (always prefetched one slot ahead)
};

__device__ __forceinline__ void l_fetch(LCtx& c)
{
    const f32x4* src = c.stream + (size_t)c.chunk_next * LS_CHUNK_F4 + c.wave * (LS_STAGE * 64) + c.lane;
#pragma unroll
    for (int i = 0; i < LS_STAGE; ++i) c.stage[i] = src[i * 64];
    c.chunk_next = (c.chunk_next + 1 == c.nchunks) ? 0 : c.chunk_next + 1;
}

__device__ __forceinline__ void l_publish(LCtx& c, f32x4* third)
{
    f32x4* dst = third + c.wave * (LS_STAGE * 64) + c.lane;
#pragma unroll
    for (int i = 0; i < LS_STAGE; ++i) dst[i * 64] = c.stage[i];
}

// start of a slot: ring bookkeeping + prefetch of the next slot into n0/n1.  The rendezvous + publish sit on the last
// slot of a chunk; the refill of the staging registers is spread over the first LS_STAGE slots of the following
// chunk, ONE global load per K-step (back-to-back VMEM issues stall the single wave of a SIMD).
__device__ __forceinline__ void l_enter(LCtx& c)
{
    if (c.p == LS_CHUNK - 1) {
        __syncthreads();
        l_publish(c, c.t2);
    }
    if (c.p < LS_STAGE) {
        const f32x4* src = c.stream + (size_t)c.chunk_next * LS_CHUNK_F4 + c.wave * (LS_STAGE * 64) + c.lane;
        // (constant subscripts only: a subscript by c.p keeps the whole context struct in scratch)
        if (c.p == 0) c.stage[0] = src[0];
        else if (c.p == 1) c.stage[1] = src[64];
        else if (c.p == 2) c.stage[2] = src[128];
        else c.stage[3] = src[192];
        static_assert(LS_STAGE == 4, "one refill load per slot for the first LS_STAGE slots");
        if (c.p == LS_STAGE - 1) c.chunk_next = (c.chunk_next + 1 == c.nchunks) ? 0 : c.chunk_next + 1;
    }
    const f32x4* nx = (c.p == LS_CHUNK - 1) ? (c.t1 + c.lane) : (c.t0 + (c.p + 1) * LS_F4 + c.lane);
    c.n0 = nx[0]; c.n1 = nx[64];
}

// end of a slot: advance, rotating the thirds at a chunk end
__device__ __forceinline__ void l_leave(LCtx& c)
{
    c.a0 = c.n0; c.a1 = c.n1;
    if (c.p == LS_CHUNK - 1) {
        f32x4* t = c.t0; c.t0 = c.t1; c.t1 = c.t2; c.t2 = t;
        c.p = 0;
    } else {
        c.p++;
    }
}

// a padding slot
__device__ __forceinline__ void l_skip(LCtx& c)
{
    l_enter(c);
    l_leave(c);
}

template <bool ZERO>
__device__ __forceinline__ void l_mfma8(const f32x4 w0, const f32x4 w1, const float bv, f32x16 (&acc)[8])
{
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    acc[0] = MFMA32(w0[0], bv, ZERO ? z : acc[0]); acc[1] = MFMA32(w0[1], bv, ZERO ? z : acc[1]);
    acc[2] = MFMA32(w0[2], bv, ZERO ? z : acc[2]); acc[3] = MFMA32(w0[3], bv, ZERO ? z : acc[3]);
    acc[4] = MFMA32(w1[0], bv, ZERO ? z : acc[4]); acc[5] = MFMA32(w1[1], bv, ZERO ? z : acc[5]);
    acc[6] = MFMA32(w1[2], bv, ZERO ? z : acc[6]); acc[7] = MFMA32(w1[3], bv, ZERO ? z : acc[7]);
}

template <bool ZERO>
__device__ __forceinline__ void l_mfma4(const f32x4 w, const float bv, f32x16 (&acc)[4])
{
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    acc[0] = MFMA32(w[0], bv, ZERO ? z : acc[0]); acc[1] = MFMA32(w[1], bv, ZERO ? z : acc[1]);
    acc[2] = MFMA32(w[2], bv, ZERO ? z : acc[2]); acc[3] = MFMA32(w[3], bv, ZERO ? z : acc[3]);
}

// one K-step over 8 output blocks.  The LDS prefetch, the ring refill (boundary steps) and the caller's VALU for the
// next operand are spread over the MFMA shadows.
template <bool ZERO>
__device__ __forceinline__ void l_step8(LCtx& c, const float bv, f32x16 (&acc)[8])
{
    const bool boundary = (c.p == LS_CHUNK - 1), fetch = (c.p < LS_STAGE);
    l_enter(c);
    l_mfma8<ZERO>(c.a0, c.a1, bv, acc);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i < 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (boundary && i >= 2 && i < 6) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        if (fetch && i == 3) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" : "+a"(acc[i]));      // pin the MFMAs to this step (see nf_mlp_h.hip)
    l_leave(c);
    __builtin_amdgcn_sched_barrier(0);
}

// two K-steps of the 4-block view branch share one slot (a0 = first half-step, a1 = second)
template <bool ZERO>
__device__ __forceinline__ void l_step4x2(LCtx& c, const float bv0, const float bv1, bool second, f32x16 (&acc)[4])
{
    l_enter(c);
    l_mfma4<ZERO>(c.a0, bv0, acc);
    if (second) l_mfma4<false>(c.a1, bv1, acc);
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+a"(acc[i]));
    l_leave(c);
    __builtin_amdgcn_sched_barrier(0);
}

// H part of an 8-block layer: dst (+)= W * relu(src), 128 K-steps = 16 chunks
// (every part starts chunk-aligned: `c.p = 0` restates that for the compiler, which cannot carry it round a loop)
template <bool ZERO_FIRST>
__device__ __forceinline__ void l_hpart8(LCtx& c, const f32x16 (&src)[8], f32x16 (&dst)[8])
{
    c.p = 0;
    f32x16 cur, nxt;
#pragma unroll
    for (int r = 0; r < 16; ++r) cur[r] = fmaxf(src[0][r], 0.f);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (b + 1 < 8) nxt[r] = fmaxf(src[b + 1][r], 0.f);       // next block's operand, in this step's shadow
            if (ZERO_FIRST && b == 0 && r == 0) l_step8<true>(c, cur[r], dst);
            else l_step8<false>(c, cur[r], dst);
        }
        cur = nxt;
    }
}

// X part of an 8-block layer: 25 groups x 4 K-steps (+ 4 pad slots).  SRC 0: global X, stashing each group in LDS;
// SRC 1: from the stash.
template <int SRC, bool ZERO_FIRST>
__device__ __forceinline__ void l_xpart8(LCtx& c, const f32x4* __restrict__ xt /* + lane */, f32x4* __restrict__ xs /* LDS + lane */,
                                         f32x16 (&dst)[8])
{
    c.p = 0;
    f32x4 xv = SRC ? xs[0] : __builtin_nontemporal_load(xt);
#pragma unroll
    for (int q = 0; q < LS_QX; ++q) {
        f32x4 xn = xv;
        if (q + 1 < LS_QX) xn = SRC ? xs[(q + 1) * 64] : __builtin_nontemporal_load(xt + (q + 1) * 64);
        if (!SRC) xs[q * 64] = xv;
        if (ZERO_FIRST && q == 0) l_step8<true>(c, xv[0], dst); else l_step8<false>(c, xv[0], dst);
        l_step8<false>(c, xv[1], dst);
        l_step8<false>(c, xv[2], dst);
        l_step8<false>(c, xv[3], dst);
        xv = xn;
    }
#pragma unroll
    for (int i = 0; i < (LS_CHUNK - (LS_QX * 4) % LS_CHUNK) % LS_CHUNK; ++i) l_skip(c);
}

// bias slot + 7 padding slots
__device__ __forceinline__ void l_bias8(LCtx& c, f32x16 (&dst)[8])
{
    c.p = 0;
    l_step8<false>(c, 1.f, dst);
#pragma unroll
    for (int i = 0; i < LS_CHUNK - 1; ++i) l_skip(c);
}

__global__ void __launch_bounds__(256) k_mlp_fwd_l(NfMlpLayout L, const float* __restrict__ packed,
                                                   const f32x4* __restrict__ wstream, int nslots,
                                                   const float* __restrict__ X, const int* __restrict__ n_rows, int max_rows,
                                                   const int* __restrict__ row_sample, float4* __restrict__ rgbsigma)
{
    extern __shared__ f32x4 lds4[];        // [3 thirds][8 slots][128] ring, then [4 waves][25 groups][64 lanes] X stash
    const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31, wave = threadIdx.x >> 6;
    const int nrows = min(*n_rows, max_rows);
    const int ntiles = (nrows + 31) >> 5;
    const int ngroups = (ntiles + 3) >> 2;
    f32x4* xs = lds4 + 3 * LS_CHUNK_F4 + wave * (LS_QX * 64) + lane;
    LCtx c;
    c.stream = wstream; c.lane = lane; c.wave = wave;
    c.t0 = lds4; c.t1 = lds4 + LS_CHUNK_F4; c.t2 = lds4 + 2 * LS_CHUNK_F4;
    c.nchunks = nslots / LS_CHUNK; c.chunk_next = 0; c.p = 0;
    // prologue: chunks 0 and 1 in place; chunk 2 is fetched piecewise during the first slots of chunk 0
    l_fetch(c); l_publish(c, c.t0);
    l_fetch(c); l_publish(c, c.t1);
    __syncthreads();
    c.a0 = c.t0[lane]; c.a1 = c.t0[64 + lane];      // slot 0; from here on every slot is prefetched by its predecessor
    for (int tg = blockIdx.x; tg < ngroups; tg += gridDim.x) {
        int tile = tg * 4 + wave;
        if (tile >= ntiles) tile = ntiles - 1;      // idle waves recompute the last tile (keeps the barriers matched)
        const bool owner = (tg * 4 + wave) < ntiles;
        const f32x4* xt = (const f32x4*)X + (size_t)tile * (LS_QX + LS_QD) * 64 + lane;
        const int row = tile * 32 + j;
        const bool row_ok = owner && row < nrows;
        const float* __restrict__ pk = packed + opaque_zero();
        f32x16 accA[8], accB[8];

        // layer 0 = xyz_encoding_1
        l_xpart8<0, true>(c, xt, xs, accA);
        l_bias8(c, accA);
#ifdef NF_L_UNROLL
#pragma unroll
#else
#pragma unroll 1
#endif
        for (int l = 1; l < 9; l += 2) {
            l_hpart8<true>(c, accA, accB);
            l_bias8(c, accB);
            if (l + 1 == 4) {                        // skip layer: cat[input_xyz, h]
                l_xpart8<1, true>(c, nullptr, xs, accA);
                l_hpart8<false>(c, accB, accA);
            } else {
                l_hpart8<true>(c, accB, accA);
            }
            l_bias8(c, accA);
        }
        // accA = xyz_encoding_final (no activation), accB = pre-activation of layer 8 (h8 = relu)
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
        // view branch: hd = relu(W_dir [final | dir feats] + b)
        f32x16 hd[4];
        c.p = 0;
        {
            const f32x4* xd = xt + LS_QX * 64;       // 7 groups = 28 half-steps = 14 slots (+ 2 pad)
            f32x4 xv = __builtin_nontemporal_load(xd);
#pragma unroll
            for (int q = 0; q < LS_QD; ++q) {
                f32x4 xn = xv;
                if (q + 1 < LS_QD) xn = __builtin_nontemporal_load(xd + (q + 1) * 64);
                if (q == 0) l_step4x2<true>(c, xv[0], xv[1], true, hd); else l_step4x2<false>(c, xv[0], xv[1], true, hd);
                l_step4x2<false>(c, xv[2], xv[3], true, hd);
                xv = xn;
            }
#pragma unroll
            for (int i = 0; i < (LS_CHUNK - (LS_QD * 2) % LS_CHUNK) % LS_CHUNK; ++i) l_skip(c);
        }
        c.p = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b)                  // H part: 128 half-steps = 64 slots, no activation on `final`
#pragma unroll
            for (int r = 0; r < 16; r += 2) l_step4x2<false>(c, accA[b][r], accA[b][r + 1], true, hd);
        c.p = 0;
        l_step4x2<false>(c, 1.f, 0.f, false, hd);    // bias half-step
#pragma unroll
        for (int i = 0; i < LS_CHUNK - 1; ++i) l_skip(c);

        const float* wr = pk + L.off_wrgb;
        float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = fmaxf(hd[b][r], 0.f);
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

extern "C" int nf_nerf_mlp_fwd_l(const float* packed, const float* wstream, int cx, int cd, const float* X,
                                 const int32_t* n_rows, int max_rows, const int32_t* row_sample, float* rgbsigma,
                                 nf_stream_t stream)
{
    NF_CHECK_ARG(packed && wstream && X && n_rows && row_sample && rgbsigma, "null pointer");
    if (max_rows <= 0) return NF_OK;
    NfMlpLayout L = mlp_layout(cx, cd);
    NF_CHECK_ARG(L.qx == LS_QX && L.qd == LS_QD, "the LDS-ring path is built for the default 198+54 feature row");
    LStream S = l_stream(L);
    int tiles = (max_rows + 31) / 32;
    int blocks = (tiles + 3) / 4;
    if (blocks > 256) blocks = 256;
    const size_t lds = (size_t)(3 * LS_CHUNK_F4 + 4 * LS_QX * 64) * sizeof(f32x4);
    static bool attr_set[64] = {};
    if (nf_first_use_on_device(attr_set)) {
        hipFuncSetAttribute((const void*)k_mlp_fwd_l, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL(k_mlp_fwd_l, dim3(blocks), dim3(256), lds, (hipStream_t)stream, L, packed, (const f32x4*)wstream,
                       S.nslots, X, n_rows, max_rows, row_sample, (float4*)rgbsigma);
    NF_CHECK_LAUNCH();
    return NF_OK;
}
