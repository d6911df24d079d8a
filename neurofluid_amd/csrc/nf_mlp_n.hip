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
// NN_WPE = workgroups per CU = waves per SIMD.  2: two activation images per workgroup (a layer writes the image the next one reads:
// one barrier per layer), <= 256 registers.  3 (round 6): ONE image — a layer's outputs replace its inputs behind a second barrier
// ("everybody has read") — 40 KB of LDS and <= 168 registers per wave, so that three tiles share a CU's matrix pipes: a tile's
// epilogues / barriers / operand latencies are covered by two other tiles instead of one, and a launch has 768 tile slots instead of
// 512 (the coarse pass of a 4 x 1024-ray training step, ~530 tiles, paid a second round for its last 20).
#ifndef NN_WPE_F
#define NN_WPE_F 3               // forward kernel
#endif
#ifndef NN_WPE_B
#define NN_WPE_B 4               // backward kernel (<= 128 registers since its head weights come from an LDS table; 4 x 35.5 KB of LDS)
#endif
#ifndef NN_IMGS_F
#define NN_IMGS_F (NN_WPE_F >= 3 ? 1 : 2)
#endif
#define NN_IMGS_B (NN_WPE_B >= 3 ? 1 : 2)
// Weight operands are fetched with buffer loads: resource = the blob (scalar registers), scalar offset = part + K-step pair,
// vector offset = the lane's constant 16-B slot.  A global load from a per-lane 64-bit pointer paid two vector adds per load
// (the pair stride of 4 KB does not fit the instruction's immediate) — on the ALUs the fp32 MFMA runs on.
typedef unsigned n_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 n_wload(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff)
{
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0));
}
// Saved activations / mask words are written with buffer stores: resource = the TILE's 32 rows (scalar registers), one vector offset per
// lane for every store of the tile (row j x pitch + the wave's and the half's share of a 256-feature slot), the slot as the scalar
// offset, block / quad as the immediate.  Per-lane 64-bit pointers (one per store site) had the forward kernel at 85 spilled registers.
struct NSave { __amdgpu_buffer_rsrc_t rs; unsigned voff; bool ok; };
// The slot's offset goes into the VECTOR offset (one v_add per epilogue), the scalar offset is the literal 0.  With the slot in an
// SGPR soffset the compiler scheduled `buffer_store_dwordx4 v[0:3], ..., s85 offen` / `v_or_b32 v0, 64, v149` back to back: LLVM's
// hazard recogniser holds that a store of more than 64 bits whose soffset is a REGISTER reads its data early enough for the next
// VALU to overwrite it (GCNHazardRecognizer::createsVALUHazard), and on gfx950 that is not so — the saved activations came back
// with the next store's address in their first dword, in lanes 12-15 of every 16, a few hundred elements per launch, run-dependent.
// Without a register soffset the compiler inserts the wait state itself.
__device__ __forceinline__ void n_save4(const NSave& sv, int slot_bytes, int imm_bytes, f32x4 v)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(n_u32x4, v), sv.rs, (int)sv.voff + slot_bytes + imm_bytes, 0, 0);
}
// mask word, one value at a time: m = 2 m + [x > 0] as v_cmp + v_addc (two vector instructions per value; beside a co-resident
// workgroup's fp32 MFMAs every vector instruction costs matrix time; the compiler's own selection of the C expression is v_cmp,
// v_cndmask, v_or3, v_lshl with s_nops).  After n values, value k sits at bit n - 1 - k.  relu as ONE v_max_f32: fmaxf(x, 0) — and
// __builtin_amdgcn_fmed3f(x, 0, inf) — compile to a canonicalising v_max x, x in front of the v_max (same value for every non-NaN x).
// HAZARD: an asm statement that reads an MFMA's result registers is invisible to the compiler's hazard recogniser — no s_nop in front
// of it when it comes first behind the MFMA (the first version of this file stored accumulator garbage into the saved activations as
// soon as the register allocator scheduled such a read first).  Every epilogue therefore starts with N_MFMA_DRAIN (18 wait states: a
// 16-pass MFMA's results are readable), and these statements are volatile, so they stay behind it.
#define N_MFMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 3")
__device__ __forceinline__ float n_relu(float x)
{
    float y;
    asm volatile("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x));
    return y;
}
__device__ __forceinline__ void n_mask_push(unsigned& m, float x)
{
    asm volatile("v_cmp_gt_f32 vcc, %1, 0\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(x) : "vcc");
}
#ifndef NH_DEPTH
#define NH_DEPTH 4               // read-ahead of the hidden parts' K loops, in K-step pairs
#endif
#ifndef NX_WD
#define NX_WD 2                  // read-ahead of the X parts' weight operands, in groups of 4 K-steps
#endif

struct NCtx {
    int lane, h, j, w, g, c0;      // wave w owns blocks 2w, 2w + 1 = half g = w >> 1, components c0, c0 + 1 of its 16-B operands
};

// dev build (-DNF_N_TIMING): where a tile's time goes — every wave stamps s_memtime at the phase boundaries below and adds the
// differences (shader cycles) to nf_n_prof[kernel][wave][phase]; read / reset through nf_dev_n_prof (tools/n_timing.py).
#ifdef NF_N_TIMING
__device__ unsigned long long nf_n_prof[2][4][16];
__device__ unsigned long long nf_n_trace[2][4096][2];        // [kernel][tile] = {start, end} (s_memtime) of the last launch
#define NT_DECL unsigned long long nt_t = __builtin_amdgcn_s_memtime(), nt_acc[16] = {}
#define NT_MARK(ph) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 0" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
                         nt_acc[ph] += t_ - nt_t; nt_t = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define NT_FLUSH(kern) do { if (c.lane == 0) { unsigned long long tot_ = 0; for (int p_ = 0; p_ < 15; ++p_) { atomicAdd(&nf_n_prof[kern][c.w][p_], nt_acc[p_]); tot_ += nt_acc[p_]; } \
                                               atomicAdd(&nf_n_prof[kern][c.w][15], 1ull); \
                                               if (c.w == 0 && tile < 4096) { nf_n_trace[kern][tile][0] = nt_t - tot_; nf_n_trace[kern][tile][1] = nt_t; } } } while (0)
extern "C" int nf_dev_n_prof(unsigned long long* out, int reset)
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(nf_n_prof), sizeof(unsigned long long) * 128) != hipSuccess) return 1;
    if (reset) { unsigned long long z[128] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(nf_n_prof), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
extern "C" int nf_dev_n_trace(unsigned long long* out)       // out[2][4096][2]
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(nf_n_trace), sizeof(unsigned long long) * 2 * 4096 * 2) != hipSuccess;
}
#else
#define NT_DECL
#define NT_MARK(ph)
#define NT_FLUSH(kern)
#endif

// bias K-step (A = bias, B = 1, C = 0) of an 8-block layer: [2][64][4].  The operand is requested by n_bias_load AHEAD of the
// epilogue + barrier in front of the layer (round 6: the load used to sit right in front of its MFMA, an L2 round trip of
// ~3 600 cycles per layer on the tile's critical path).
__device__ __forceinline__ f32x2 n_bias_load(const NCtx& c, const f32x4* __restrict__ p)
{
    return *(const f32x2*)((const float*)(p + c.g * 64 + c.lane) + c.c0);
}
__device__ __forceinline__ void n_bias2(const f32x2 wv, f32x16 (&acc)[2])
{
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    acc[0] = MFMA32(wv[0], 1.f, z);
    acc[1] = MFMA32(wv[1], 1.f, z);
}

// K-steps whose B operand comes from the feature matrix (25 groups x 4 steps), read from global X both times (layer 0 and the
// skip layer ~30 us later: 32 KB per tile that the L2 still holds; no LDS stash, so two workgroups fit a CU and each SIMD
// has a second tile's wave to issue from while the first one waits at a layer barrier).
// (X, too, is read with buffer loads: resource = the tile's feature groups, vector offset 16 lane, the group as the immediate)
#define N_XLOAD(q) n_wload(xr_, xvoff, (q) * 1024)
template <int nq>
__device__ __forceinline__ void n_xpart2(const NCtx& c, __amdgpu_buffer_rsrc_t wr_, int poff /* part offset in bytes, N layout */,
                                         __amdgpu_buffer_rsrc_t xr_ /* the tile's X */, f32x16 (&acc)[2])
{
    const unsigned xvoff = (unsigned)c.lane * 16u;
    const unsigned voff = (unsigned)(c.w * 64 + c.lane) * 16u;          // pair stride: 4 KB; a group of 4 K-steps = 2 pairs
    // a group is only 4 K-steps x 2 MFMAs = 512 cycles here: X (HBM the first time, L2 the second) is requested XD groups
    // ahead, the weights (L2) two groups ahead
    constexpr int XD = 6, WD = NX_WD;
    f32x4 xr[XD];
    f32x4 wr[WD][2];
#pragma unroll
    for (int q = 0; q < XD; ++q) xr[q] = q < nq ? N_XLOAD(q) : N_XLOAD(0);
#pragma unroll
    for (int q = 0; q < WD; ++q)
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
        for (int k = 0; k + 1 < WD; ++k)
#pragma unroll
            for (int i = 0; i < 2; ++i) wr[k][i] = wr[k + 1][i];
        if (q + XD < nq) xr[XD - 1] = N_XLOAD(q + XD);
        if (q + WD < nq) {
#pragma unroll
            for (int i = 0; i < 2; ++i) wr[WD - 1][i] = n_wload(wr_, voff, poff + ((q + WD) * 2 + i) * 4096);
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
// the first NH_DEPTH weight operands of a hidden part, requested AHEAD of the epilogue + barrier in front of that part (they do not
// depend on the barrier; the activations do): the part then starts on operands that have landed instead of on an L2 round trip
__device__ __forceinline__ void n_hpre(const NCtx& c, __amdgpu_buffer_rsrc_t wr_, int poff, f32x4 (&pre)[NH_DEPTH])
{
    const unsigned voff = (unsigned)(c.w * 64 + c.lane) * 16u;
#pragma unroll
    for (int s = 0; s < NH_DEPTH; ++s) pre[s] = n_wload(wr_, voff, poff + s * 4096);
    __builtin_amdgcn_sched_barrier(0);
}

template <int NS = 128, bool PRE = false>
__device__ __forceinline__ void n_hpart2(const NCtx& c, __amdgpu_buffer_rsrc_t wr_, int poff /* part offset in bytes, N layout */,
                                         const float* __restrict__ act, f32x16 (&acc)[2], const f32x4* pre = nullptr)
{
    constexpr int NP = NS / 2, D = NH_DEPTH;          // K-step pairs; read-ahead in pairs
    const unsigned voff = (unsigned)(c.w * 64 + c.lane) * 16u;
    const float* ap = act + (4 * c.h) * 32 + c.j;           // feature frag_feature(b, r, h) = 32 b + (r & 3) + 8 (r >> 2) + 4 h
    f32x4 ring[D + 1];
    float b0[D + 1], b1[D + 1];
#define NH_F(S) ((32 * ((S) >> 4) + ((S) & 3) + 8 * (((S) & 15) >> 2)) * 32)
#pragma unroll
    for (int s = 0; s < D; ++s) {
        if (PRE) ring[s] = pre[s];
        else ring[s] = n_wload(wr_, voff, poff + s * 4096);
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

// the wave's two blocks -> LDS image (RELU or raw) and, when training, the saved-activation row (row-major, as k_mlp_fwd).
// mword (training): the sign bits of the wave's 32 values, bit 31 - (16 i + r) = [acc[i][r] > 0] — the backward kernel's ReLU masks, one
// dword per lane and layer instead of the 128 B of activations it used to fetch (HBM) at every layer of every tile.
// SIG: this is h8 — the wave also sums its 64 features' share of the sigma head (a partial per lane -> sp[wave][lane]).
template <bool RELU, bool SAVE, bool SIG = false>
__device__ __forceinline__ void n_store2(const NCtx& c, const f32x16 (&acc)[2], float* __restrict__ act, const NSave& sv, int slot,
                                         unsigned* __restrict__ mtile = nullptr, const float* __restrict__ wsig = nullptr,
                                         float* __restrict__ sp = nullptr)
{
    unsigned m = 0;
    float part = 0.f;
    N_MFMA_DRAIN();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int b = 2 * c.w + i;
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            v[r] = RELU ? n_relu(acc[i][r]) : acc[i][r];
            act[(32 * b + (r & 3) + 8 * (r >> 2) + 4 * c.h) * 32 + c.j] = v[r];
            if (SAVE && RELU) n_mask_push(m, acc[i][r]);              // value 16 i + r -> bit 31 - (16 i + r)
            if (SIG) part += v[r] * wsig[(i * 16 + r) * 2];            // wsig: the LDS table + 64 w + h
        }
        if (SAVE && sv.ok) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                f32x4 o = {v[4 * rq], v[4 * rq + 1], v[4 * rq + 2], v[4 * rq + 3]};
                n_save4(sv, slot * 1024, i * 128 + rq * 32, o);        // row + 256 slot + 32 b + 8 rq + 4 h
            }
        }
    }
    if (SAVE && RELU && mtile) mtile[slot * 256 + c.w * 64 + c.lane] = m;
    if (SIG) sp[c.w * 64 + c.lane] = part;
}

#define NN_HEADS (4 * 64 + 12 * 64 + 512 + 384)     // floats behind the two images: sigma partials [wave][lane], rgb partials [wave][channel][lane],
                                                    // the head weights w_sigma [256][2], w_rgb [3][64][2] (per-lane reads: + half)
#define NF_AMASK_SLOTS 10                  // mask words per (tile, wave, lane): activation slots 0..7 (h1..h8) and 9 (view branch); 8 unused

template <bool SAVE, int QX, int QD>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NN_WPE_F, NN_WPE_F))) k_mlp_fwd_n(NfMlpLayout L, const float* __restrict__ packed, const float* __restrict__ X,
                                                   const int* __restrict__ n_rows, int max_rows,
                                                   const int* __restrict__ row_sample, float4* __restrict__ rgbsigma,
                                                   float* __restrict__ acts, unsigned* __restrict__ amask)
{
    extern __shared__ float nlds[];        // act image A, act image B, head partials
    float* actA = nlds;
    float* actB = nlds + (NN_IMGS_F - 1) * NN_ACT;
    float* sp = nlds + NN_IMGS_F * NN_ACT;   // [4][64] sigma partials
    float* rp = sp + 4 * 64;               // [4][3][64] rgb partials
    float* hw = rp + 12 * 64;              // head weights
    NCtx c;
    c.lane = threadIdx.x & 63; c.h = c.lane >> 5; c.j = c.lane & 31; c.w = threadIdx.x >> 6; c.g = c.w >> 1; c.c0 = 2 * (c.w & 1);
    const int nrows = min(*n_rows, max_rows);
    const int ntiles = (nrows + 31) >> 5;
    constexpr int Q = QX + QD;
    for (int i = threadIdx.x; i < 512 + 384; i += 256) hw[i] = i < 512 ? packed[L.off_wsig + i] : packed[L.off_wrgb + i - 512];
    __syncthreads();
    const float* hw_sig = hw + 64 * c.w + c.h;             // this wave's two blocks of w_sigma, this lane's half
    const float* hw_rgb = hw + 512 + 32 * c.w + c.h;       // this wave's block of w_rgb (channel stride 128)
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int z0 = opaque_zero();
        const float* __restrict__ pk = packed + z0;
        const f32x4* P4 = (const f32x4*)pk;
        const __amdgpu_buffer_rsrc_t wr_ = __builtin_amdgcn_make_buffer_rsrc((void*)packed, 0, L.total * 4, 0x27000);
        const __amdgpu_buffer_rsrc_t xr_ = __builtin_amdgcn_make_buffer_rsrc((void*)(X + (size_t)tile * Q * 256), 0, Q * 1024, 0x27000);
        const unsigned xvoff = (unsigned)c.lane * 16u;
        const int row = tile * 32 + c.j;
        const bool row_ok = row < nrows;
        NSave sv;
        sv.ok = SAVE && row_ok;
        sv.voff = (unsigned)c.j * (unsigned)(NF_ACT_STRIDE * 4) + (unsigned)c.w * 256u + (unsigned)c.h * 16u;
        sv.rs = __builtin_amdgcn_make_buffer_rsrc(SAVE ? (void*)(acts + (size_t)tile * 32 * NF_ACT_STRIDE) : (void*)packed, 0,
                                                  32 * NF_ACT_STRIDE * 4, 0x27000);
        unsigned* mtile = (SAVE && amask) ? amask + (size_t)tile * NF_AMASK_SLOTS * 256 : nullptr;     // [slot][wave][lane]
        f32x16 acc[2];
        f32x4 xdir[QD];         // the view-direction feature groups: requested in front of the last layer's K loop, used behind it
        NT_DECL;

        // layer 0 = xyz_encoding_1 -> h1 in A
        n_bias2(n_bias_load(c, P4 + (L.off_bstep[0] >> 2)), acc);
        n_xpart2<QX>(c, wr_, L.off_x[0] * 4 + z0, xr_, acc);
        NT_MARK(0);
        f32x2 bnext = n_bias_load(c, P4 + (L.off_bstep[1] >> 2));
        __builtin_amdgcn_sched_barrier(0);
#ifdef NN_PREFETCH
        f32x4 pre[NH_DEPTH];
        n_hpre(c, wr_, L.off_h[1] * 4 + z0, pre);
#endif
        n_store2<true, SAVE>(c, acc, actA, sv, 0, mtile);
        NT_MARK(1);
        __syncthreads();
        NT_MARK(2);
        float* cur = actA;
        float* nxt = actB;
        float sigma = 0.f;
        float bdir = 0.f;
#pragma unroll 1
        for (int l = 1; l < 9; ++l) {
            n_bias2(bnext, acc);
            if (L.off_x[l] >= 0) n_xpart2<QX>(c, wr_, L.off_x[l] * 4 + z0, xr_, acc);
            NT_MARK(3);
            if (l == 8) {
#pragma unroll
                for (int q = 0; q < QD; ++q) xdir[q] = N_XLOAD(QX + q);
            }
#ifdef NN_PREFETCH
            n_hpart2<128, true>(c, wr_, L.off_h[l] * 4 + z0, cur, acc, pre);
            if (l < 8) n_hpre(c, wr_, L.off_h[l + 1] * 4 + z0, pre);
#else
            n_hpart2(c, wr_, L.off_h[l] * 4 + z0, cur, acc);
#endif
            NT_MARK(4);
            // the next layer's bias operand (the view branch's after the last layer), ahead of this layer's epilogue + barrier
            if (l < 8) bnext = n_bias_load(c, P4 + (L.off_bstep[l + 1] >> 2));
            else bdir = ((const float*)(P4 + (L.off_bstep_dir >> 2) + c.lane))[c.w];
            __builtin_amdgcn_sched_barrier(0);
            if (NN_IMGS_F == 1) __syncthreads();          // one image: everybody has read this layer's inputs
            if (l == 8) {
                // the sigma head: the four waves' partials over h8 (left in sp by layer 7's epilogue, behind its barrier)
                if (c.w == 0) {
                    const float part = ((sp[c.lane] + sp[64 + c.lane]) + sp[128 + c.lane]) + sp[192 + c.lane];
                    sigma = part + __shfl_xor(part, 32, 64) + pk[L.off_bsig];
                }
                NT_MARK(5);
                n_store2<false, SAVE>(c, acc, nxt, sv, 8);      // xyz_encoding_final: no activation
            } else if (l == 7) {
                n_store2<true, SAVE, true>(c, acc, nxt, sv, l, mtile, hw_sig, sp);    // h8
            } else {
                n_store2<true, SAVE>(c, acc, nxt, sv, l, mtile);        // h_{l+1}
            }
            NT_MARK(6);
            __syncthreads();
            NT_MARK(7);
            float* t = cur; cur = nxt; nxt = t;
        }
        // view branch: hd = relu(W_dir [final | dir feats] + b), one block per wave; `cur` holds final
        f32x16 hd;
        {
            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            hd = MFMA32(bdir, 1.f, z);
            const unsigned voff = (unsigned)(c.w * 64 + c.lane) * 16u;               // N layout: [group of 4 steps][wave = block][64] x 16 B
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
        NT_MARK(8);
        {
            // epilogue of the view branch: the wave's block of hd -> saved activations + mask word, and its share of the rgb head
            // (three partial sums over its 32 features' lane halves -> rp; wave 0 adds the four waves' partials behind the barrier).
            // k_mlp_fwd sums a head's 128 / 256 products in ONE chain per lane; here four chains of a quarter each: rgb / sigma agree
            // with it to the last bits, not bit for bit (the hidden activations still do).
            float v[16];
            unsigned m = 0;
            float c0 = 0.f, c1 = 0.f, c2 = 0.f;
            N_MFMA_DRAIN();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                v[r] = n_relu(hd[r]);
                if (SAVE) n_mask_push(m, hd[r]);                       // value r -> bit 15 - r
                c0 += v[r] * hw_rgb[2 * r];
                c1 += v[r] * hw_rgb[128 + 2 * r];
                c2 += v[r] * hw_rgb[256 + 2 * r];
            }
            rp[(c.w * 3 + 0) * 64 + c.lane] = c0;
            rp[(c.w * 3 + 1) * 64 + c.lane] = c1;
            rp[(c.w * 3 + 2) * 64 + c.lane] = c2;
            if (SAVE && sv.ok) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    f32x4 o = {v[4 * rq], v[4 * rq + 1], v[4 * rq + 2], v[4 * rq + 3]};
                    n_save4(sv, 9 * 1024, rq * 32 - (int)c.w * 128, o);        // block w of slot 9: 32 w, not the 64 w of the vector offset
                }
            }
            if (SAVE && mtile) mtile[9 * 256 + c.w * 64 + c.lane] = m;
        }
        NT_MARK(9);
        __syncthreads();
        NT_MARK(10);
        if (c.w == 0) {
            float c0 = ((rp[c.lane] + rp[192 + c.lane]) + rp[384 + c.lane]) + rp[576 + c.lane];
            float c1 = ((rp[64 + c.lane] + rp[256 + c.lane]) + rp[448 + c.lane]) + rp[640 + c.lane];
            float c2 = ((rp[128 + c.lane] + rp[320 + c.lane]) + rp[512 + c.lane]) + rp[704 + c.lane];
            c0 += __shfl_xor(c0, 32, 64); c1 += __shfl_xor(c1, 32, 64); c2 += __shfl_xor(c2, 32, 64);
            c0 += pk[L.off_brgb]; c1 += pk[L.off_brgb + 1]; c2 += pk[L.off_brgb + 2];
            if (c.h == 0 && row_ok) {
                float4 o;
                o.x = 1.f / (1.f + expf(-c0)); o.y = 1.f / (1.f + expf(-c1)); o.z = 1.f / (1.f + expf(-c2)); o.w = sigma;
                rgbsigma[row_sample[row]] = o;
            }
        }
        NT_MARK(11);
        __syncthreads();        // the images and the partials are rewritten by the next tile
        NT_MARK(12);
        NT_FLUSH(0);
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
// BITS: the ReLU masks come from the forward kernel's mask word (bit 16 i + r of `mbits`) instead of the saved activations
template <int MODE, bool BITS = false>
__device__ __forceinline__ void n_bwd_slot(const NCtx& c, const f32x16 (&raw)[2], const float* __restrict__ hrow, const float* __restrict__ wsig,
                                           float dsig, float* __restrict__ img, const NSave& sv, int slot, unsigned mbits = 0)
{
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int b = 2 * c.w + i;
        float v[16];
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            f32x4 hv = {1.f, 1.f, 1.f, 1.f};
            if (MODE != 0 && !BITS) hv = *(const f32x4*)(hrow + 32 * b + 8 * rq + 4 * c.h);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * rq + e;
                float x = raw[i][r];
                if (MODE == 2) x += dsig * wsig[(i * 16 + r) * 2];              // wsig: the LDS table + 64 w + h
                if (MODE != 0) {
                    if (BITS) x = __builtin_bit_cast(float, __builtin_bit_cast(int, x) & ((int)(mbits << (16 * i + r)) >> 31));    // bit 31 - (16 i + r)
                    else x = hv[e] > 0.f ? x : 0.f;
                }
                v[r] = x;
                if (img) img[(32 * b + (r & 3) + 8 * (r >> 2) + 4 * c.h) * 32 + c.j] = x;
            }
        }
        if (sv.ok) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                f32x4 o = {v[4 * rq], v[4 * rq + 1], v[4 * rq + 2], v[4 * rq + 3]};
                n_save4(sv, slot * 1024, i * 128 + rq * 32, o);
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

// the wave's two blocks of a feature-gradient accumulator -> dX rows (row-major, pitch cx + cd floats): columns col0 + 64 w + 32 i +
// 8 rq + 4 h + e; whole quads where they fit under `ncols`, single floats at the boundary (the next column belongs to the other part
// of the row, or to the next row)
__device__ __forceinline__ void n_dx_store(const NCtx& c, const f32x16 (&acc)[2], const NSave& sx, int col0, int ncols)
{
    if (!sx.ok) return;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int f0 = 64 * c.w + 32 * i + 8 * rq + 4 * c.h;
            const f32x4 o = {acc[i][4 * rq], acc[i][4 * rq + 1], acc[i][4 * rq + 2], acc[i][4 * rq + 3]};
            if (f0 + 4 <= ncols) n_save4(sx, col0 * 4, i * 128 + rq * 32, o);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (f0 + e < ncols)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i][4 * rq + e]), sx.rs, (int)sx.voff + col0 * 4 + i * 128 + rq * 32 + e * 4, 0, 0);
            }
        }
}

// DX (round 6): the kernel also produces dL/dX = dpre_1 W_1[:, :cx] + dpre_5 W_5[:, :cx] | dpre_dir W_dir[:, 256:] (the gradient of the
// feature row, what the end-to-end step scatters to the particles) as three more K loops over images it has in LDS anyway, instead
// of three latency-bound GEMMs over dpre per pass behind it (53 us each at the end-to-end step's 5-17 thousand rows).
template <bool BITS, bool DX>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DX ? 2 : NN_WPE_B, DX ? 2 : NN_WPE_B))) k_mlp_bwd_n(NfMlpLayout L, NfMlpLayoutT T, const float* __restrict__ packed,
                                                   const float* __restrict__ packed_t, const float* __restrict__ acts,
                                                   const unsigned* __restrict__ amask,
                                                   const int* __restrict__ n_rows, int max_rows, const int* __restrict__ row_sample,
                                                   const float4* __restrict__ rgbsigma, const float4* __restrict__ d_rgbsigma,
                                                   float* __restrict__ dpre, float* __restrict__ dX)
{
    extern __shared__ float nlds[];
    float* cur = nlds;
    float* nxt = nlds + (NN_IMGS_B - 1) * NN_ACT;
    float* hw = nlds + NN_IMGS_B * NN_ACT;          // head weights w_sigma [256][2], w_rgb [3][64][2]: per-lane reads (+ half) without per-lane pointers
    NCtx c;
    c.lane = threadIdx.x & 63; c.h = c.lane >> 5; c.j = c.lane & 31; c.w = threadIdx.x >> 6; c.g = c.w >> 1; c.c0 = 2 * (c.w & 1);
    for (int i = threadIdx.x; i < 512 + 384; i += 256) hw[i] = i < 512 ? packed[L.off_wsig + i] : packed[L.off_wrgb + i - 512];
    __syncthreads();
    const float* hw_sig = hw + 64 * c.w + c.h;
    const float* hw_rgb = hw + 512 + 32 * c.w + c.h;
    const int nrows = min(*n_rows, max_rows);
    const int ntiles = (nrows + 31) >> 5;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int z0 = opaque_zero();
        const float* __restrict__ pk = packed + z0;
        const __amdgpu_buffer_rsrc_t wt_ = __builtin_amdgcn_make_buffer_rsrc((void*)packed_t, 0, T.total * 4, 0x27000);
        const int row = tile * 32 + c.j;
        const bool valid = row < nrows;
        const float* arow = BITS ? nullptr : acts + (size_t)(valid ? row : 0) * NF_ACT_STRIDE;
        NSave sv;              // this lane's dpre row, as the forward's saved-activation stores (pitch NF_DPRE_STRIDE)
        sv.ok = valid;
        sv.voff = (unsigned)c.j * (unsigned)(NF_DPRE_STRIDE * 4) + (unsigned)c.w * 256u + (unsigned)c.h * 16u;
        sv.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(dpre + (size_t)tile * 32 * NF_DPRE_STRIDE), 0, 32 * NF_DPRE_STRIDE * 4, 0x27000);
        NSave sx;              // this lane's dX row
        const int xpitch = L.cx + L.cd;
        sx.ok = DX && valid;
        sx.voff = (unsigned)c.j * (unsigned)(xpitch * 4) + (unsigned)c.w * 256u + (unsigned)c.h * 16u;
        sx.rs = __builtin_amdgcn_make_buffer_rsrc(DX ? (void*)(dX + (size_t)tile * 32 * xpitch) : (void*)dpre, 0, 32 * xpitch * 4, 0x27000);
        f32x16 accx[2];        // DX: the position-like features' gradient, summed over the skip layer (slot 4) and the first layer (slot 0)
        // BITS: the ReLU masks of this wave's blocks, one dword per layer (nf_nerf_mlp_fwd_n2 wrote them): the view branch's now, a
        // layer's in front of the K loop that precedes its use — k_mlp_bwd_n without them fetched 128 B of saved activations per lane
        // (HBM) at every slot, with nothing to do until they arrived: as long as the MFMAs of the layer (179 k of 413 k cycles per tile)
        const unsigned* mrow = BITS ? amask + ((size_t)tile * NF_AMASK_SLOTS * 4 + c.w) * 64 + c.lane : nullptr;
        unsigned mcur = BITS ? mrow[9 * 256] : 0u;
        float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f), g4 = o4;
        NT_DECL;
        if (valid) {
            const int sample = row_sample[row];
            o4 = rgbsigma[sample];
            g4 = d_rgbsigma[sample];
        }
        // rgb = sigmoid(z): dz = g * y (1 - y)
        const float dz0 = g4.x * o4.x * (1.f - o4.x), dz1 = g4.y * o4.y * (1.f - o4.y), dz2 = g4.z * o4.z * (1.f - o4.z);
        const float dsig = g4.w;
        if (valid && c.h == 0 && c.w == 0) *(float4*)(dpre + (size_t)row * NF_DPRE_STRIDE + 2432) = make_float4(dz0, dz1, dz2, dsig);
        // slot 9: d(view-branch hidden) = [hd > 0] W_rgb^T dz, block w by wave w
        {
            const int b = c.w;
            float v[16];
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                f32x4 hv = {1.f, 1.f, 1.f, 1.f};
                if (!BITS) hv = *(const f32x4*)(arow + 9 * 256 + 32 * b + 8 * rq + 4 * c.h);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * rq + e;
                    const float x = dz0 * hw_rgb[2 * r] + dz1 * hw_rgb[128 + 2 * r] + dz2 * hw_rgb[256 + 2 * r];
                    v[r] = BITS ? __builtin_bit_cast(float, __builtin_bit_cast(int, x) & ((int)(mcur << (16 + r)) >> 31)) : (hv[e] > 0.f ? x : 0.f);
                    cur[(32 * b + (r & 3) + 8 * (r >> 2) + 4 * c.h) * 32 + c.j] = v[r];
                }
            }
            if (valid) {
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    f32x4 o = {v[4 * rq], v[4 * rq + 1], v[4 * rq + 2], v[4 * rq + 3]};
                    n_save4(sv, 9 * 1024, rq * 32 - (int)c.w * 128, o);        // block w of slot 9: 32 w, not the 64 w of the vector offset
                }
            }
        }
        NT_MARK(0);
#ifdef NN_PREFETCH
        f32x4 pre[NH_DEPTH];
        n_hpre(c, wt_, T.off_dir * 4 + z0, pre);
#endif
        __syncthreads();
        NT_MARK(1);
        // raw d(final) = W_dir[:, :256]^T slot 9: 64 K-steps over the 128 hidden units
        f32x16 acc[2];
        n_zero2(acc);
#ifdef NN_PREFETCH
        n_hpart2<64, true>(c, wt_, T.off_dir * 4 + z0, cur, acc, pre);
#else
        n_hpart2<64>(c, wt_, T.off_dir * 4 + z0, cur, acc);
#endif
        if (DX) {              // the direction-like features' gradient, from the same image (slot 9)
            n_zero2(accx);
            n_hpart2<64>(c, wt_, T.off_dxd * 4 + z0, cur, accx);
            n_dx_store(c, accx, sx, L.cx, L.cd);
            n_zero2(accx);
        }
        NT_MARK(2);
        const float* ws_ = hw_sig;
#pragma unroll 1
        for (int g = 8; g >= 1; --g) {
#ifdef NN_PREFETCH
            n_hpre(c, wt_, T.off_h[g] * 4 + z0, pre);
#endif
            if (NN_IMGS_B == 1) __syncthreads();
            if (g == 8) n_bwd_slot<0, BITS>(c, acc, nullptr, ws_, dsig, nxt, sv, 8);
            else if (g == 7) n_bwd_slot<2, BITS>(c, acc, arow + 7 * 256, ws_, dsig, nxt, sv, 7, mcur);
            else n_bwd_slot<1, BITS>(c, acc, arow + g * 256, ws_, dsig, nxt, sv, g, mcur);
            NT_MARK(3);
            __syncthreads();
            NT_MARK(4);
            float* t = cur; cur = nxt; nxt = t;
            if (BITS) mcur = mrow[(g - 1) * 256];           // the next slot's masks (slot g - 1; slot 0 after the loop)
            n_zero2(acc);
#ifdef NN_PREFETCH
            n_hpart2<128, true>(c, wt_, T.off_h[g] * 4 + z0, cur, acc, pre);
#else
            n_hpart2<128>(c, wt_, T.off_h[g] * 4 + z0, cur, acc);
#endif
            if (DX && g == 4) n_hpart2<128>(c, wt_, T.off_dx4 * 4 + z0, cur, accx);        // the image holds slot 4 (the skip layer's dpre)
            NT_MARK(5);
        }
        // slot 0 = [h1 > 0] d_h1
        if (DX) {              // slot 0 goes to the image, too: dX += dpre_1 W_1[:, :cx]
            if (NN_IMGS_B == 1) __syncthreads();
            n_bwd_slot<1, BITS>(c, acc, arow, ws_, dsig, nxt, sv, 0, mcur);
            __syncthreads();
            n_hpart2<128>(c, wt_, T.off_dx0 * 4 + z0, nxt, accx);
            n_dx_store(c, accx, sx, 0, L.cx);
        } else
        n_bwd_slot<1, BITS>(c, acc, arow, ws_, dsig, nullptr, sv, 0, mcur);
        NT_MARK(6);
        __syncthreads();        // the images are rewritten by the next tile
        NT_MARK(7);
        NT_FLUSH(1);
    }
}

#ifdef NN_PERSIST            // dev: a fixed grid of NN_PERSIST workgroups walking the tiles (is it the dispatch of ~2 000 workgroups that costs?)
#define NN_GRID(t) ((t) < NN_PERSIST ? (t) : NN_PERSIST)
#else
#define NN_GRID(t) (t)
#endif
static int mlp_bwd_n_launch(const float* packed, const float* packed_t, int cx, int cd, const float* acts, const uint32_t* amask,
                            const int32_t* n_rows, int max_rows, const int32_t* row_sample, const float* rgbsigma,
                            const float* d_rgbsigma, float* dpre, float* dX, nf_stream_t stream)
{
    NF_CHECK_ARG(packed && packed_t && (acts || amask) && n_rows && row_sample && rgbsigma && d_rgbsigma && dpre, "null pointer");
    NF_CHECK_ARG(cx >= 1 && cx <= 256 && cd >= 1 && cd <= 256, "bad channel counts");
    if (max_rows <= 0) return NF_OK;
    NfMlpLayout L = mlp_layout(cx, cd);
    NfMlpLayoutT T = mlp_layout_t();
    const int tiles = (max_rows + 31) / 32;
    const size_t lds = (size_t)(NN_IMGS_B * NN_ACT + 512 + 384) * sizeof(float);       // 35.5 KB (NN_WPE = 3) / 67.5 KB
    static bool attr_set[64] = {};
    if (nf_first_use_on_device(attr_set)) {
        hipFuncSetAttribute((const void*)k_mlp_bwd_n<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)k_mlp_bwd_n<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)k_mlp_bwd_n<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    if (amask && dX)
        hipLaunchKernelGGL((k_mlp_bwd_n<true, true>), dim3(NN_GRID(tiles)), dim3(256), lds, (hipStream_t)stream, L, T, packed, packed_t, acts, amask, n_rows,
                           max_rows, row_sample, (const float4*)rgbsigma, (const float4*)d_rgbsigma, dpre, dX);
    else if (amask)
        hipLaunchKernelGGL((k_mlp_bwd_n<true, false>), dim3(NN_GRID(tiles)), dim3(256), lds, (hipStream_t)stream, L, T, packed, packed_t, acts, amask, n_rows,
                           max_rows, row_sample, (const float4*)rgbsigma, (const float4*)d_rgbsigma, dpre, dX);
    else
        hipLaunchKernelGGL((k_mlp_bwd_n<false, false>), dim3(NN_GRID(tiles)), dim3(256), lds, (hipStream_t)stream, L, T, packed, packed_t, acts, amask, n_rows,
                           max_rows, row_sample, (const float4*)rgbsigma, (const float4*)d_rgbsigma, dpre, dX);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_nerf_mlp_bwd_n(const float* packed, const float* packed_t, int cx, int cd, const float* acts,
                                 const int32_t* n_rows, int max_rows, const int32_t* row_sample, const float* rgbsigma,
                                 const float* d_rgbsigma, float* dpre, nf_stream_t stream)
{
    NF_CHECK_ARG(acts, "null pointer");
    return mlp_bwd_n_launch(packed, packed_t, cx, cd, acts, nullptr, n_rows, max_rows, row_sample, rgbsigma, d_rgbsigma, dpre, nullptr, stream);
}

extern "C" int nf_nerf_mlp_bwd_n2(const float* packed, const float* packed_t, int cx, int cd, const uint32_t* amask,
                                  const int32_t* n_rows, int max_rows, const int32_t* row_sample, const float* rgbsigma,
                                  const float* d_rgbsigma, float* dpre, nf_stream_t stream)
{
    NF_CHECK_ARG(amask, "null pointer");
    return mlp_bwd_n_launch(packed, packed_t, cx, cd, nullptr, amask, n_rows, max_rows, row_sample, rgbsigma, d_rgbsigma, dpre, nullptr, stream);
}

extern "C" int nf_nerf_mlp_bwd_n3(const float* packed, const float* packed_t, int cx, int cd, const uint32_t* amask,
                                  const int32_t* n_rows, int max_rows, const int32_t* row_sample, const float* rgbsigma,
                                  const float* d_rgbsigma, float* dpre, float* dX, nf_stream_t stream)
{
    NF_CHECK_ARG(amask && dX, "null pointer");
    return mlp_bwd_n_launch(packed, packed_t, cx, cd, nullptr, amask, n_rows, max_rows, row_sample, rgbsigma, d_rgbsigma, dpre, dX, stream);
}

extern "C" size_t nf_nerf_amask_words(int max_rows) { return (size_t)((max_rows + 31) / 32) * NF_AMASK_SLOTS * 256; }

static int mlp_fwd_n_launch(const float* packed, int cx, int cd, const float* X, const int32_t* n_rows, int max_rows,
                            const int32_t* row_sample, float* rgbsigma, float* acts, uint32_t* amask, nf_stream_t stream)
{
    NF_CHECK_ARG(packed && X && n_rows && row_sample && rgbsigma, "null pointer");
    NF_CHECK_ARG(cx >= 1 && cx <= 256 && cd >= 1 && cd <= 256, "bad channel counts");
    if (max_rows <= 0) return NF_OK;
    NfMlpLayout L = mlp_layout(cx, cd);
    NF_CHECK_ARG(L.qx == 25 && L.qd == 7, "built for the default 198 + 54 feature row (other encodings: nf_nerf_mlp_fwd)");
    const int tiles = (max_rows + 31) / 32;
    const size_t lds = (size_t)(NN_IMGS_F * NN_ACT + NN_HEADS) * sizeof(float);       // 39.5 KB (NN_WPE = 3) / 71.5 KB
    static bool attr_set[64] = {};
    if (nf_first_use_on_device(attr_set)) {
        hipFuncSetAttribute((const void*)k_mlp_fwd_n<true, 25, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)k_mlp_fwd_n<false, 25, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    hipStream_t st = (hipStream_t)stream;
    if (acts)
        hipLaunchKernelGGL((k_mlp_fwd_n<true, 25, 7>), dim3(NN_GRID(tiles)), dim3(256), lds, st, L, packed, X, n_rows, max_rows, row_sample,
                           (float4*)rgbsigma, acts, amask);
    else
        hipLaunchKernelGGL((k_mlp_fwd_n<false, 25, 7>), dim3(NN_GRID(tiles)), dim3(256), lds, st, L, packed, X, n_rows, max_rows, row_sample,
                           (float4*)rgbsigma, acts, (uint32_t*)nullptr);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_nerf_mlp_fwd_n(const float* packed, int cx, int cd, const float* X, const int32_t* n_rows, int max_rows,
                                 const int32_t* row_sample, float* rgbsigma, float* acts, nf_stream_t stream)
{
    return mlp_fwd_n_launch(packed, cx, cd, X, n_rows, max_rows, row_sample, rgbsigma, acts, nullptr, stream);
}

extern "C" int nf_nerf_mlp_fwd_n2(const float* packed, int cx, int cd, const float* X, const int32_t* n_rows, int max_rows,
                                  const int32_t* row_sample, float* rgbsigma, float* acts, uint32_t* amask, nf_stream_t stream)
{
    NF_CHECK_ARG(acts && amask, "null pointer (the mask words go with saved activations)");
    return mlp_fwd_n_launch(packed, cx, cd, X, n_rows, max_rows, row_sample, rgbsigma, acts, amask, stream);
}
