// nf_cconv_gf.hip — G-free continuous convolution for the inference step (models/transmodel.py:116-131, conv1..conv3 + their
// Linear branches), round 3.  Open3D's own order: gather a patch per output point, then ONE contraction with the filter —
//     y[i][co] = sum_{node c} sum_{ci} Z_c[i][ci] * K[c][ci][co] + Linear(x_i)[co] + biases (+ residual),
//     Z_c[i][ci] = sum over the pairs (i <- j) that touch filter node c of  w_c(i, j) * relu(x[j][ci])
// instead of round 2's "transform, then gather", which materialised G[j][node][co] = x[j] K[node] (82 MB written + 132 MB read
// back per 64-channel layer — the whole cost of those kernels).  Nothing of size n x 64 x C exists here:
//
//   * The contraction index is (node, ci): 64 x Cin (+ Cin for the Linear branch, "node 64").  Work is cut into UNITS =
//     (tile of 32 output points) x (one x-row of 4 filter nodes: same (y, z) node, x = 0..3), 16 per tile, + 1 unit for the
//     Linear branch.  A unit's Z is 4 x 32 x Cin floats (48 KB at Cin = 96) and lives in LDS only.
//   * A workgroup is 8 waves: 4 PRODUCER waves build Z of unit u + 1 while 4 CONSUMER waves multiply Z of unit u (double
//     buffer, one s_barrier per unit).  Producers: 8 threads per output point walk that point's ROW-ENTRY LIST for the unit's
//     (y, z) row (nf_trans.hip:k_trans_front bucketed the pairs by row; an entry = neighbour, x base node, two weights),
//     gather relu(x[j]) with 16-byte loads and accumulate the 4 nodes of the row in REGISTERS — every x[j] row is read once
//     per (pair, row) = 4 times per pair instead of 8, no read-modify-write, no scan of the other rows' pairs.  The entries of
//     the NEXT unit and the row offsets of the one after are requested a phase ahead, so a phase's dependent chain is the
//     x gathers alone.  Consumers: v_mfma_f32_32x32x2_f32 with M = the 32 points (A = Z from LDS, one ds_read_b128 per 4
//     K-steps), N = Cout, K split in quarters over the 4 waves; B = the filter pre-packed so that a lane's operands of 4
//     consecutive K-steps are one 16-byte load from L2 (1.5 MB per layer, shared by all workgroups).
//   * Stream-K: the (tile, row) units of the whole launch are dealt to one persistent workgroup per CU in contiguous,
//     cost-balanced ranges (154 tiles do not fill 256 CUs; 2 618 units do).  A tile that is shared by several workgroups
//     leaves one partial accumulator slab per (workgroup, K-quarter); a small epilogue launch adds them up in a fixed order
//     (deterministic) with the biases, the residual and — for the last layer — the position / velocity update.
#include "nf_common.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

#define GF_TILE 32
#define GF_UNITS 17            // 16 rows of 4 filter nodes + the Linear branch
#define GF_COST 65             // filter nodes per tile, Linear branch included
#ifndef GF_PWAVES
#define GF_PWAVES 4
#endif                          // producer waves (8 were measured: no gain on the 64-channel layer, spills on the 96-channel one)
#define GF_THREADS (256 + 64 * GF_PWAVES)
#define GF_ROFF_PITCH 20       // uint16 per particle (17 used)
#define GF_TPP (2 * GF_PWAVES)   // producer threads per output point
#define GF_SLOTS (32 / GF_TPP)  // entry slots per producer thread: 32 entries of a (point, row) list are prefetched

struct GfArgs {
    const float* x;            // (n x CIN) features of the previous layer (ReLU applied on load when relu)
    int n, relu;
    const uint16_t* roff;      // [n][GF_ROFF_PITCH] row offsets into the particle's entry list
    const uint32_t* ent;       // [n][4 * pitch][3] row entries {j | cx << 30, w(cx), w(cx + 1)}
    int pitch;                 // pairs per particle
    const float* wp;           // packed filter (nf_cconv_gf_pack)
    float* scratch;            // [tiles][maxseg][4][COUTP][32] partial accumulators
    int tiles, nwg, maxseg, ctot;
    volatile int* done_word;   // or null: a host-mapped word that receives step_id when this launch STARTS (nf_trans_step: everything in
    int step_id;               // front of the layers — search, overflow words — is complete then; the host spins on it)
};

// workgroup w owns the units g with gf_begin(w) <= g < gf_begin(w + 1); unit g = tile * 17 + u starts at cost
// tile * 65 + 4 u.  owner(g) = floor(cost(g) * nwg / ctot).
__device__ __host__ __forceinline__ int gf_begin(int w, int nwg, int ctot)
{
    const long long tgt = ((long long)w * ctot + nwg - 1) / nwg;
    const int tile = (int)(tgt / GF_COST), rem = (int)(tgt % GF_COST);
    return tile * GF_UNITS + (rem + 3) / 4;
}
__device__ __host__ __forceinline__ int gf_owner(long long cost, int nwg, int ctot) { return (int)(cost * nwg / ctot); }


// SPLIT = false: the contraction in fp32 (v_mfma_f32_32x32x2_f32: exact fp32 products and sums, the reference's arithmetic).
// SPLIT = true:  every operand as hi + lo fp16 (z = zh + zl, 22 significant bits), three v_mfma_f32_32x32x16_f16 per product
//                block (zh wh + zh wl + zl wh), fp32 accumulate — fp32-LEVEL accuracy (the dropped zl wl term is < 2^-22
//                relative) on the fp16 matrix pipe.  On gfx950 the fp32 MFMA runs at the fp32 VECTOR rate, i.e. on the same
//                ALUs as the producers' gather arithmetic (measured: producer-only 45 us + consumer-only 44 us = 81 us together,
//                no overlap at all); the fp16 matrix pipe is a separate unit, 16x faster, and does overlap.
template <int CIN, int NB, bool RELU, bool SPLIT>
__global__ void __launch_bounds__(GF_THREADS) k_cconv_gf(GfArgs A)
{
    constexpr int PITCH = CIN + 4;             // floats per Z row: (CIN + 4) / 4 odd -> ds_read_b128 of 16 rows conflict-free
    constexpr int ZC = GF_TILE * PITCH;        // one filter node
    constexpr int ZB = 4 * ZC;                 // one buffer (4 nodes)
    constexpr int NQ = (CIN / 4 + GF_TPP - 1) / GF_TPP;   // 16-byte quads of an x row per producer thread (16 threads per point; at
                                               // CIN = 96 the second quad exists for the first 8 threads of a point only)
    constexpr int QW = CIN / 32;               // quad-groups (4 K-steps each) per consumer wave and node
    constexpr int COUTP = 32 * NB;
    // split layout (halves): per buffer [hi: 4 nodes][lo: 4 nodes], a node = 32 rows of CIN + 8 halves (208 / 144 bytes:
    // 16-byte aligned, and 13 / 9 x 16 bytes -> the ds_read_b128 of 16 rows fall on distinct banks)
    constexpr int PH = CIN + 8, ZCH = GF_TILE * PH, ZLO = 4 * ZCH, ZBH = 2 * ZLO;
    extern __shared__ float Z[];               // fp32: 2 * ZB floats; split: 2 * ZBH halves
    _Float16* const Zh16 = (_Float16*)Z;

    // the arguments as plain locals (a by-value struct whose address reaches a lambda is kept — and re-read — in scratch memory)
    const float* const a_x = A.x;
    const int a_n = A.n, a_pitch = A.pitch, a_nwg = A.nwg, a_maxseg = A.maxseg, a_ctot = A.ctot;
    const uint16_t* const a_roff = A.roff;
    const uint32_t* const a_ent = A.ent;
    const float* const a_wp = A.wp;
    float* const a_scratch = A.scratch;
    const int w = blockIdx.x;
    if (A.done_word && w == 0 && threadIdx.x == 0) *A.done_word = A.step_id;
    const int g0 = gf_begin(w, a_nwg, a_ctot), g1 = gf_begin(w + 1, a_nwg, a_ctot);
    const int nun = g1 - g0;
    // the role is a SCALAR (wave-uniform) value: the two halves below are separate scalar branches, each wave meets exactly
    // the s_barriers of its own half
    const int hwave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
#ifdef GF_AB_SPLIT_SIMD
    // roles by SIMD: a workgroup's waves go to the SIMDs in a cyclic order, so waves w and w + 4 share one.  Consumers = waves
    // {0, 1, 4, 5} (two SIMDs run nothing but MFMAs), producers = waves {2, 3, 6, 7} (the other two run nothing but the gather)
    const bool is_consumer = ((hwave >> 1) & 1) == 0;
    const int wave = (hwave & 1) | ((hwave >> 2) << 1);          // index inside the role: 0..3
#else
    const bool is_consumer = hwave < 4;
    const int wave = hwave & 3;
#endif

    if (is_consumer) {
        // ------------------------------------------------------------------ consumers
#ifdef GF_AB_PRIO_C
        __builtin_amdgcn_s_setprio(GF_AB_PRIO_C);
#endif
      if constexpr (SPLIT) {
        constexpr int KS = CIN / 16, KP = 4 / NB, SP = KS / KP;      // K-steps per node; K parts; K-steps of this wave per node
        static_assert(KS % KP == 0, "the K-steps of a node must split evenly over the waves of an N-block");
        const int nbw = wave % NB, kp = wave / NB, m = lane & 31, h = lane >> 5;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const h8* wp8 = (const h8*)a_wp;
        h8 Bh[2][SP], Bl[2][SP], Ah[2][SP], Al[2][SP];
        auto load_b = [&](int node, h8 (&bh)[SP], h8 (&bl)[SP]) __attribute__((always_inline)) {
#pragma unroll
            for (int si = 0; si < SP; ++si) {
                const size_t base = ((size_t)((node * KS + kp * SP + si) * NB + nbw) * 2) * 64 + lane;
                bh[si] = wp8[base];
                bl[si] = wp8[base + 64];
            }
        };
        auto load_a = [&](const _Float16* Zc, h8 (&ah)[SP], h8 (&al)[SP]) __attribute__((always_inline)) {
#pragma unroll
            for (int si = 0; si < SP; ++si) {
                ah[si] = *(const h8*)(Zc + 16 * (kp * SP + si));
                al[si] = *(const h8*)(Zc + ZLO + 16 * (kp * SP + si));
            }
        };
        auto mma = [&](const h8 (&ah)[SP], const h8 (&al)[SP], const h8 (&bh)[SP], const h8 (&bl)[SP]) __attribute__((always_inline)) {
#pragma unroll
            for (int si = 0; si < SP; ++si) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[si], bh[si], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[si], bl[si], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[si], bh[si], acc, 0, 0, 0);
            }
        };
        auto first_node = [&](int g) __attribute__((always_inline)) { const int u = g % GF_UNITS; return u < 16 ? 4 * u : 64; };
        auto run_unit = [&](auto par, auto ncells, const _Float16* Zb, int cell0, int next_node) __attribute__((always_inline)) {
            constexpr int P0 = decltype(par)::value, NC = decltype(ncells)::value;
            load_a(Zb, Ah[P0], Al[P0]);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int cur = (P0 + c) & 1, nx = cur ^ 1;
                const int nxt = c + 1 < NC ? cell0 + c + 1 : next_node;
                if (c + 1 < NC) load_a(Zb + (c + 1) * ZCH, Ah[nx], Al[nx]);
                if (nxt >= 0) load_b(nxt, Bh[nx], Bl[nx]);
#ifndef GF_AB_NO_MFMA
                mma(Ah[cur], Al[cur], Bh[cur], Bl[cur]);
#endif
            }
        };
        int parity = 0;
        if (nun > 0) load_b(first_node(g0), Bh[0], Bl[0]);
        for (int p = 0; p <= nun; ++p) {
            if (p >= 1) {
                const int g = g0 + p - 1, tile = g / GF_UNITS, u = g - tile * GF_UNITS;
                const _Float16* Zb = Zh16 + ((p - 1) & 1) * ZBH + m * PH + 8 * h;
                const int next_node = p < nun ? first_node(g + 1) : -1;
                if (u < 16) {
                    if (parity == 0) run_unit(std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{}, Zb, 4 * u, next_node);
                    else run_unit(std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{}, Zb, 4 * u, next_node);
                } else {
                    if (parity == 0) run_unit(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, Zb, 64, next_node);
                    else run_unit(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, Zb, 64, next_node);
                    parity ^= 1;
                }
                const bool last_of_tile = (p == nun) || ((g + 1) / GF_UNITS != tile);
                if (last_of_tile) {
                    // partial slab of (tile, segment, wave): the wave's N-block [32 cols][32 rows], its share of K
                    const int seg = w - gf_owner((long long)tile * GF_COST, a_nwg, a_ctot);
                    float* slab = a_scratch + ((size_t)(tile * a_maxseg + seg) * 4 + wave) * (32 * GF_TILE);
#pragma unroll
                    for (int gr = 0; gr < 4; ++gr) {
                        const float4 v = make_float4(acc[4 * gr], acc[4 * gr + 1], acc[4 * gr + 2], acc[4 * gr + 3]);
                        *(float4*)(slab + m * GF_TILE + 8 * gr + 4 * h) = v;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                }
            }
            __syncthreads();
        }
        return;
      } else {
        const int kq = wave, m = lane & 31, h = lane >> 5;
        f32x16 acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
        const float4* wp4 = (const float4*)a_wp;
        // B operands (filter, from L2) run one filter node AHEAD of the MFMAs along the workgroup's whole unit sequence —
        // also across the s_barrier of a unit boundary (the filter does not depend on it; only the A operands from LDS do).
        // Two operand buffers in STATIC ping-pong (the parity is a template argument): copying a prefetched register into
        // the "current" one would wait for the load it was meant to hide.
        float4 Bf[2][QW][NB], Af[2][QW];
#if defined(GF_AB_NO_BLOAD) || defined(GF_AB_NO_ALOAD)
#pragma unroll
        for (int z1 = 0; z1 < 2; ++z1)
#pragma unroll
            for (int z2 = 0; z2 < QW; ++z2) {
                Af[z1][z2] = make_float4(1.f + lane, 2.f, 3.f, 4.f);
#pragma unroll
                for (int z3 = 0; z3 < NB; ++z3) {
                    Bf[z1][z2][z3] = make_float4(0.5f, 0.25f + lane, 0.125f, 1.f);
                }
            }
#endif
        auto load_b = [&](int node, float4 (&b)[QW][NB]) {
#ifdef GF_AB_NO_BLOAD
            return;
#endif
#pragma unroll
            for (int gq = 0; gq < QW; ++gq)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    b[gq][nb] = wp4[((size_t)(node * (CIN / 8) + kq * QW + gq) * NB + nb) * 64 + lane];
                }
        };
        auto load_a = [&](const float* Zc, float4 (&a)[QW]) {
#ifdef GF_AB_NO_ALOAD
            return;
#endif
#pragma unroll
            for (int gq = 0; gq < QW; ++gq) a[gq] = *(const float4*)(Zc + 8 * (kq * QW + gq));
        };
        auto mma = [&](const float4 (&a)[QW], const float4 (&b)[QW][NB]) {
#pragma unroll
            for (int gq = 0; gq < QW; ++gq) {
                const float av[4] = {a[gq].x, a[gq].y, a[gq].z, a[gq].w};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const float bv = s == 0 ? b[gq][nb].x : (s == 1 ? b[gq][nb].y : (s == 2 ? b[gq][nb].z : b[gq][nb].w));
                        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv, acc[nb], 0, 0, 0);
                    }
                }
            }
        };
        auto first_node = [&](int g) __attribute__((always_inline)) { const int u = g % GF_UNITS; return u < 16 ? 4 * u : 64; };
        // a unit of NC nodes whose first B operand sits in buffer P0; next_node = first node of the following unit (-1: none)
        auto run_unit = [&](auto par, auto ncells, const float* Zb, int cell0, int next_node) __attribute__((always_inline)) {
            constexpr int P0 = decltype(par)::value, NC = decltype(ncells)::value;
            load_a(Zb, Af[P0]);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                constexpr int dummy = 0; (void)dummy;
                const int cur = (P0 + c) & 1, nx = cur ^ 1;
                const int nxt = c + 1 < NC ? cell0 + c + 1 : next_node;
                if (c + 1 < NC) load_a(Zb + (c + 1) * ZC, Af[nx]);
                if (nxt >= 0) load_b(nxt, Bf[nx]);
#ifndef GF_AB_NO_MFMA
                mma(Af[cur], Bf[cur]);
#endif
            }
        };
        int parity = 0;
        if (nun > 0) load_b(first_node(g0), Bf[0]);
        for (int p = 0; p <= nun; ++p) {
            if (p >= 1) {
                const int g = g0 + p - 1, tile = g / GF_UNITS, u = g - tile * GF_UNITS;
                const float* Zb = Z + ((p - 1) & 1) * ZB + m * PITCH + 4 * h;
                const int next_node = p < nun ? first_node(g + 1) : -1;
                if (u < 16) {
                    if (parity == 0) run_unit(std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{}, Zb, 4 * u, next_node);
                    else run_unit(std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{}, Zb, 4 * u, next_node);
                } else {
                    if (parity == 0) run_unit(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, Zb, 64, next_node);
                    else run_unit(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, Zb, 64, next_node);
                    parity ^= 1;
                }
                const bool last_of_tile = (p == nun) || ((g + 1) / GF_UNITS != tile);
                if (last_of_tile) {
                    // partial slab of (tile, this workgroup's segment, K-quarter): [col][row]; a lane's 4 consecutive
                    // accumulator registers are 4 consecutive rows of one column -> 16-byte stores
                    const int seg = w - gf_owner((long long)tile * GF_COST, a_nwg, a_ctot);
                    float* slab = a_scratch + ((size_t)(tile * a_maxseg + seg) * 4 + kq) * (COUTP * GF_TILE);
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                        for (int gr = 0; gr < 4; ++gr) {
                            const float4 v = make_float4(acc[nb][4 * gr], acc[nb][4 * gr + 1], acc[nb][4 * gr + 2], acc[nb][4 * gr + 3]);
                            *(float4*)(slab + (32 * nb + m) * GF_TILE + 8 * gr + 4 * h) = v;
                        }
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
                }
            }
            __syncthreads();
        }
        return;
      }
    }

    // ---------------------------------------------------------------------- producers
    // The producers run at a raised wave priority (round 4): on gfx950 the fp32 MFMA executes on the vector ALUs and holds them for its
    // 64 cycles, so next to a streaming consumer a producer gets ONE dependent vector instruction through per MFMA; with priority its
    // ready instructions at least go first whenever the ALUs free up.  conv1 81.8 -> 73.6 us, conv2 57.2 -> 55.4 us over a 30-step rollout
    // (the lists get longer as the body settles; 75.2 -> 73.8 / 56.9 -> 55.7 on the initial state), priority 1 vs 0; 3 vs 0 the same;
    // consumers above producers: no change from equal priorities.
#ifndef GF_AB_PRIO_P
#define GF_AB_PRIO_P 1
#endif
    __builtin_amdgcn_s_setprio(GF_AB_PRIO_P);
    // 8 producer waves = 2 per SIMD (next to a consumer wave): the gather is bound by the loads a wave keeps in flight times the L2
    // latency, and by the wave's issue rate — twice the waves double both (4 waves, 8 threads per point: 45 us for conv1's gather
    // alone, whatever was done to the instruction stream)
    const int pt = wave * 64 + lane, pp = pt / GF_TPP, t = pt % GF_TPP;      // (lane = threadIdx.x & 63 as for the consumers)
    auto has_quad = [&](int k) __attribute__((always_inline)) { return t + GF_TPP * k < CIN / 4; };
    auto load_roff = [&](int g, int& e0, int& e1) __attribute__((always_inline)) {
        const int tile = g / GF_UNITS, u = g - tile * GF_UNITS, i = tile * GF_TILE + pp;
        e0 = e1 = 0;
        if (u < 16 && i < a_n) {
            const uint16_t* r = a_roff + (size_t)i * GF_ROFF_PITCH + u;
            e0 = r[0]; e1 = r[1];
        }
    };
    // (plain scalars, not a struct: a struct that is copied and captured by reference stays in scratch memory)
    // (nine separate scalars per unit, not arrays: the slot of an entry is picked at run time, and an array that is indexed —
    // or selected from, which the optimiser turns back into indexing — at run time lives in scratch memory)
    auto load_one = [&](const uint32_t* eb, int e0, int cnt_, int e, int& jc, float& w0, float& w1) __attribute__((always_inline)) {
        jc = 0; w0 = 0.f; w1 = 0.f;
        if (e < cnt_) {
            const uint32_t* src = eb + 3 * (size_t)(e0 + e);
            jc = (int)src[0]; w0 = __uint_as_float(src[1]); w1 = __uint_as_float(src[2]);
        }
    };
#define GF_LOAD_ENTRIES(G, E0, E1, P)                                                                           \
    do {                                                                                                        \
        const int tl_ = (G) / GF_UNITS, ii_ = min(tl_ * GF_TILE + pp, a_n - 1);                                 \
        const uint32_t* eb_ = a_ent + (size_t)ii_ * (size_t)(4 * a_pitch) * 3;                                  \
        P##e0 = (E0); P##cnt = (E1) - (E0);                                                                     \
        _Pragma("unroll") for (int sl_ = 0; sl_ < GF_SLOTS; ++sl_)                                              \
            load_one(eb_, P##e0, P##cnt, GF_TPP * sl_ + t, P##j[sl_], P##a[sl_], P##b[sl_]);                     \
    } while (0)

    // (register arrays indexed by compile-time constants only)
    int cj[GF_SLOTS], nj[GF_SLOTS];
    float ca[GF_SLOTS], cb[GF_SLOTS], na[GF_SLOTS], nb[GF_SLOTS];       // a = w(cx), b = w(cx + 1)
#pragma unroll
    for (int q = 0; q < GF_SLOTS; ++q) { cj[q] = nj[q] = 0; ca[q] = cb[q] = na[q] = nb[q] = 0.f; }
    int ccnt = 0, ce0 = 0, ncnt = 0, ne0 = 0;
    int r1a = 0, r1b = 0, r2a = 0, r2b = 0;
    if (nun > 0) {
        int a, b;
        load_roff(g0, a, b);
        GF_LOAD_ENTRIES(g0, a, b, c);
        if (nun > 1) load_roff(g0 + 1, r1a, r1b);
    }
    for (int p = 0; p <= nun; ++p) {
        if (p < nun) {
            const int g = g0 + p, tile = g / GF_UNITS, u = g - tile * GF_UNITS;
            const int i = tile * GF_TILE + pp;
            // requests for the units ahead go out first: they land while this unit is built
            if (p + 2 < nun) load_roff(g + 2, r2a, r2b);
            if (p + 1 < nun) GF_LOAD_ENTRIES(g + 1, r1a, r1b, n); else ncnt = ne0 = 0;
            float* Zb = Z + (p & 1) * ZB + pp * PITCH;
            _Float16* Zbh = Zh16 + (p & 1) * ZBH + pp * PH;
            // a finished quad of Z: fp32 as it is, or split into hi = fp16(z) and lo = fp16(z - hi)
            auto put_z = [&](int c, int k, float zx, float zy, float zz, float zw) __attribute__((always_inline)) {
                if (!has_quad(k)) return;
                if constexpr (SPLIT) {
                    const f32x4v v = {zx, zy, zz, zw};
                    const h4 hi = __builtin_convertvector(v, h4);
                    const h4 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x4v), h4);
                    *(h4*)(Zbh + c * ZCH + 4 * (t + GF_TPP * k)) = hi;
                    *(h4*)(Zbh + ZLO + c * ZCH + 4 * (t + GF_TPP * k)) = lo;
                } else {
                    *(float4*)(Zb + c * ZC + 4 * (t + GF_TPP * k)) = make_float4(zx, zy, zz, zw);
                }
            };
            if (u == 16) {
                // Linear branch: Z = relu(x_i)
#pragma unroll
                for (int k = 0; k < NQ; ++k) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (i < a_n && has_quad(k)) v = *(const float4*)(a_x + (size_t)i * CIN + 4 * (t + GF_TPP * k));
                    if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    put_z(0, k, v.x, v.y, v.z, v.w);
                }
            } else {
                struct Acc4 { f32x2 lo, hi; };
                Acc4 acc[4][NQ];
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int k = 0; k < NQ; ++k) { acc[c][k].lo = f32x2{0.f, 0.f}; acc[c][k].hi = f32x2{0.f, 0.f}; }
                const int cnt = ccnt;
                const int isafe = min(i, a_n - 1);
                // one entry: gather relu(x[j]) and add it into the two touched nodes (weights of the other two are zero)
                auto add_entry = [&](int jc, float w0, float w1, const float4 (&xv)[NQ]) __attribute__((always_inline)) {
                    const int cx = (int)((unsigned)jc >> 30);
                    const float wc0 = cx == 0 ? w0 : 0.f;
                    const float wc1 = cx == 0 ? w1 : (cx == 1 ? w0 : 0.f);
                    const float wc2 = cx == 1 ? w1 : (cx == 2 ? w0 : 0.f);
                    const float wc3 = cx == 2 ? w1 : 0.f;
                    const f32x2 p0 = {wc0, wc0}, p1 = {wc1, wc1}, p2 = {wc2, wc2}, p3 = {wc3, wc3};
#pragma unroll
                    for (int k = 0; k < NQ; ++k) {
                        float4 v = xv[k];
                        if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                        // explicit (packed) FMAs: the library is built with -ffp-contract=off, which turns `acc += w * v` into a
                        // multiply AND an add — twice the issue slots of the wave that bounds this kernel
                        const f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
                        acc[0][k].lo = __builtin_elementwise_fma(p0, lo, acc[0][k].lo); acc[0][k].hi = __builtin_elementwise_fma(p0, hi, acc[0][k].hi);
                        acc[1][k].lo = __builtin_elementwise_fma(p1, lo, acc[1][k].lo); acc[1][k].hi = __builtin_elementwise_fma(p1, hi, acc[1][k].hi);
#ifndef GF_AB_HALF_FMA
                        acc[2][k].lo = __builtin_elementwise_fma(p2, lo, acc[2][k].lo); acc[2][k].hi = __builtin_elementwise_fma(p2, hi, acc[2][k].hi);
                        acc[3][k].lo = __builtin_elementwise_fma(p3, lo, acc[3][k].lo); acc[3][k].hi = __builtin_elementwise_fma(p3, hi, acc[3][k].hi);
#endif
                    }
                };
                struct XS { int jc; float w0, w1; float4 xv[NQ]; };
                struct TB { int jc; float w0, w1; };
                // Entry E (a compile-time index: the loop below is fully unrolled, so slot and source lane are constants — a
                // run-time slot choice among the 12 entry registers is turned into a table in scratch memory by the compiler).
                // The 8 threads of a point hold entries e0 + 8 s + t; bcast() fetches entry E's record from its owner inside the
                // point's 8 lanes (three ds_bpermute), ONE STEP before request() turns it into the x-row loads — the LDS round trip
                // hides behind the accumulation in between instead of stalling every request.
                auto bcast = [&](auto ec, TB& T) __attribute__((always_inline)) {
                    constexpr int E = decltype(ec)::value, SL = E / GF_TPP;
                    const int jcS = cj[SL];
                    const float w0S = ca[SL];
                    const float w1S = cb[SL];
#if GF_TPP == 8 && !defined(GF_AB_SHFL_BCAST)
                    // lane (E & 7) of every 8-lane group to the whole group on the DPP path: a quad_perm broadcast inside the source's
                    // quad, then a row shift by 4 into the group's other quad (bank mask) — two VALU moves per value instead of a
                    // ds_bpermute round trip through the LDS crossbar (three per entry, ~100 per unit and wave before)
                    auto oct = [&](int v) __attribute__((always_inline)) {
                        constexpr int K = E & 7, QP = (K & 3) * 0x55;
                        const int t = __builtin_amdgcn_update_dpp(v, v, QP, 0xf, 0xf, false);
                        return K < 4 ? __builtin_amdgcn_update_dpp(t, t, 0x114, 0xf, 0xa, false)       // row_shr:4 into quads 1, 3
                                     : __builtin_amdgcn_update_dpp(t, t, 0x104, 0xf, 0x5, false);      // row_shl:4 into quads 0, 2
                    };
                    const int jc = oct(jcS);
                    const float w0 = __int_as_float(oct(__float_as_int(w0S))), w1 = __int_as_float(oct(__float_as_int(w1S)));
#else
                    const int src = (lane & ~(GF_TPP - 1)) | (E & (GF_TPP - 1));
                    const int jc = __shfl(jcS, src, 64);
                    const float w0 = __shfl(w0S, src, 64), w1 = __shfl(w1S, src, 64);
#endif
                    const bool ok = E < cnt;
                    T.jc = ok ? jc : 0; T.w0 = ok ? w0 : 0.f; T.w1 = ok ? w1 : 0.f;      // masked: row 0, zero weights
                };
                auto request = [&](const TB& T, XS& S) __attribute__((always_inline)) {
                    S.jc = T.jc; S.w0 = T.w0; S.w1 = T.w1;
#ifdef GF_AB_LOCAL_J        /* timing-only (garbage results): every gather reads one of 64 rows — what the producers would cost if a tile's neighbour rows sat next to them */
                    const int j = (T.jc & 0x3fffffff) & 63;
#else
                    const int j = T.jc & 0x3fffffff;
#endif
#pragma unroll
                    for (int k = 0; k < NQ; ++k) {
                        S.xv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (has_quad(k)) S.xv[k] = *(const float4*)(a_x + (size_t)j * CIN + 4 * (t + GF_TPP * k));
                    }
                };
                int wmax = cnt;                                   // the longest list among the wave's 8 points
#pragma unroll
                for (int o = GF_TPP; o <= 32; o <<= 1) wmax = max(wmax, __shfl_xor(wmax, o, 64));
                // FOUR entries in flight: a slot is refilled (x-row request of the entry four ahead) right after its entry has
                // been accumulated.  Straight-line groups of 4 with early exits on a SCALAR trip count; inside a group nothing is
                // conditional (an entry past a point's own list is a masked request) — with conditional requests the compiler can
                // no longer count the loads in flight and waits for ALL of them before every use.
#ifdef GF_AB_NO_GATHER
                const int ne4 = 0;
#else
                const int ne4 = __builtin_amdgcn_readfirstlane((min(wmax, GF_SLOTS * GF_TPP) + 3) & ~3);
#endif
                XS S0, S1, S2, S3;
                TB T0, T1;
                if (ne4 > 0) {
                    bcast(std::integral_constant<int, 0>{}, T0); bcast(std::integral_constant<int, 1>{}, T1);
                    request(T0, S0); request(T1, S1);
                    bcast(std::integral_constant<int, 2>{}, T0); bcast(std::integral_constant<int, 3>{}, T1);
                    request(T0, S2); request(T1, S3);
                    bcast(std::integral_constant<int, 4>{}, T0);
                }
                // step E: accumulate entry E, re-use its slot for the request of entry E + 4 (record broadcast one step ago),
                // broadcast the record of entry E + 5
#define GF_STEP(E, S, TC, TN)                                                                                             \
                    add_entry(S.jc, S.w0, S.w1, S.xv);                                                                    \
                    if ((E) + 4 < GF_SLOTS * GF_TPP) request(TC, S);                                                           \
                    if ((E) + 5 < GF_SLOTS * GF_TPP) bcast(std::integral_constant<int, ((E) + 5) % (GF_SLOTS * GF_TPP)>{}, TN);
#define GF_GROUP(E)                                                                                                       \
                if (ne4 > (E)) {                                                                                          \
                    GF_STEP((E), S0, T0, T1) GF_STEP((E) + 1, S1, T1, T0) GF_STEP((E) + 2, S2, T0, T1) GF_STEP((E) + 3, S3, T1, T0)
                GF_GROUP(0) GF_GROUP(4) GF_GROUP(8) GF_GROUP(12) GF_GROUP(16) GF_GROUP(20) GF_GROUP(24) GF_GROUP(28)
                }}}}}}}}
#undef GF_GROUP
#undef GF_STEP
                if (__any(cnt > GF_SLOTS * GF_TPP)) {
                    // rows with more than 32 entries (rare): the tail straight from the list, one entry at a time
                    const uint32_t* eb = a_ent + (size_t)isafe * (size_t)(4 * a_pitch) * 3;
                    int emax = cnt;
#pragma unroll
                    for (int o = GF_TPP; o <= 32; o <<= 1) emax = max(emax, __shfl_xor(emax, o, 64));
                    for (int e = GF_SLOTS * GF_TPP; e < emax; ++e) {
                        const bool ok = e < cnt;
                        const uint32_t* src = eb + 3 * (size_t)(ce0 + (ok ? e : 0));
                        int jc = ok ? (int)src[0] : 0;
                        const float w0 = ok ? __uint_as_float(src[1]) : 0.f, w1 = ok ? __uint_as_float(src[2]) : 0.f;
                        const int j = ok ? (jc & 0x3fffffff) : isafe;
                        float4 xv[NQ];
#pragma unroll
                        for (int k = 0; k < NQ; ++k) {
                            xv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                            if (has_quad(k)) xv[k] = *(const float4*)(a_x + (size_t)j * CIN + 4 * (t + GF_TPP * k));
                        }
                        add_entry(jc, w0, w1, xv);
                    }
                }
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int k = 0; k < NQ; ++k) put_z(c, k, acc[c][k].lo.x, acc[c][k].lo.y, acc[c][k].hi.x, acc[c][k].hi.y);
            }
#pragma unroll
            for (int q = 0; q < GF_SLOTS; ++q) { cj[q] = nj[q]; ca[q] = na[q]; cb[q] = nb[q]; }
            ccnt = ncnt; ce0 = ne0;
            r1a = r2a; r1b = r2b;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// epilogue: y[i][co] = sum over the tile's partial slabs (segment-major, K-quarter-minor: a fixed order) + conv bias + Linear
// bias (+ residual); for the last layer (cout = 3) also pos_correction = y / 128 and update_pos_vel (models/transmodel.py:141-148)
// ------------------------------------------------------------------------------------------------
struct GfEpi {
    const float* scratch; int tiles, nwg, maxseg, ctot, coutp, cout, n, split;
    const float* bias_c; const float* bias_d; const float* residual; float* out;
    float* out_relu;          // optional: max(y, 0), what the next layer gathers (so that it need not apply the ReLU per gathered row)
    const float* pos; const float* pos_new; float* pos_c; float* vel_c; float scale, dt;       // pos == null: no update
};

// sum of a tile's partial slabs for output (row, col) + biases (+ residual): a fixed order (segment-major, K-quarter-minor)
__device__ __forceinline__ float gf_epi_value(const GfEpi& E, int tile, int nseg, int row, int col, int i)
{
    float v = 0.f;
    if (E.split) {
        // slabs [tile][segment][wave][32 cols][32 rows]; wave = K-part * NB + N-block: the K-parts of this column's N-block
        const int nb_ = E.coutp / 32, kparts = 4 / nb_, slab = 32 * GF_TILE;
        const float* s = E.scratch + (size_t)tile * E.maxseg * 4 * slab + (col & 31) * GF_TILE + row;
        for (int sg = 0; sg < nseg; ++sg)
            for (int kp = 0; kp < kparts; ++kp) v += s[(size_t)(sg * 4 + kp * nb_ + (col >> 5)) * slab];
    } else {
        const int nq = nseg * 4;
        const int slab = E.coutp * GF_TILE;
        const float* s = E.scratch + (size_t)tile * E.maxseg * 4 * slab + col * GF_TILE + row;
        int q = 0;
        for (; q + 4 <= nq; q += 4) {
            const float p0 = s[(size_t)q * slab], p1 = s[(size_t)(q + 1) * slab], p2 = s[(size_t)(q + 2) * slab], p3 = s[(size_t)(q + 3) * slab];
            v += p0; v += p1; v += p2; v += p3;
        }
        for (; q < nq; ++q) v += s[(size_t)q * slab];
    }
    v += E.bias_c[col] + E.bias_d[col];
    if (E.residual) v += E.residual[(size_t)i * E.cout + col];
    return v;
}

__global__ void __launch_bounds__(256) k_cconv_gf_epi(GfEpi E)
{
    // one output per thread: block = (tile, group of 8 columns); thread = (column, row): the slab reads of a wave are two
    // runs of 128 contiguous bytes, all of a thread's nseg * 4 loads are independent (one round trip)
    const int tile = blockIdx.x;
    const int col = blockIdx.y * 8 + (threadIdx.x >> 5), row = threadIdx.x & 31, i = tile * GF_TILE + row;
    if (col >= E.cout || i >= E.n) return;
    const int wfirst = gf_owner((long long)tile * GF_COST, E.nwg, E.ctot), wlast = gf_owner((long long)tile * GF_COST + 64, E.nwg, E.ctot);
    const int nseg = wlast - wfirst + 1;
    const float v = gf_epi_value(E, tile, nseg, row, col, i);
    if (E.out) E.out[(size_t)i * E.cout + col] = v;
    if (E.out_relu) E.out_relu[(size_t)i * E.cout + col] = fmaxf(v, 0.f);
    if (E.pos) {                                              // cout == 3: col = coordinate (k_trans_update's expressions)
        const size_t e = (size_t)i * 3 + col;
        const float pc = E.pos_new[e] + E.scale * v;
        E.pos_c[e] = pc;
        E.vel_c[e] = (pc - E.pos[e]) / E.dt;
    }
}

// ------------------------------------------------------------------------------------------------
// filter packing: wp[(((c * (CIN/8) + q) * NB + nb) * 64 + lane) * 4 + e] = W[c][ci = 8 q + 4 (lane >> 5) + e][co = 32 nb + (lane & 31)]
// with W[c] = kernel[c] (c < 64, kernel (4,4,4,CIN,COUT) = [z][y][x] nodes) and W[64][ci][co] = dense_w[co][ci]; co >= COUT -> 0
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_cconv_gf_pack(const float* __restrict__ kernel, const float* __restrict__ dense_w, int cin, int cout,
                                                       int nb_, float* __restrict__ wp)
{
    const size_t total = (size_t)65 * (cin / 8) * nb_ * 256;
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const int e = (int)(id & 3), lane = (int)((id >> 2) & 63);
    size_t r = id >> 8;
    const int nb = (int)(r % nb_); r /= nb_;
    const int q = (int)(r % (cin / 8)), c = (int)(r / (cin / 8));
    const int ci = 8 * q + 4 * (lane >> 5) + e, co = 32 * nb + (lane & 31);
    float v = 0.f;
    if (co < cout) v = c < 64 ? kernel[((size_t)c * cin + ci) * cout + co] : dense_w[(size_t)co * cin + ci];
    wp[id] = v;
}

// split packing: halves[((((c * KS + s) * NB + nb) * 2 + hl) * 64 + lane) * 8 + e] = hi / lo (hl = 0 / 1) of
// W[c][ci = 16 s + 8 (lane >> 5) + e][co = 32 nb + (lane & 31)]
__global__ void __launch_bounds__(256) k_cconv_gf_pack_split(const float* __restrict__ kernel, const float* __restrict__ dense_w, int cin,
                                                             int cout, int nb_, _Float16* __restrict__ wp)
{
    const int ks = cin / 16;
    const size_t total = (size_t)65 * ks * nb_ * 2 * 512;
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const int e = (int)(id & 7), lane = (int)((id >> 3) & 63), hl = (int)((id >> 9) & 1);
    size_t r = id >> 10;
    const int nb = (int)(r % nb_); r /= nb_;
    const int sidx = (int)(r % ks), c = (int)(r / ks);
    const int ci = 16 * sidx + 8 * (lane >> 5) + e, co = 32 * nb + (lane & 31);
    float v = 0.f;
    if (co < cout) v = c < 64 ? kernel[((size_t)c * cin + ci) * cout + co] : dense_w[(size_t)co * cin + ci];
    const _Float16 hi = (_Float16)v;
    wp[id] = hl ? (_Float16)(v - (float)hi) : hi;
}

extern "C" size_t nf_cconv_gf_packed_split_bytes(int cin, int cout)
{
    if (cin <= 0 || cin % 32 || cout <= 0 || cout > 64) return 0;
    const int nb = cout > 32 ? 2 : 1;
    return (size_t)65 * (cin / 16) * nb * 2 * 512 * sizeof(_Float16);
}

extern "C" int nf_cconv_gf_pack_split(const float* kernel, const float* dense_w, int cin, int cout, void* packed, nf_stream_t stream)
{
    NF_CHECK_ARG(kernel && dense_w && packed, "null pointer");
    const size_t total = nf_cconv_gf_packed_split_bytes(cin, cout) / sizeof(_Float16);
    NF_CHECK_ARG(total > 0, "cin must be a multiple of 32, cout in [1, 64]");
    hipLaunchKernelGGL(k_cconv_gf_pack_split, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, kernel, dense_w,
                       cin, cout, cout > 32 ? 2 : 1, (_Float16*)packed);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" size_t nf_cconv_gf_packed_floats(int cin, int cout)
{
    if (cin <= 0 || cin % 32 || cout <= 0 || cout > 64) return 0;
    const int nb = cout > 32 ? 2 : 1;
    return (size_t)65 * (cin / 8) * nb * 256;
}

extern "C" int nf_cconv_gf_pack(const float* kernel, const float* dense_w, int cin, int cout, float* packed, nf_stream_t stream)
{
    NF_CHECK_ARG(kernel && dense_w && packed, "null pointer");
    const size_t total = nf_cconv_gf_packed_floats(cin, cout);
    NF_CHECK_ARG(total > 0, "cin must be a multiple of 32, cout in [1, 64]");
    hipLaunchKernelGGL(k_cconv_gf_pack, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, kernel, dense_w, cin, cout,
                       cout > 32 ? 2 : 1, packed);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// plan of a launch: workgroups (<= CUs, every one with at least two units), segments per tile, scratch size
extern "C" int nf_cconv_gf_plan(int n, int cout, int max_wg, int* tiles, int* nwg, int* maxseg, size_t* scratch_floats)
{
    NF_CHECK_ARG(n > 0 && cout >= 1 && cout <= 64 && max_wg >= 1, "bad arguments");
    const int T = (n + GF_TILE - 1) / GF_TILE, ctot = T * GF_COST;
    int w = ctot / 8;                                   // >= 8 filter nodes per workgroup: no workgroup without a unit
    if (w > max_wg) w = max_wg;
    if (w < 1) w = 1;
    int ms = 1;
    for (int t = 0; t < T; ++t) {
        const int s = gf_owner((long long)t * GF_COST + 64, w, ctot) - gf_owner((long long)t * GF_COST, w, ctot) + 1;
        if (s > ms) ms = s;
    }
    if (tiles) *tiles = T;
    if (nwg) *nwg = w;
    if (maxseg) *maxseg = ms;
    if (scratch_floats) *scratch_floats = (size_t)T * ms * 4 * (cout > 32 ? 64 : 32) * GF_TILE;
    return NF_OK;
}

template <int CIN, int NB, bool RELU, bool SPLIT>
static int gf_launch(const GfArgs& a, hipStream_t st)
{
    const size_t lds = SPLIT ? (size_t)2 * 2 * 4 * GF_TILE * (CIN + 8) * sizeof(_Float16) : (size_t)2 * 4 * GF_TILE * (CIN + 4) * sizeof(float);
    static bool attr_set[64] = {};
    if (nf_first_use_on_device(attr_set))
        hipFuncSetAttribute((const void*)k_cconv_gf<CIN, NB, RELU, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_cconv_gf<CIN, NB, RELU, SPLIT>), dim3(a.nwg), dim3(GF_THREADS), lds, st, a);
    return 0;
}

// One G-free layer: y = cconv(act(x)) + Linear(act(x)) + biases (+ residual) [+ position / velocity update when pos != NULL]
// the contraction kernel of one layer into the partial slabs; fills the epilogue's description of them
static int gf_run_conv(const float* x, int n, int cin, int cout, int relu, const uint16_t* roff, const uint32_t* ent, int pitch,
                       const void* packed, int split, float* scratch, int max_wg, hipStream_t st, GfEpi* e,
                       int32_t* done_word = nullptr, int step_id = 0)
{
    GfArgs a;
    a.done_word = (volatile int*)done_word; a.step_id = step_id;
    a.x = x; a.n = n; a.relu = relu; a.roff = roff; a.ent = ent; a.pitch = pitch; a.wp = (const float*)packed; a.scratch = scratch;
    size_t sf;
    if (nf_cconv_gf_plan(n, cout, max_wg, &a.tiles, &a.nwg, &a.maxseg, &sf) != NF_OK) return NF_EINVAL;
    a.ctot = a.tiles * GF_COST;
    const int nb = cout > 32 ? 2 : 1;
    // (the ReLU-on-load variants serve callers that hand over pre-activation features; nf_trans_step stores activated arrays)
#define GF_PICK(R, S)                                           \
    do {                                                        \
        if (cin == 96 && nb == 2) gf_launch<96, 2, R, S>(a, st); \
        else if (cin == 64 && nb == 2) gf_launch<64, 2, R, S>(a, st); \
        else gf_launch<64, 1, R, S>(a, st);                     \
    } while (0)
    if (split) { if (relu) GF_PICK(true, true); else GF_PICK(false, true); }
    else { if (relu) GF_PICK(true, false); else GF_PICK(false, false); }
#undef GF_PICK
    NF_CHECK_LAUNCH();
    e->split = split;
    e->scratch = scratch; e->tiles = a.tiles; e->nwg = a.nwg; e->maxseg = a.maxseg; e->ctot = a.ctot; e->coutp = 32 * nb; e->cout = cout; e->n = n;
    return NF_OK;
}

static int gf_layer(const float* x, int n, int cin, int cout, int relu, const uint16_t* roff, const uint32_t* ent,
                    int pitch, const void* packed, int split, const float* bias_conv, const float* bias_dense,
                    const float* residual, float* out, float* out_relu, float* scratch, int max_wg, const float* pos,
                    const float* pos_new, float scale, float dt, float* pos_c, float* vel_c, nf_stream_t stream, int32_t* done_word, int step_id);

extern "C" int nf_cconv_gf_layer(const float* x, int n, int cin, int cout, int relu, const uint16_t* roff, const uint32_t* ent,
                                 int pitch, const void* packed, int split, const float* bias_conv, const float* bias_dense,
                                 const float* residual, float* out, float* out_relu, float* scratch, int max_wg, const float* pos,
                                 const float* pos_new, float scale, float dt, float* pos_c, float* vel_c, nf_stream_t stream)
{
    return gf_layer(x, n, cin, cout, relu, roff, ent, pitch, packed, split, bias_conv, bias_dense, residual, out, out_relu, scratch, max_wg, pos,
                    pos_new, scale, dt, pos_c, vel_c, stream, nullptr, 0);
}

static int gf_layer(const float* x, int n, int cin, int cout, int relu, const uint16_t* roff, const uint32_t* ent,
                    int pitch, const void* packed, int split, const float* bias_conv, const float* bias_dense,
                    const float* residual, float* out, float* out_relu, float* scratch, int max_wg, const float* pos,
                    const float* pos_new, float scale, float dt, float* pos_c, float* vel_c, nf_stream_t stream, int32_t* done_word, int step_id)
{
    NF_CHECK_ARG(x && roff && ent && packed && bias_conv && bias_dense && (out || out_relu) && scratch, "null pointer");
    NF_CHECK_ARG(((cin == 96 && cout > 32) || cin == 64) && cout >= 1 && cout <= 64,
                 "supported shapes: 96 -> 33..64, 64 -> 1..64 channels (the transition model's layers)");
    NF_CHECK_ARG(!pos || (cout == 3 && pos_new && pos_c && vel_c), "the update epilogue belongs to the 3-channel layer");
    if (n <= 0) return NF_OK;
    hipStream_t st = (hipStream_t)stream;
    GfEpi e;
    const int rc = gf_run_conv(x, n, cin, cout, relu, roff, ent, pitch, packed, split, scratch, max_wg, st, &e, done_word, step_id);
    if (rc != NF_OK) return rc;
    e.bias_c = bias_conv; e.bias_d = bias_dense; e.residual = residual; e.out = out; e.out_relu = out_relu;
    e.pos = pos; e.pos_new = pos_new; e.pos_c = pos_c; e.vel_c = vel_c; e.scale = scale; e.dt = dt;
    hipLaunchKernelGGL(k_cconv_gf_epi, dim3(e.tiles, (cout + 7) / 8), dim3(256), 0, st, e);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// ------------------------------------------------------------------------------------------------
// The last layer (conv3 + dense3: 64 -> 3 channels, models/transmodel.py:121-131 at i = 3, then :141-148).  Its contraction is
// 2 % of the step's FLOPs; on the tile machinery above it would pad 3 output channels to a 32-wide MFMA block and pay the full
// gather.  Here "transform, then gather" IS the right order, because the transformed array is tiny:
//   G3[j][node][co] = sum_ci relu(a2)[j][ci] * K3[node][ci][co]       (65 x 3 floats = 780 bytes per particle, 3.8 MB in all;
//                                                                      node 64 = the Linear branch)
//   y3[i][co] = sum over the row entries of i of  w(cx) G3[j][4 rho + cx][co] + w(cx + 1) G3[j][4 rho + cx + 1][co]
//               + G3[i][64][co] + biases;      pos_correction = y3 / 128, update_pos_vel.
// ------------------------------------------------------------------------------------------------
#define G3_PITCH 196            // floats per particle (65 x 3 = 195, padded)
#define G3_TILE 8               // (rounds 2-3: particles per workgroup of the VALU transform; round 4: 32-particle tiles on the matrix pipe)

// filter of the last layer as the transform reads it: kt[ci][o], o = node * 3 + co (195 columns, pitch G3_PITCH; node 64 = dense3) —
// a thread's 64 weights are then 64 coalesced loads across the workgroup instead of 64 loads strided by 768 bytes
__global__ void __launch_bounds__(256) k_cconv3_pack(const float* __restrict__ kernel /* (64, 64, 3) */, const float* __restrict__ dense_w /* (3, 64) */,
                                                     float* __restrict__ kt)
{
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= 64 * G3_PITCH) return;
    const int ci = id / G3_PITCH, o = id - ci * G3_PITCH;
    float v = 0.f;
    if (o < 195) { const int node = o / 3, co = o - 3 * node; v = node < 64 ? kernel[((size_t)node * 64 + ci) * 3 + co] : dense_w[co * 64 + ci]; }
    kt[id] = v;
}

extern "C" size_t nf_cconv3_packed_floats(void) { return (size_t)64 * G3_PITCH; }

extern "C" int nf_cconv3_pack(const float* kernel, const float* dense_w, float* packed, nf_stream_t stream)
{
    NF_CHECK_ARG(kernel && dense_w && packed, "null pointer");
    hipLaunchKernelGGL(k_cconv3_pack, dim3((64 * G3_PITCH + 255) / 256), dim3(256), 0, (hipStream_t)stream, kernel, dense_w, packed);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// G3[32 x 195] = xs[32 x 64] . kt[64 x 195] of one tile on the matrix pipe (round 4): 7 column blocks of 32, one per wave, 32 K-steps of
// v_mfma_f32_32x32x2_f32 each — 224 MFMAs per tile instead of ~700 vector instructions per thread.  An element's sum runs over
// k = 0..63 in ONE chain (exact fp32: fmaf per product).  xs: LDS tile with an odd pitch (the A fragment reads 32 rows at one k).
#define G3_XS_PITCH 65
__device__ __forceinline__ void g3_transform_tile(const float* __restrict__ xs, const float* __restrict__ kt, float* __restrict__ G3,
                                                  int i0, int n, int wave, int lane)
{
    if (wave >= 7) return;
    const int c = lane & 31, kh = lane >> 5;
    const int ncol = 32 * wave + c;
    const bool col_ok = ncol < 195;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bw[32];
#pragma unroll
    for (int st = 0; st < 32; ++st) bw[st] = col_ok ? kt[(2 * st + kh) * G3_PITCH + ncol] : 0.f;
#pragma unroll
    for (int st = 0; st < 32; ++st) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[c * G3_XS_PITCH + 2 * st + kh], bw[st], acc, 0, 0, 0);
    if (col_ok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (i0 + row < n) G3[(size_t)(i0 + row) * G3_PITCH + ncol] = acc[r];
        }
    }
}

// stand-alone transform (nf_cconv3_layer): a 32-particle tile per workgroup of 8 waves (7 column blocks)
__global__ void __launch_bounds__(512) k_cconv3_transform(const float* __restrict__ xr /* n x 64, activated */, int n,
                                                          const float* __restrict__ kt /* nf_cconv3_pack */, float* __restrict__ G3)
{
    __shared__ float xs[GF_TILE * G3_XS_PITCH];
    const int i0 = blockIdx.x * GF_TILE;
    for (int t = threadIdx.x; t < GF_TILE * 16; t += 512) {
        const int r = t >> 4, q = t & 15;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i0 + r < n) v = *(const float4*)(xr + (size_t)(i0 + r) * 64 + 4 * q);
        float* d = xs + r * G3_XS_PITCH + 4 * q;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    g3_transform_tile(xs, kt, G3, i0, n, threadIdx.x >> 6, threadIdx.x & 63);
}

// conv2's epilogue and conv3's transform in one kernel (a tile of 32 particles per workgroup): the transform of a particle needs
// only that particle's own 64 activated channels, which the epilogue has just summed — they go through LDS instead of through
// a2 in global memory and a launch of their own.  Same expressions as k_cconv_gf_epi and k_cconv3_transform: identical G3.
__global__ void __launch_bounds__(1024) k_cconv_gf_epi_g3(GfEpi E, const float* __restrict__ kt /* nf_cconv3_pack */, float* __restrict__ G3)
{
    // 1 024 threads per 32-particle tile: the epilogue reads the slabs in whole 128-byte rows (thread = row, column group); the
    // transform then runs on the matrix pipe (g3_transform_tile: 12.6 -> 9.4 us for the kernel; rounds 2-3 ran it on the vector
    // ALUs as four 256-thread groups of 8 particles).  (256 threads per tile: 26 us, 154 workgroups walking 32 rows each; 8-row
    // workgroups: 16 us, the slab reads fall apart into 32-byte pieces.)
    __shared__ float xs[GF_TILE * G3_XS_PITCH];
    const int tile = blockIdx.x, i0 = tile * GF_TILE;
    const int wfirst = gf_owner((long long)tile * GF_COST, E.nwg, E.ctot), wlast = gf_owner((long long)tile * GF_COST + 64, E.nwg, E.ctot);
    const int nseg = wlast - wfirst + 1;
    {
        const int row = threadIdx.x & 31, i = i0 + row;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int col = 32 * k + (threadIdx.x >> 5);
            float v = 0.f;
            if (i < E.n) {
                v = gf_epi_value(E, tile, nseg, row, col, i);
                if (E.out) E.out[(size_t)i * 64 + col] = v;
                if (E.out_relu) E.out_relu[(size_t)i * 64 + col] = fmaxf(v, 0.f);
            }
            xs[row * G3_XS_PITCH + col] = fmaxf(v, 0.f);
        }
    }
    __syncthreads();
    g3_transform_tile(xs, kt, G3, i0, E.n, threadIdx.x >> 6, threadIdx.x & 63);      // same function as k_cconv3_transform: identical G3
}

struct G3Epi { const float* pos; const float* pos_new; float* pos_c; float* vel_c; float scale, dt; };

__global__ void __launch_bounds__(256) k_cconv3_gather(const float* __restrict__ G3, int n, const uint16_t* __restrict__ roff,
                                                       const uint32_t* __restrict__ ent, int pitch, const float* __restrict__ bias_c,
                                                       const float* __restrict__ bias_d, float* __restrict__ y3, G3Epi E)
{
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    // the 17 row offsets of this particle: lane r holds roff[r]; an entry's row = number of offsets (1..15) it has reached
    const int myoff = lane < 17 ? (int)roff[(size_t)i * GF_ROFF_PITCH + lane] : 0;
    const int total = __builtin_amdgcn_readlane(myoff, 16);
    int bound[15];
#pragma unroll
    for (int r = 0; r < 15; ++r) bound[r] = __builtin_amdgcn_readlane(myoff, r + 1);
    const uint32_t* eb = ent + (size_t)i * (size_t)(4 * pitch) * 3;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
    for (int e = lane; e < total; e += 64) {
        const uint32_t jc = eb[3 * (size_t)e];
        const float w0 = __uint_as_float(eb[3 * (size_t)e + 1]), w1 = __uint_as_float(eb[3 * (size_t)e + 2]);
        int rho = 0;
#pragma unroll
        for (int r = 0; r < 15; ++r) rho += e >= bound[r] ? 1 : 0;
        const int j = (int)(jc & 0x3fffffffu), cx = (int)(jc >> 30);
        const float* g = G3 + (size_t)j * G3_PITCH + (4 * rho + cx) * 3;      // nodes cx and cx + 1: six consecutive floats
        acc0 = fmaf(w1, g[3], fmaf(w0, g[0], acc0));
        acc1 = fmaf(w1, g[4], fmaf(w0, g[1], acc1));
        acc2 = fmaf(w1, g[5], fmaf(w0, g[2], acc2));
    }
    acc0 = nf_wave_sum(acc0); acc1 = nf_wave_sum(acc1); acc2 = nf_wave_sum(acc2);
    if (lane < 3) {
        const float a = lane == 0 ? acc0 : (lane == 1 ? acc1 : acc2);
        const float v = a + G3[(size_t)i * G3_PITCH + 192 + lane] + bias_c[lane] + bias_d[lane];
        y3[(size_t)i * 3 + lane] = v;
        if (E.pos) {                              // k_trans_update's expressions
            const size_t q = (size_t)i * 3 + lane;
            const float pc = E.pos_new[q] + E.scale * v;
            E.pos_c[q] = pc;
            E.vel_c[q] = (pc - E.pos[q]) / E.dt;
        }
    }
}

extern "C" size_t nf_cconv3_workspace_floats(int n) { return (size_t)(n > 0 ? n : 0) * G3_PITCH; }

extern "C" int nf_cconv3_layer(const float* x_act, int n, const uint16_t* roff, const uint32_t* ent, int pitch, const float* packed,
                               const float* bias_conv, const float* bias_dense, float* workspace, float* y3,
                               const float* pos, const float* pos_new, float scale, float dt, float* pos_c, float* vel_c,
                               nf_stream_t stream)
{
    NF_CHECK_ARG(x_act && roff && ent && packed && bias_conv && bias_dense && workspace && y3, "null pointer");
    NF_CHECK_ARG(!pos || (pos_new && pos_c && vel_c), "the update needs pos_new / pos_c / vel_c");
    if (n <= 0) return NF_OK;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_cconv3_transform, dim3((n + GF_TILE - 1) / GF_TILE), dim3(512), 0, st, x_act, n, packed, workspace);
    G3Epi E;
    E.pos = pos; E.pos_new = pos_new; E.pos_c = pos_c; E.vel_c = vel_c; E.scale = scale; E.dt = dt;
    hipLaunchKernelGGL(k_cconv3_gather, dim3((n + 3) / 4), dim3(256), 0, st, (const float*)workspace, n, roff, ent, pitch, bias_conv,
                       bias_dense, y3, E);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_cconv_gf_layer_g3(const float* x, int n, int cin, int relu, const uint16_t* roff, const uint32_t* ent, int pitch,
                                    const void* packed, int split, const float* bias_conv, const float* bias_dense,
                                    const float* residual, float* out, float* out_relu, float* scratch, int max_wg,
                                    const float* packed3, float* g3, nf_stream_t stream)
{
    NF_CHECK_ARG(x && roff && ent && packed && bias_conv && bias_dense && scratch && packed3 && g3, "null pointer");
    NF_CHECK_ARG(cin == 96 || cin == 64, "supported shapes: 96 -> 64, 64 -> 64 channels");
    if (n <= 0) return NF_OK;
    hipStream_t st = (hipStream_t)stream;
    GfEpi e;
    const int rc = gf_run_conv(x, n, cin, 64, relu, roff, ent, pitch, packed, split, scratch, max_wg, st, &e);
    if (rc != NF_OK) return rc;
    e.bias_c = bias_conv; e.bias_d = bias_dense; e.residual = residual; e.out = out; e.out_relu = out_relu;
    e.pos = nullptr; e.pos_new = nullptr; e.pos_c = nullptr; e.vel_c = nullptr; e.scale = 0.f; e.dt = 0.f;
    hipLaunchKernelGGL(k_cconv_gf_epi_g3, dim3(e.tiles), dim3(1024), 0, st, e, packed3, g3);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_cconv3_gather(const float* g3, int n, const uint16_t* roff, const uint32_t* ent, int pitch, const float* bias_conv,
                                const float* bias_dense, float* y3, const float* pos, const float* pos_new, float scale, float dt,
                                float* pos_c, float* vel_c, nf_stream_t stream)
{
    NF_CHECK_ARG(g3 && roff && ent && bias_conv && bias_dense && y3, "null pointer");
    NF_CHECK_ARG(!pos || (pos_new && pos_c && vel_c), "the update needs pos_new / pos_c / vel_c");
    if (n <= 0) return NF_OK;
    G3Epi E;
    E.pos = pos; E.pos_new = pos_new; E.pos_c = pos_c; E.vel_c = vel_c; E.scale = scale; E.dt = dt;
    hipLaunchKernelGGL(k_cconv3_gather, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, g3, n, roff, ent, pitch, bias_conv, bias_dense,
                       y3, E);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// ------------------------------------------------------------------------------------------------
// The whole inference step behind ONE call (models/transmodel.py:151-163): prepare (integrate + fluid grid) -> front (search,
// row-entry lists, layer 0) -> conv1 (+ epilogue) -> conv2 (+ epilogue fused with conv3's transform) -> conv3's gather + update.
// 7 launches, no host round trip inside; the host reads
// overflow2 (largest neighbour count above its pitch, or 0) to decide whether the step has to be redone on the exact path.
// ------------------------------------------------------------------------------------------------
extern "C" int nf_trans_step(const nf_trans_step_t* s, const float* pos, const float* vel, float* num_nbrs, float* pos_c,
                             float* vel_c, int32_t* host_flag3, int step_id, nf_stream_t stream)
{
    NF_CHECK_ARG(s && pos && vel && num_nbrs && pos_c && vel_c, "null pointer");
    // stage 1 (grid build beside the container half) + the fluid half of the front (nf_trans.hip)
    int rc = nf_trans_stage12(s, pos, vel, num_nbrs, host_flag3, step_id, stream);
    if (rc != NF_OK) return rc;
    // every layer reads relu(previous layer) (models/transmodel.py:124): the producers of a0 / a1 / a2 store the activated
    // values (a0r, a1r, a2r), a1 itself is kept for conv2's residual
    // (the completion word the host spins on is written by the FIRST layer's launch as it starts: the front of the step — both searches
    // and their overflow words — is complete then; a last-workgroup protocol at the end of the front kernel cost that kernel ~2 us)
    rc = gf_layer(s->a0, s->n, 96, 64, 0, s->roff, s->ent, s->pitch_f, s->wp1, s->split, s->bc1, s->bd1, nullptr, s->a1, s->a1r, s->scratch,
                  s->max_wg, nullptr, nullptr, 0.f, 0.f, nullptr, nullptr, stream, host_flag3 ? host_flag3 + 2 : nullptr, step_id);
    if (rc != NF_OK) return rc;
    // conv2's output is read by conv3 only (a 64 -> 3 layer has no residual): it never goes to global memory
    rc = nf_cconv_gf_layer_g3(s->a1r, s->n, 64, 0, s->roff, s->ent, s->pitch_f, s->wp2, s->split, s->bc2, s->bd2, s->a1, nullptr, nullptr,
                              s->scratch, s->max_wg, s->wp3, s->g3, stream);
    if (rc != NF_OK) return rc;
    return nf_cconv3_gather(s->g3, s->n, s->roff, s->ent, s->pitch_f, s->bc3, s->bd3, s->y3, pos, s->pos_new, s->scale, s->dt, pos_c, vel_c,
                            stream);
}
