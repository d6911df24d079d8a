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

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GF_TILE 32
#define GF_UNITS 17            // 16 rows of 4 filter nodes + the Linear branch
#define GF_COST 65             // filter nodes per tile, Linear branch included
#define GF_THREADS 512
#define GF_ROFF_PITCH 20       // uint16 per particle (17 used)

struct GfArgs {
    const float* x;            // (n x CIN) features of the previous layer (ReLU applied on load when relu)
    int n, relu;
    const uint16_t* roff;      // [n][GF_ROFF_PITCH] row offsets into the particle's entry list
    const uint32_t* ent;       // [n][4 * pitch][3] row entries {j | cx << 30, w(cx), w(cx + 1)}
    int pitch;                 // pairs per particle
    const float* wp;           // packed filter (nf_cconv_gf_pack)
    float* scratch;            // [tiles][maxseg][4][COUTP][32] partial accumulators
    int tiles, nwg, maxseg, ctot;
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

// broadcast inside groups of 8 lanes: every lane reads lane (lane & ~7) | K
template <int K>
__device__ __forceinline__ int gf_bcast8(int v) { return __builtin_amdgcn_ds_swizzle(v, 0x18 | (K << 5)); }

template <int CIN, int NB>
__global__ void __launch_bounds__(GF_THREADS) k_cconv_gf(GfArgs A)
{
    constexpr int PITCH = CIN + 4;             // floats per Z row: (CIN + 4) / 4 odd -> ds_read_b128 of 16 rows conflict-free
    constexpr int ZC = GF_TILE * PITCH;        // one filter node
    constexpr int ZB = 4 * ZC;                 // one buffer (4 nodes)
    constexpr int NQ = CIN / 32;               // 16-byte quads of an x row per producer thread (8 threads per point)
    constexpr int QW = CIN / 32;               // quad-groups (4 K-steps each) per consumer wave and node
    constexpr int COUTP = 32 * NB;
    extern __shared__ float Z[];               // 2 * ZB floats

    const int w = blockIdx.x;
    const int g0 = gf_begin(w, A.nwg, A.ctot), g1 = gf_begin(w + 1, A.nwg, A.ctot);
    const int nun = g1 - g0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;

    if (wave < 4) {
        // ------------------------------------------------------------------ consumers
        const int kq = wave, m = lane & 31, h = lane >> 5;
        f32x16 acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
        const float4* wp4 = (const float4*)A.wp;
        for (int p = 0; p <= nun; ++p) {
            if (p >= 1) {
                const int g = g0 + p - 1, tile = g / GF_UNITS, u = g - tile * GF_UNITS;
                const int ncell = u < 16 ? 4 : 1, cell0 = u < 16 ? 4 * u : 64;
                const float* Zb = Z + ((p - 1) & 1) * ZB + m * PITCH + 4 * h;
                for (int c = 0; c < ncell; ++c) {
                    float4 a[QW], b[QW][NB];
#pragma unroll
                    for (int gq = 0; gq < QW; ++gq) {
                        const int q = kq * QW + gq;
                        a[gq] = *(const float4*)(Zb + c * ZC + 8 * q);
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            b[gq][nb] = wp4[((size_t)((cell0 + c) * (CIN / 8) + q) * NB + nb) * 64 + lane];
                    }
#pragma unroll
                    for (int gq = 0; gq < QW; ++gq) {
                        const float av[4] = {a[gq].x, a[gq].y, a[gq].z, a[gq].w};
#pragma unroll
                        for (int s = 0; s < 4; ++s)
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb) {
                                const float bv = s == 0 ? b[gq][nb].x : (s == 1 ? b[gq][nb].y : (s == 2 ? b[gq][nb].z : b[gq][nb].w));
                                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv, acc[nb], 0, 0, 0);
                            }
                    }
                }
                const bool last_of_tile = (p == nun) || ((g + 1) / GF_UNITS != tile);
                if (last_of_tile) {
                    // partial slab of (tile, this workgroup's segment, K-quarter): [col][row]; a lane's 4 consecutive
                    // accumulator registers are 4 consecutive rows of one column -> 16-byte stores
                    const int seg = w - gf_owner((long long)tile * GF_COST, A.nwg, A.ctot);
                    float* slab = A.scratch + ((size_t)(tile * A.maxseg + seg) * 4 + kq) * (COUTP * GF_TILE);
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

    // ---------------------------------------------------------------------- producers
    const int pt = threadIdx.x - 256, pp = pt >> 3, t = pt & 7;
    struct Ent { int jc[3]; float w0[3], w1[3]; int e0, cnt; };
    auto load_roff = [&](int g, int& e0, int& e1) {
        const int tile = g / GF_UNITS, u = g - tile * GF_UNITS, i = tile * GF_TILE + pp;
        e0 = e1 = 0;
        if (u < 16 && i < A.n) {
            const uint16_t* r = A.roff + (size_t)i * GF_ROFF_PITCH + u;
            e0 = r[0]; e1 = r[1];
        }
    };
    auto load_entries = [&](int g, int e0, int e1, Ent& E) {
        const int tile = g / GF_UNITS, i = min(tile * GF_TILE + pp, A.n - 1);
        const uint32_t* eb = A.ent + (size_t)i * (size_t)(4 * A.pitch) * 3;
        E.e0 = e0; E.cnt = e1 - e0;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int e = 8 * s + t;
            E.jc[s] = 0; E.w0[s] = 0.f; E.w1[s] = 0.f;
            if (e < E.cnt) {
                const uint32_t* src = eb + 3 * (size_t)(e0 + e);
                E.jc[s] = (int)src[0]; E.w0[s] = __uint_as_float(src[1]); E.w1[s] = __uint_as_float(src[2]);
            }
        }
    };

    Ent Ecur, Enext;
    int r1a = 0, r1b = 0, r2a = 0, r2b = 0;
    Ecur.cnt = Ecur.e0 = 0;
    if (nun > 0) {
        int a, b;
        load_roff(g0, a, b);
        load_entries(g0, a, b, Ecur);
        if (nun > 1) load_roff(g0 + 1, r1a, r1b);
    }
    for (int p = 0; p <= nun; ++p) {
        if (p < nun) {
            const int g = g0 + p, tile = g / GF_UNITS, u = g - tile * GF_UNITS;
            const int i = tile * GF_TILE + pp;
            // requests for the units ahead go out first: they land while this unit is built
            if (p + 2 < nun) load_roff(g + 2, r2a, r2b);
            if (p + 1 < nun) load_entries(g + 1, r1a, r1b, Enext); else Enext.cnt = Enext.e0 = 0;
            float* Zb = Z + (p & 1) * ZB + pp * PITCH;
            if (u == 16) {
                // Linear branch: Z = relu(x_i)
#pragma unroll
                for (int k = 0; k < NQ; ++k) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (i < A.n) v = *(const float4*)(A.x + (size_t)i * CIN + 4 * (t + 8 * k));
                    if (A.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    *(float4*)(Zb + 4 * (t + 8 * k)) = v;
                }
            } else {
                float4 acc[4][NQ];
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int k = 0; k < NQ; ++k) acc[c][k] = make_float4(0.f, 0.f, 0.f, 0.f);
                const int cnt = Ecur.cnt;
                const int isafe = min(i, A.n - 1);
                // one entry: gather relu(x[j]) and add it into the two touched nodes (weights of the other two are zero)
                auto add_entry = [&](int jc, float w0, float w1, const float4 (&xv)[NQ]) {
                    const int cx = (int)((unsigned)jc >> 30);
                    const float wc0 = cx == 0 ? w0 : 0.f;
                    const float wc1 = cx == 0 ? w1 : (cx == 1 ? w0 : 0.f);
                    const float wc2 = cx == 1 ? w1 : (cx == 2 ? w0 : 0.f);
                    const float wc3 = cx == 2 ? w1 : 0.f;
#pragma unroll
                    for (int k = 0; k < NQ; ++k) {
                        float4 v = xv[k];
                        if (A.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                        acc[0][k].x += wc0 * v.x; acc[0][k].y += wc0 * v.y; acc[0][k].z += wc0 * v.z; acc[0][k].w += wc0 * v.w;
                        acc[1][k].x += wc1 * v.x; acc[1][k].y += wc1 * v.y; acc[1][k].z += wc1 * v.z; acc[1][k].w += wc1 * v.w;
                        acc[2][k].x += wc2 * v.x; acc[2][k].y += wc2 * v.y; acc[2][k].z += wc2 * v.z; acc[2][k].w += wc2 * v.w;
                        acc[3][k].x += wc3 * v.x; acc[3][k].y += wc3 * v.y; acc[3][k].z += wc3 * v.z; acc[3][k].w += wc3 * v.w;
                    }
                };
                // the 8 threads of a point hold entries e0 + 8 s + t; four at a time are broadcast, their x rows requested
                // together, then accumulated in list order
#define GF_HALF(S, K0)                                                                                                   \
                if (__any(8 * (S) + (K0) < cnt)) {                                                                       \
                    int jc4[4]; float w04[4], w14[4]; float4 xv4[4][NQ];                                                 \
                    jc4[0] = gf_bcast8<(K0)>(Ecur.jc[S]);     jc4[1] = gf_bcast8<(K0) + 1>(Ecur.jc[S]);                  \
                    jc4[2] = gf_bcast8<(K0) + 2>(Ecur.jc[S]); jc4[3] = gf_bcast8<(K0) + 3>(Ecur.jc[S]);                  \
                    w04[0] = __int_as_float(gf_bcast8<(K0)>(__float_as_int(Ecur.w0[S])));                                \
                    w04[1] = __int_as_float(gf_bcast8<(K0) + 1>(__float_as_int(Ecur.w0[S])));                            \
                    w04[2] = __int_as_float(gf_bcast8<(K0) + 2>(__float_as_int(Ecur.w0[S])));                            \
                    w04[3] = __int_as_float(gf_bcast8<(K0) + 3>(__float_as_int(Ecur.w0[S])));                            \
                    w14[0] = __int_as_float(gf_bcast8<(K0)>(__float_as_int(Ecur.w1[S])));                                \
                    w14[1] = __int_as_float(gf_bcast8<(K0) + 1>(__float_as_int(Ecur.w1[S])));                            \
                    w14[2] = __int_as_float(gf_bcast8<(K0) + 2>(__float_as_int(Ecur.w1[S])));                            \
                    w14[3] = __int_as_float(gf_bcast8<(K0) + 3>(__float_as_int(Ecur.w1[S])));                            \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                      \
                        const bool ok = 8 * (S) + (K0) + e < cnt;                                                        \
                        const int j = ok ? (jc4[e] & 0x3fffffff) : isafe;                                                \
                        if (!ok) { w04[e] = 0.f; w14[e] = 0.f; jc4[e] = 0; }                                             \
                        _Pragma("unroll") for (int k = 0; k < NQ; ++k)                                                   \
                            xv4[e][k] = *(const float4*)(A.x + (size_t)j * CIN + 4 * (t + 8 * k));                       \
                    }                                                                                                    \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) add_entry(jc4[e], w04[e], w14[e], xv4[e]);             \
                }
                GF_HALF(0, 0) GF_HALF(0, 4) GF_HALF(1, 0) GF_HALF(1, 4) GF_HALF(2, 0) GF_HALF(2, 4)
#undef GF_HALF
                if (__any(cnt > 24)) {
                    // rows with more than 24 entries (rare): the tail straight from the list, one entry at a time
                    const uint32_t* eb = A.ent + (size_t)isafe * (size_t)(4 * A.pitch) * 3;
                    int emax = cnt;
#pragma unroll
                    for (int o = 8; o <= 32; o <<= 1) emax = max(emax, __shfl_xor(emax, o, 64));
                    for (int e = 24; e < emax; ++e) {
                        const bool ok = e < cnt;
                        const uint32_t* src = eb + 3 * (size_t)(Ecur.e0 + (ok ? e : 0));
                        int jc = ok ? (int)src[0] : 0;
                        const float w0 = ok ? __uint_as_float(src[1]) : 0.f, w1 = ok ? __uint_as_float(src[2]) : 0.f;
                        const int j = ok ? (jc & 0x3fffffff) : isafe;
                        float4 xv[NQ];
#pragma unroll
                        for (int k = 0; k < NQ; ++k) xv[k] = *(const float4*)(A.x + (size_t)j * CIN + 4 * (t + 8 * k));
                        add_entry(jc, w0, w1, xv);
                    }
                }
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int k = 0; k < NQ; ++k) *(float4*)(Zb + c * ZC + 4 * (t + 8 * k)) = acc[c][k];
            }
            Ecur = Enext;
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
    const float* scratch; int tiles, nwg, maxseg, ctot, coutp, cout, n;
    const float* bias_c; const float* bias_d; const float* residual; float* out;
    const float* pos; const float* pos_new; float* pos_c; float* vel_c; float scale, dt;       // pos == null: no update
};

__global__ void __launch_bounds__(256) k_cconv_gf_epi(GfEpi E)
{
    const int tile = blockIdx.x;
    const int wfirst = gf_owner((long long)tile * GF_COST, E.nwg, E.ctot), wlast = gf_owner((long long)tile * GF_COST + 64, E.nwg, E.ctot);
    const int nseg = wlast - wfirst + 1;
    const int slab = E.coutp * GF_TILE;
    for (int o = threadIdx.x; o < slab; o += 256) {
        const int col = o >> 5, row = o & 31, i = tile * GF_TILE + row;      // consecutive threads: consecutive rows of a column
        if (col >= E.cout || i >= E.n) continue;
        const float* s = E.scratch + (size_t)tile * E.maxseg * 4 * slab + o;
        float v = 0.f;
        for (int q = 0; q < nseg * 4; ++q) v += s[(size_t)q * slab];
        v += E.bias_c[col] + E.bias_d[col];
        if (E.residual) v += E.residual[(size_t)i * E.cout + col];
        E.out[(size_t)i * E.cout + col] = v;
        if (E.pos) {                                                          // cout == 3: col = coordinate (k_trans_update's expressions)
            const size_t e = (size_t)i * 3 + col;
            const float pc = E.pos_new[e] + E.scale * v;
            E.pos_c[e] = pc;
            E.vel_c[e] = (pc - E.pos[e]) / E.dt;
        }
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

template <int CIN, int NB>
static int gf_launch(const GfArgs& a, hipStream_t st)
{
    const size_t lds = (size_t)2 * 4 * GF_TILE * (CIN + 4) * sizeof(float);
    static bool attr_set[64] = {};
    int dev = 0;
    hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        hipFuncSetAttribute((const void*)k_cconv_gf<CIN, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((k_cconv_gf<CIN, NB>), dim3(a.nwg), dim3(GF_THREADS), lds, st, a);
    return 0;
}

// One G-free layer: y = cconv(act(x)) + Linear(act(x)) + biases (+ residual) [+ position / velocity update when pos != NULL]
extern "C" int nf_cconv_gf_layer(const float* x, int n, int cin, int cout, int relu, const uint16_t* roff, const uint32_t* ent,
                                 int pitch, const float* packed, const float* bias_conv, const float* bias_dense,
                                 const float* residual, float* out, float* scratch, int max_wg, const float* pos,
                                 const float* pos_new, float scale, float dt, float* pos_c, float* vel_c, nf_stream_t stream)
{
    NF_CHECK_ARG(x && roff && ent && packed && bias_conv && bias_dense && out && scratch, "null pointer");
    NF_CHECK_ARG((cin == 96 || cin == 64) && cout >= 1 && cout <= 64, "cin must be 96 or 64 (the transition model's layers), cout <= 64");
    NF_CHECK_ARG(!pos || (cout == 3 && pos_new && pos_c && vel_c), "the update epilogue belongs to the 3-channel layer");
    if (n <= 0) return NF_OK;
    GfArgs a;
    a.x = x; a.n = n; a.relu = relu; a.roff = roff; a.ent = ent; a.pitch = pitch; a.wp = packed; a.scratch = scratch;
    size_t sf;
    if (nf_cconv_gf_plan(n, cout, max_wg, &a.tiles, &a.nwg, &a.maxseg, &sf) != NF_OK) return NF_EINVAL;
    a.ctot = a.tiles * GF_COST;
    hipStream_t st = (hipStream_t)stream;
    const int nb = cout > 32 ? 2 : 1;
    if (cin == 96 && nb == 2) gf_launch<96, 2>(a, st);
    else if (cin == 96) gf_launch<96, 1>(a, st);
    else if (nb == 2) gf_launch<64, 2>(a, st);
    else gf_launch<64, 1>(a, st);
    NF_CHECK_LAUNCH();
    GfEpi e;
    e.scratch = scratch; e.tiles = a.tiles; e.nwg = a.nwg; e.maxseg = a.maxseg; e.ctot = a.ctot; e.coutp = 32 * nb; e.cout = cout; e.n = n;
    e.bias_c = bias_conv; e.bias_d = bias_dense; e.residual = residual; e.out = out;
    e.pos = pos; e.pos_new = pos_new; e.pos_c = pos_c; e.vel_c = vel_c; e.scale = scale; e.dt = dt;
    hipLaunchKernelGGL(k_cconv_gf_epi, dim3(a.tiles), dim3(256), 0, st, e);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// ------------------------------------------------------------------------------------------------
// The whole inference step behind ONE call (models/transmodel.py:151-163): prepare (integrate + fluid grid) -> front (search,
// row-entry lists, layer 0) -> conv1 -> conv2 -> conv3 + update.  8 launches, no host round trip inside; the host reads
// overflow2 (largest neighbour count above its pitch, or 0) to decide whether the step has to be redone on the exact path.
// ------------------------------------------------------------------------------------------------
extern "C" int nf_trans_step(const nf_trans_step_t* s, const float* pos, const float* vel, float* num_nbrs, float* pos_c,
                             float* vel_c, int64_t* host_overflow2, void* event, nf_stream_t stream)
{
    NF_CHECK_ARG(s && pos && vel && num_nbrs && pos_c && vel_c, "null pointer");
    int rc = nf_trans_prepare(pos, vel, s->gravity, s->dt, s->n, s->radius, s->bbox, s->grid_ws, s->grid_ws_bytes, s->pos_new,
                              s->vel_new, s->feats, stream);
    if (rc != NF_OK) return rc;
    rc = nf_trans_front(s->grid_ws, s->box_grid, s->pos_new, s->feats, s->box_feats, s->n, s->radius, s->extent, s->use_window,
                        s->pitch_f, s->pitch_b, s->counts2, num_nbrs, s->idx_f, s->d2_f, s->roff, s->ent, s->k_fluid, s->b_fluid,
                        s->k_obst, s->b_obst, s->dense0_w, s->dense0_b, s->a0, s->overflow2, stream);
    if (rc != NF_OK) return rc;
    if (host_overflow2) {
        // the overflow record is final behind the front kernel: it travels to the host (pinned memory) while the three
        // convolutions run, so the caller's wait on `event` costs no GPU time
        if (hipMemcpyAsync(host_overflow2, s->overflow2, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
            (event && hipEventRecord((hipEvent_t)event, (hipStream_t)stream) != hipSuccess)) {
            nf_set_error("nf_trans_step: overflow read-back failed");
            return NF_ELAUNCH;
        }
    }
    rc = nf_cconv_gf_layer(s->a0, s->n, 96, 64, 1, s->roff, s->ent, s->pitch_f, s->wp1, s->bc1, s->bd1, nullptr, s->a1, s->scratch,
                           s->max_wg, nullptr, nullptr, 0.f, 0.f, nullptr, nullptr, stream);
    if (rc != NF_OK) return rc;
    rc = nf_cconv_gf_layer(s->a1, s->n, 64, 64, 1, s->roff, s->ent, s->pitch_f, s->wp2, s->bc2, s->bd2, s->a1, s->a2, s->scratch,
                           s->max_wg, nullptr, nullptr, 0.f, 0.f, nullptr, nullptr, stream);
    if (rc != NF_OK) return rc;
    return nf_cconv_gf_layer(s->a2, s->n, 64, 3, 1, s->roff, s->ent, s->pitch_f, s->wp3, s->bc3, s->bd3, nullptr, s->y3, s->scratch,
                             s->max_wg, pos, s->pos_new, s->scale, s->dt, pos_c, vel_c, stream);
}
