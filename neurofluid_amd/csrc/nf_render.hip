// nf_render.hip — the non-MLP stages of RenderNet.forward (models/renderer.py:211-270):
//   classify (A1 + empty-space rejection), search (A2 + A7 + active-row compaction),
//   features (A3 + A4 + A5, written in the MLP operand layout), composite (A8), importance (A9).
//
// HBM/LDS notes: all of these are streaming or cache-resident integer/fp32 VALU
// kernels.  The particle cloud (59 KB at 4 913 points) and its cell lists stay in L2; the only
// HBM-sized traffic is the per-sample arrays (num_nn / mask / rgbsigma, 21 B per sample) and the
// feature matrix X (1 KB per ACTIVE row only, thanks to the compaction licensed by use_mask).
#include "nf_common.h"
#include <math.h>
#include <type_traits>

// wave-aggregated append: returns this lane's slot (valid only where pred)
__device__ __forceinline__ int wave_append(bool pred, int* counter)
{
    unsigned long long m = __ballot(pred);
    if (m == 0ull) return 0;
    int lane = threadIdx.x & 63;
    int n = __popcll(m);
    int leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, n);
    base = __shfl(base, leader, 64);
    return base + __popcll(m & ((1ull << lane) - 1ull));
}

__device__ __forceinline__ void sample_xyz(const float* __restrict__ rays, const float* __restrict__ z,
                                           const float* __restrict__ z_table, int S, int sample, float& x, float& y,
                                           float& zz_, float& zval)
{
    int r = sample / S, s = sample - r * S;
    const float* ry = rays + 6 * (size_t)r;
    zval = z ? z[sample] : z_table[s];
    x = nf_madd_nofma(ry[0], ry[3], zval);
    y = nf_madd_nofma(ry[1], ry[4], zval);
    zz_ = nf_madd_nofma(ry[2], ry[5], zval);
}

// ------------------------------------------------------------------------------------------------
// classify
// ------------------------------------------------------------------------------------------------
// Each thread takes CL_GROUPS quads of 4 consecutive samples (same ray when S % 4 == 0: the ray is read once,
// z as one 16-B load).  num_nn / mask of the whole quad are cleared with one 16-B and one 4-B store — the
// search overwrites the candidates afterwards — and nothing is written to rgbsigma: compositing reads it only
// where mask = 1 (use_mask) and the MLP fills exactly those rows.
#ifndef CL_GROUPS
#define CL_GROUPS 1      // quads per thread; 1 / 2 / 4: 103 / 114 / 147 us per launch (400^2 frame, coarse + fine average)
#endif
template <bool HAS_MASK>
__global__ void __launch_bounds__(256) k_classify(const void* __restrict__ ws, const float* __restrict__ rays,
                                                  const float* __restrict__ z, const float* __restrict__ z_table, int R,
                                                  int S, float radius, int use_mask, int* __restrict__ num_nn,
                                                  uint8_t* __restrict__ mask, int* __restrict__ cand,
                                                  int* __restrict__ cand_count)
{
    const float r2 = radius * radius;
    // one atomic per 2048 samples: per-thread flags -> block scan -> single reservation
    __shared__ int wsum[4];
    __shared__ int block_base;
    NfGridView g = nf_grid_view(ws);
    const int total = R * S;
    const int base = blockIdx.x * (256 * CL_GROUPS * 4);
    const bool s4 = (S & 3) == 0;
    unsigned flags = 0;
#pragma unroll
    for (int u = 0; u < CL_GROUPS; ++u) {
        const int i0 = base + (u * 256 + threadIdx.x) * 4;
        if (i0 >= total) continue;
        const bool whole = i0 + 3 < total;
        int r = i0 / S, sidx = i0 - r * S;
        const float* ry = rays + 6 * (size_t)r;
        float o0 = ry[0], o1 = ry[1], o2 = ry[2], d0 = ry[3], d1 = ry[4], d2 = ry[5];
        float zq[4];
        if (z && whole) { const float4 t = *(const float4*)(z + i0); zq[0] = t.x; zq[1] = t.y; zq[2] = t.z; zq[3] = t.w; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = i0 + e;
            if (i >= total) break;
            if (!s4 && e > 0 && sidx + e >= S) {      // quad straddles two rays (S % 4 != 0 only)
                const int r2_ = i / S;
                const float* rz = rays + 6 * (size_t)r2_;
                o0 = rz[0]; o1 = rz[1]; o2 = rz[2]; d0 = rz[3]; d1 = rz[4]; d2 = rz[5];
            }
            const float zv = z ? (whole ? zq[e] : z[i]) : z_table[i % S];
            const float x = nf_madd_nofma(o0, d0, zv), y = nf_madd_nofma(o1, d1, zv), zz = nf_madd_nofma(o2, d2, zv);
#ifdef CL_AB_NOREACH      /* dev probe (wrong candidates): what the 27-box reach test costs */
            if (!use_mask || nf_near_points_aabb(g, x, y, zz, radius))
#else
            if (!use_mask || (nf_near_points_aabb(g, x, y, zz, radius) && nf_any_cell_in_reach(g, x, y, zz, r2)))
#endif
                flags |= 1u << (u * 4 + e);
        }
        if (whole) {
            *(int4*)(num_nn + i0) = make_int4(0, 0, 0, 0);
            if (HAS_MASK) *(uchar4*)(mask + i0) = make_uchar4(0, 0, 0, 0);
        } else {
            for (int i = i0; i < total; ++i) { num_nn[i] = 0; if (HAS_MASK) mask[i] = 0; }
        }
    }
    int n = __popc(flags);
    // wave inclusive scan
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int x = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int yv = __shfl_up(x, o, 64);
        if (lane >= o) x += yv;
    }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        block_base = t ? atomicAdd(cand_count, t) : 0;
    }
    __syncthreads();
    int off = block_base + (x - n);
    for (int k = 0; k < w; ++k) off += wsum[k];
#pragma unroll
    for (int b = 0; b < CL_GROUPS * 4; ++b)
        if (flags & (1u << b)) cand[off++] = base + ((b >> 2) * 256 + threadIdx.x) * 4 + (b & 3);
}

extern "C" int nf_render_classify(const void* ws, const float* rays, const float* z, const float* z_table, int R, int S,
                                  float radius, int use_mask, int32_t* num_nn, uint8_t* mask, int32_t* cand,
                                  int32_t* cand_count, nf_stream_t stream)
{
    NF_CHECK_ARG(ws && rays && (z || z_table) && num_nn && cand && cand_count, "null pointer");
    NF_CHECK_ARG(R >= 0 && S > 0 && (long)R * S < 0x7fffffffL, "bad R/S");
    NF_CHECK_ARG(radius > 0.f, "bad radius");
    if (R == 0) return NF_OK;
    int total = R * S;
    int per_block = 256 * CL_GROUPS * 4;
    if (mask)
        hipLaunchKernelGGL(k_classify<true>, dim3((total + per_block - 1) / per_block), dim3(256), 0, (hipStream_t)stream, ws, rays,
                           z, z_table, R, S, radius, use_mask, num_nn, mask, cand, cand_count);
    else
        hipLaunchKernelGGL(k_classify<false>, dim3((total + per_block - 1) / per_block), dim3(256), 0, (hipStream_t)stream, ws, rays,
                           z, z_table, R, S, radius, use_mask, num_nn, mask, cand, cand_count);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// ------------------------------------------------------------------------------------------------
// search + compaction of active rows
// ------------------------------------------------------------------------------------------------
// A quarter-wave (16 lanes) per candidate.  The candidate's dilated list is walked in the reference's scan order,
// but 16 entries at a time: first the 16 lanes test the AABBs of the next 16 chunks at once, then every chunk that
// can reach the query has its 16 entries tested by the 16 lanes in one step; hits are placed by the prefix popcount
// of the group's ballot bits, so they land in list order and the first K of them are exactly the K lowest indices.
// (Measured on the 400^2 frame: on par with a thread per candidate, 0.46 vs 0.48 ms per launch — 93 % of the
// candidates are full rows that meet their K-th hit within the first chunks.)
#define SG_LANES 16
#define SG_BLOCK 256
#define SG_WAVES (SG_BLOCK / 64)
__global__ void __launch_bounds__(SG_BLOCK) k_search(const void* __restrict__ ws, const float* __restrict__ rays,
                                                     const float* __restrict__ z, const float* __restrict__ z_table,
                                                     int S, float r2, int K, int use_mask,
                                                     const int* __restrict__ cand, const int* __restrict__ cand_count,
                                                     int* __restrict__ num_nn,
                                                     int* __restrict__ row_sample, int* __restrict__ row_nbr,
                                                     int* __restrict__ n_rows, int cpr)
{
    // a wave takes cpr (64, or 16 for small launches) consecutive candidates per round, 4 at a time; their neighbour lists wait in LDS until the round
    // is over, so that the active rows of the round are reserved with ONE atomic (as with a thread per candidate)
    // LDS sized by K (dynamic): [wave][slot][pitch] lists, pitch = K | 1 (odd: conflict-free for both access patterns) +
    // [wave][slot] counts and flags.  At K = 20 that is 23.5 KB per workgroup instead of the 35.8 KB of a fixed pitch 33:
    // 6 instead of 4 workgroups per CU, and the kernel is latency-bound (rocprofv3: SQ_WAIT_ANY 0.62 of the wave cycles)
    extern __shared__ int s_dyn[];
    const int pitch = K | 1;
    int* const s_list_w = s_dyn + (threadIdx.x >> 6) * 64 * pitch;
    int* const s_cnt_w = s_dyn + SG_WAVES * 64 * pitch + (threadIdx.x >> 6) * 64;
    int* const s_full_w = s_cnt_w + SG_WAVES * 64;
    NfGridView g = nf_grid_view(ws);
    const int ncand = *cand_count;
    const int lane = threadIdx.x & 63, l = lane & (SG_LANES - 1), gsh = lane & ~(SG_LANES - 1), grp = lane >> 4;
    const int wv = threadIdx.x >> 6;
    const unsigned lt = (1u << l) - 1u;
    for (int base = (blockIdx.x * SG_WAVES + wv) * cpr; base < ncand; base += gridDim.x * SG_WAVES * cpr) {
        for (int step = 0; step < (cpr >> 2); ++step) {
            const int slot = step * 4 + grp;
            const int c = base + slot;
            int cnt = 0, nz = 0, sample = 0;
            if (c < ncand) {
                sample = cand[c];
                float x, y, zz, zv;
                sample_xyz(rays, z, z_table, S, sample, x, y, zz, zv);
                const int cx = nf_cell_coord(x, g.ox, g.icx, g.dx);
                const int cy = nf_cell_coord(y, g.oy, g.icy, g.dy);
                const int cz = nf_cell_coord(zz, g.oz, g.icz, g.dz);
                const int cell = (cz * g.dy + cy) * g.dx + cx;
                const int s = g.dil_start[cell], e = g.dil_start[cell + 1];
                const int cb0 = s & ~(NF_DIL_CHUNK - 1);
                const int nchunks = (e - cb0 + NF_DIL_CHUNK - 1) / NF_DIL_CHUNK;
                for (int cbase = 0; cbase < nchunks && cnt < K; cbase += SG_LANES) {
                    // 16 chunk boxes at once
                    const int ch = cbase + l;
                    bool pass = false;
                    if (ch < nchunks) {
                        const float4 blo = g.dil_box[2 * (cb0 / NF_DIL_CHUNK + ch)], bhi = g.dil_box[2 * (cb0 / NF_DIL_CHUNK + ch) + 1];
                        const float bb[6] = {blo.x, blo.y, blo.z, bhi.x, bhi.y, bhi.z};
                        pass = nf_box_dist2(bb, x, y, zz) < r2;
                    }
                    unsigned m = (unsigned)(__ballot(pass) >> gsh) & 0xffffu;
                    // (requesting the next passing chunks ahead of the test was measured: 961 -> 1 223 us for the fine pass — the
                    // loads wasted by the early exit at the K-th hit cost more than the latency they hide; the kernel is bound
                    // by the requests it issues, not by the requests it keeps in flight)
#ifndef SG_PAIR
#define SG_PAIR 1        // passing chunks walked TWO per trip (round 5): both chunks' entries are requested together, so a full row's ~8 dependent trips become ~4
#endif
                    while (m && cnt < K) {
                        const int bit = __ffs(m) - 1;
                        m &= m - 1u;
                        const int t = cb0 + (cbase + bit) * NF_DIL_CHUNK + l;      // NF_DIL_CHUNK == SG_LANES: one entry per lane
#if SG_PAIR
                        const bool two = m != 0u;                                  // (uniform inside the quarter-wave)
                        const int bit2 = two ? __ffs(m) - 1 : 0;
                        if (two) m &= m - 1u;
                        const int t2 = cb0 + (cbase + bit2) * NF_DIL_CHUNK + l;
                        const bool in1 = t >= s && t < e, in2 = two && t2 >= s && t2 < e;
                        float4 p = make_float4(0.f, 0.f, 0.f, 0.f), p2 = p;
                        if (in1) p = g.dil_pos[t];
                        if (in2) p2 = g.dil_pos[t2];
                        bool hit = false, nonzero = false, hit2 = false, nonzero2 = false;
                        if (in1) {
                            const float d2 = nf_dist2(x, y, zz, p.x, p.y, p.z);
                            hit = d2 < r2;
                            nonzero = d2 != 0.f;
                        }
                        if (in2) {
                            const float d2 = nf_dist2(x, y, zz, p2.x, p2.y, p2.z);
                            hit2 = d2 < r2;
                            nonzero2 = d2 != 0.f;
                        }
                        const unsigned hm = (unsigned)(__ballot(hit) >> gsh) & 0xffffu;
                        const unsigned hm2 = (unsigned)(__ballot(hit2) >> gsh) & 0xffffu;
                        const int pos = cnt + __popc(hm & lt);
                        const int cnt1 = cnt + __popc(hm);
                        const int pos2 = cnt1 + __popc(hm2 & lt);
                        const bool take = hit && pos < K, take2 = hit2 && pos2 < K;
                        if (take) s_list_w[slot * pitch + pos] = __float_as_int(p.w);
                        if (take2) s_list_w[slot * pitch + pos2] = __float_as_int(p2.w);
                        // (round 6, measured: counting the kept hits arithmetically and subtracting the zero-distance ones behind a wave-uniform
                        // test — two ballots fewer per trip in the common case — is SLOWER, 2 172 vs 2 127 us per launch on the honeycone 800^2 frame)
                        nz += __popc((unsigned)(__ballot(take && nonzero) >> gsh) & 0xffffu) +       // nn_mask = dists.ne(0)
                              __popc((unsigned)(__ballot(take2 && nonzero2) >> gsh) & 0xffffu);
                        cnt = cnt1 + __popc(hm2);
#else
                        bool hit = false, nonzero = false;
                        int pi = 0;
                        if (t >= s && t < e) {
                            const float4 p = g.dil_pos[t];
                            const float d2 = nf_dist2(x, y, zz, p.x, p.y, p.z);
                            hit = d2 < r2;
                            nonzero = d2 != 0.f;
                            pi = __float_as_int(p.w);
                        }
                        const unsigned hm = (unsigned)(__ballot(hit) >> gsh) & 0xffffu;
                        const int pos = cnt + __popc(hm & lt);
                        const bool take = hit && pos < K;
                        if (take) s_list_w[slot * pitch + pos] = pi;
                        nz += __popc((unsigned)(__ballot(take && nonzero) >> gsh) & 0xffffu);   // nn_mask = dists.ne(0)
                        cnt += __popc(hm);
#endif
                    }
                }
                if (cnt > K) cnt = K;
                if (l == 0) {
                    const bool full = (nz == K);
                    num_nn[sample] = nz;
                    s_cnt_w[slot] = cnt;
                    s_full_w[slot] = full ? 1 : 0;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        // ---- end of the round: lane i owns slot i
        const int c = base + lane;
        bool active = false;
        int cnt = 0, sample = 0;
        if (lane < cpr && c < ncand) {
            sample = cand[c];
            cnt = s_cnt_w[lane];
            active = s_full_w[lane] || !use_mask;
        }
        const int row = wave_append(active, n_rows);
        // the round's active rows are consecutive (wave_append): their K-lists leave as ONE contiguous run written by the
        // whole wave (64 consecutive ints per store) instead of every lane storing its own 80-B row 4 bytes at a time
        const unsigned long long am = __ballot(active);
        if (am) {
            const int na = __popcll(am);
            const int row0 = __shfl(row, __ffsll((long long)am) - 1, 64);
            if (active) {
                row_sample[row] = sample;
                s_full_w[row - row0] = lane;        // rank -> slot (s_full is dead: `active` has been derived from it)
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            const unsigned invK = (1u << 16) / (unsigned)K + 1u;      // idx / K for idx < 64 K, K <= 32 (exact: checked offline)
            int* const dst = row_nbr + (size_t)row0 * K;
            for (int idx = lane; idx < na * K; idx += 64) {
                const int rr = (int)(((unsigned)idx * invK) >> 16), k = idx - rr * K;
                const int slot = s_full_w[rr];
                dst[idx] = k < s_cnt_w[slot] ? s_list_w[slot * pitch + k] : -1;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
}

// mask[sample] = (num_nn[sample] == K) for every candidate: only for callers that want the byte array (the fused renderer
// keeps no mask array, nf_composite_* derive the bit from num_nn).
__global__ void k_mask_from_num_nn(const int* __restrict__ cand, const int* __restrict__ cand_count, const int* __restrict__ num_nn,
                                   int K, uint8_t* __restrict__ mask)
{
    const int n = *cand_count;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
        const int sample = cand[c];
        mask[sample] = num_nn[sample] == K ? 1 : 0;
    }
}

extern "C" int nf_render_search(const void* ws, const float* rays, const float* z, const float* z_table, int R, int S,
                                float radius, int K, int use_mask, const int32_t* cand, const int32_t* cand_count,
                                int32_t* num_nn, uint8_t* mask, int32_t* row_sample, int32_t* row_nbr,
                                int32_t* n_rows, nf_stream_t stream)
{
    NF_CHECK_ARG(ws && rays && (z || z_table) && cand && cand_count && num_nn && row_sample && row_nbr && n_rows,
                 "null pointer");
    NF_CHECK_ARG(K >= 1 && K <= 32 && radius > 0.f, "bad K/radius");
    if (R == 0) return NF_OK;
    static_assert(NF_DIL_CHUNK == SG_LANES, "one list entry per lane of a group");
    long total = (long)R * S;
    long want = (total + SG_BLOCK - 1) / SG_BLOCK;
    int blocks = (int)(want < 8192 ? want : 8192);          // grid-stride over the candidates (count known on device only)
    // Small launches (the training steps: < 1 M samples, ~10 % of them candidates) leave most of these waves without work
    // while the others walk 64 candidates one group of 4 after the other (16 dependent steps ~ 130 us whatever the count):
    // rounds of 16 candidates spread the same candidates over 4x the waves.
    const int cpr = total < (1L << 20) ? 16 : 64;
    const size_t lds = (size_t)SG_WAVES * 64 * ((K | 1) + 2) * sizeof(int);
    hipLaunchKernelGGL(k_search, dim3(blocks), dim3(SG_BLOCK), lds, (hipStream_t)stream, ws, rays, z, z_table, S,
                       radius * radius, K, use_mask, cand, cand_count, num_nn, row_sample, row_nbr, n_rows, cpr);
    if (mask) hipLaunchKernelGGL(k_mask_from_num_nn, dim3(blocks), dim3(256), 0, (hipStream_t)stream, cand, cand_count, (const int*)num_nn, K, mask);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// ------------------------------------------------------------------------------------------------
// features
// ------------------------------------------------------------------------------------------------
extern "C" int nf_render_feature_dims(int enc_flags, int* cx, int* cd, int* qx, int* qd)
{
    int x = 63, d = 27;
    if (enc_flags & 1) x += 9;
    if (enc_flags & 2) x += 63;
    if (enc_flags & 4) x += 63;
    if (enc_flags & 8) d += 27;
    if (cx) *cx = x;
    if (cd) *cd = d;
    if (qx) *qx = (x + 7) / 8;
    if (qd) *qd = (d + 7) / 8;
    return NF_OK;
}

// Streams features of one row into X[tile][q][lane][4]: 4 consecutive features share one 16-B store.
// Every value is put at an EXPLICIT feature index that is a compile-time constant once the callers' loops are unrolled (the
// encodings' offsets follow from the FLAGS template argument): the staging registers are then addressed statically.  (With a
// running counter inside the emitter the compiler kept it — and the fp16 emitter's 16-entry staging array — in memory: the
// array was promoted to LDS, ~2 000 ds_ instructions and ~450 address computations in k_features<15, true>.)
struct FeatEmitter {
    float4* base;  // &X[tile][0][j]  (lane h=0); h=1 is +32 float4, q is +64 float4
    float b[4];
    __device__ __forceinline__ void put(int idx, float v)
    {
        b[idx & 3] = v;
        if ((idx & 3) == 3) {
            const int group = idx >> 2, q = group >> 1, h = group & 1;
            typedef float nf_f4v __attribute__((ext_vector_type(4)));
            const nf_f4v o = {b[0], b[1], b[2], b[3]};
            __builtin_nontemporal_store(o, (nf_f4v*)&base[q * 64 + h * 32]);   // streamed: read once, by the MLP
        }
    }
};

// fp16 variant for the fp16-MFMA MLP: Xh[tile][t][lane] = 16 B = the lane's 8 halves of K-step t, i.e. features
// 16t + 4h + e (e < 4) and 16t + 8 + 4h + e — the h8 B operand itself (RNE conversion, identical to converting the
// fp32 layout inside the kernel), at half the bytes.
typedef _Float16 nf_h8 __attribute__((ext_vector_type(8)));
struct FeatEmitterH {
    nf_h8* base;   // &Xh[tile][0][j]  (lane h=0); h=1 is +32, K-step t is +64
    _Float16 f[16];
    __device__ __forceinline__ void put(int idx, float v)
    {
        f[idx & 15] = (_Float16)v;
        if ((idx & 15) == 15) {
            const int t = idx >> 4;
            nf_h8 lo = {f[0], f[1], f[2], f[3], f[8], f[9], f[10], f[11]};
            nf_h8 hi = {f[4], f[5], f[6], f[7], f[12], f[13], f[14], f[15]};
            __builtin_nontemporal_store(lo, &base[t * 64]);
            __builtin_nontemporal_store(hi, &base[t * 64 + 32]);
        }
    }
};

// zero-fill features [from, to)
template <int FROM, int TO, typename EM>
__device__ __forceinline__ void emit_pad(EM& em)
{
#pragma unroll
    for (int i = FROM; i < TO; ++i) em.put(i, 0.f);
}

// Positional encoding [x, sin(2^k x), cos(2^k x)]_k (models/nerf.py:17-41; the reference evaluates torch.sin / cos of
// the fp32 product 2^k * x, which is exact).  One sincos of x in DOUBLE, then angle doubling in double
// (sin 2a = 2 sin a cos a, cos 2a = 1 - 2 sin^2 a): ten octaves cost one argument reduction instead of ten, and the
// doubling error (< 2^10 * 1e-16) stays far below fp32 rounding, so every emitted value is the correctly rounded
// sin / cos of the reference's argument.  (Ten independent sincosf calls made this kernel VALU-bound.)
template <int BASE, int C, int NF, typename EM>
__device__ __forceinline__ void emit_pe(EM& em, const float (&x)[C])
{
#pragma unroll
    for (int c = 0; c < C; ++c) em.put(BASE + c, x[c]);
    double sn[C], cs[C];
#pragma unroll
    for (int c = 0; c < C; ++c) sincos((double)x[c], &sn[c], &cs[c]);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
        for (int c = 0; c < C; ++c) em.put(BASE + C * (1 + 2 * f) + c, (float)sn[c]);
#pragma unroll
        for (int c = 0; c < C; ++c) em.put(BASE + C * (2 + 2 * f) + c, (float)cs[c]);
        if (f + 1 < NF) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const double s2 = 2.0 * sn[c] * cs[c], c2 = 1.0 - 2.0 * sn[c] * sn[c];
                sn[c] = s2; cs[c] = c2;
            }
        }
    }
}

// KC = K when K is 20 (the configs' value; 0 = any K): the row's neighbour indices are five 16-B loads kept in registers
// for both sweeps, and the sweeps unroll, so the 20 position gathers of a sweep are in flight together instead of one
// index -> position chain after the other.  Same arithmetic, same order.
template <int FLAGS, bool HALF, int KC>
__global__ void __launch_bounds__(128)
#ifndef NF_FEAT_NOCAP
__attribute__((amdgpu_waves_per_eu(4, 8)))
#endif
k_features(const float* __restrict__ particles, const float* __restrict__ rays,
                                                  const float* __restrict__ z, const float* __restrict__ z_table, int S,
                                                  float radius, int K, const float* __restrict__ ro_base, int ro_stride,
                                                  const int* __restrict__ row_sample, const int* __restrict__ row_nbr,
                                                  const int* __restrict__ n_rows, int max_rows, void* __restrict__ X, int blend)
{
    constexpr int CX = 63 + ((FLAGS & 1) ? 9 : 0) + ((FLAGS & 2) ? 63 : 0) + ((FLAGS & 4) ? 63 : 0);
    constexpr int CD = 27 + ((FLAGS & 8) ? 27 : 0);
    constexpr int QX = (CX + 7) / 8, QD = (CD + 7) / 8, Q = QX + QD;
    int nrows = min(*n_rows, max_rows);
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < nrows; row += gridDim.x * blockDim.x) {
        int sample = row_sample[row];
        float px[3], zv;
        sample_xyz(rays, z, z_table, S, sample, px[0], px[1], px[2], zv);
        const float* ry = rays + 6 * (size_t)(sample / S);
        const float* ro = ro_base + (size_t)ro_stride * (sample / S);   // camera position: shared (stride 0) or per ray
        // --- A3 smoothing + A4 variance over the K slots (padded slots: nn = 0, models/renderer.py:96-109,:163-169)
        float sw = 0.f, swx = 0.f, swy = 0.f, swz = 0.f;
        float sx = 0.f, sy = 0.f, sz = 0.f;
        int nvalid = 0;
        int nb[KC > 0 ? KC : 1];
        if constexpr (KC > 0) {
#pragma unroll
            for (int q4 = 0; q4 < KC / 4; ++q4) {
                const int4 v = *(const int4*)(row_nbr + (size_t)row * KC + 4 * q4);
                nb[4 * q4] = v.x; nb[4 * q4 + 1] = v.y; nb[4 * q4 + 2] = v.z; nb[4 * q4 + 3] = v.w;
            }
        }
        const int Kn = KC > 0 ? KC : K;
        // measured on the 400^2 frame (us per launch, coarse + fine average), gathers first, then the static-index emitters:
        // fp32 X 479 -> 328 (full unroll) -> 315 (<= 128 registers) -> 300 (emitters; sweeps unrolled by 5 / 10 / 20: 303 / 300 / 337)
        // fp16 X 421 -> 327 (sweeps unrolled by 5; 2 / 4 / 10 / 20: 338 / 332 / 346 / 444) -> 234 (emitters)
#ifndef NF_FEAT_UNH
#define NF_FEAT_UNH 5
#endif
#ifndef NF_FEAT_UNF
#define NF_FEAT_UNF 10
#endif
        constexpr int UN = HALF ? NF_FEAT_UNH : NF_FEAT_UNF;
#pragma unroll UN
        for (int k = 0; k < Kn; ++k) {
            int j;
            if constexpr (KC > 0) j = nb[k]; else j = row_nbr[(size_t)row * K + k];
            float nx = 0.f, ny = 0.f, nz = 0.f;
            bool valid = false;
            if (j >= 0) {
                nx = particles[3 * (size_t)j]; ny = particles[3 * (size_t)j + 1]; nz = particles[3 * (size_t)j + 2];
                valid = nf_dist2(px[0], px[1], px[2], nx, ny, nz) != 0.f;
            }
            float dx = nx - px[0], dy = ny - px[1], dz = nz - px[2];
            float dist = sqrtf(dx * dx + dy * dy + dz * dz);
            float t = dist / radius;
            float w = fmaxf(1.f - t * t * t, 0.f);
            sw += w; swx += w * nx; swy += w * ny; swz += w * nz;
            if (valid) { sx += dx; sy += dy; sz += dz; ++nvalid; }
        }
        float den = sw + 1e-12f;
        float sm[3] = {swx / den, swy / den, swz / den};
        if (blend) {
            // encoding.exclude_ray=False (models/renderer.py:100-106): pos = ray_pos * (1 - alpha) + weighted_nn * alpha with
            // alpha = 0.9, or 0.1 where num_nn <= 20 (the literal of :105) unless same_smooth_factor (blend == 2);
            // (1 - alpha) is the fp32 difference torch forms, the sum is mul, mul, add (no FMA: -ffp-contract=off)
            const float alpha = (blend == 2 || nvalid > 20) ? 0.9f : 0.1f, oma = 1.f - alpha;
#pragma unroll
            for (int c = 0; c < 3; ++c) sm[c] = px[c] * oma + sm[c] * alpha;
        }
        float nn_f = (float)nvalid + 1e-12f;
        float mean[3] = {sx / nn_f, sy / nn_f, sz / nn_f};
        float var[3] = {0.f, 0.f, 0.f};
        if (FLAGS & 4) {
#pragma unroll UN
            for (int k = 0; k < Kn; ++k) {
                int j;
                if constexpr (KC > 0) j = nb[k]; else j = row_nbr[(size_t)row * K + k];
                if (j < 0) continue;
                float nx = particles[3 * (size_t)j], ny = particles[3 * (size_t)j + 1], nz = particles[3 * (size_t)j + 2];
                if (nf_dist2(px[0], px[1], px[2], nx, ny, nz) == 0.f) continue;
                float ex = (nx - px[0]) - mean[0], ey = (ny - px[1]) - mean[1], ez = (nz - px[2]) - mean[2];
                var[0] += ex * ex; var[1] += ey * ey; var[2] += ez * ez;
            }
            var[0] /= nn_f; var[1] /= nn_f; var[2] /= nn_f;
        }
        // --- emit in the reference's column order (models/renderer.py:141-175, cat at :230)
        int tile = row >> 5, jj = row & 31;
        typename std::conditional<HALF, FeatEmitterH, FeatEmitter>::type em;
        if constexpr (HALF) em.base = (nf_h8*)X + (size_t)tile * (Q / 2) * 64 + jj;
        else em.base = (float4*)X + (size_t)tile * Q * 64 + jj;
        constexpr int O_DEN = 63, O_SM = O_DEN + ((FLAGS & 1) ? 9 : 0), O_VAR = O_SM + ((FLAGS & 2) ? 63 : 0);
        constexpr int O_DIR = QX * 8, O_SDIR = O_DIR + 27;
        static_assert(O_VAR + ((FLAGS & 4) ? 63 : 0) == CX && O_SDIR + ((FLAGS & 8) ? 27 : 0) == O_DIR + CD, "feature offsets");
        emit_pe<0, 3, 10>(em, px);
        if constexpr ((FLAGS & 1) != 0) { float d1[1] = {sw}; emit_pe<O_DEN, 1, 4>(em, d1); }
        if constexpr ((FLAGS & 2) != 0) emit_pe<O_SM, 3, 10>(em, sm);
        if constexpr ((FLAGS & 4) != 0) emit_pe<O_VAR, 3, 10>(em, var);
        emit_pad<CX, QX * 8>(em);
        float rd[3] = {ry[3], ry[4], ry[5]};
        emit_pe<O_DIR, 3, 4>(em, rd);
        if constexpr ((FLAGS & 8) != 0) {
            float ddx = sm[0] - ro[0], ddy = sm[1] - ro[1], ddz = sm[2] - ro[2];
            float nrm = sqrtf(ddx * ddx + ddy * ddy + ddz * ddz);
            float sd[3] = {ddx / nrm, ddy / nrm, ddz / nrm};
            emit_pe<O_SDIR, 3, 4>(em, sd);
        }
        emit_pad<O_DIR + CD, Q * 8>(em);
    }
}

extern "C" int nf_render_features(const float* particles, const float* rays, const float* z, const float* z_table, int R,
                                  int S, float radius, int K, int enc_flags, const float* ro, int ro_per_ray,
                                  const int32_t* row_sample, const int32_t* row_nbr, const int32_t* n_rows, int max_rows,
                                  void* X, int x_fp16, nf_stream_t stream)
{
    NF_CHECK_ARG(particles && rays && (z || z_table) && ro && row_sample && row_nbr && n_rows && X, "null pointer");
    NF_CHECK_ARG(enc_flags >= 0 && enc_flags < 64 && (enc_flags >> 4) != 2, "bad enc_flags");
    const int blend = (enc_flags & 16) ? ((enc_flags & 32) ? 2 : 1) : 0;       // exclude_ray=False [, same_smooth_factor]
    enc_flags &= 15;
    if (x_fp16) {
        int qx = 0, qd = 0;
        nf_render_feature_dims(enc_flags, nullptr, nullptr, &qx, &qd);
        NF_CHECK_ARG(((qx + qd) & 1) == 0, "the fp16 operand layout needs an even number of 8-feature groups");
    }
    if (max_rows <= 0) return NF_OK;
    int blocks = (max_rows + 127) / 128;
    if (blocks > 4096) blocks = 4096;
    hipStream_t st = (hipStream_t)stream;
    // K = 20 with the row lists 16-B aligned (80-B rows): the specialised kernel; only the default encoding gets it (15)
    const bool k20 = K == 20 && enc_flags == 15 && (((uintptr_t)row_nbr) & 15) == 0;
#define NF_FEAT_CASE(F)                                                                                              \
    case F:                                                                                                          \
        if (x_fp16 && k20)                                                                                           \
            hipLaunchKernelGGL((k_features<F, true, 20>), dim3(blocks), dim3(128), 0, st, particles, rays, z, z_table, S, radius, \
                               K, ro, ro_per_ray ? 3 : 0, row_sample, row_nbr, n_rows, max_rows, X, blend);               \
        else if (x_fp16)                                                                                             \
            hipLaunchKernelGGL((k_features<F, true, 0>), dim3(blocks), dim3(128), 0, st, particles, rays, z, z_table, S, radius, \
                               K, ro, ro_per_ray ? 3 : 0, row_sample, row_nbr, n_rows, max_rows, X, blend);               \
        else if (k20)                                                                                                \
            hipLaunchKernelGGL((k_features<F, false, 20>), dim3(blocks), dim3(128), 0, st, particles, rays, z, z_table, S, radius, \
                               K, ro, ro_per_ray ? 3 : 0, row_sample, row_nbr, n_rows, max_rows, X, blend);               \
        else                                                                                                         \
            hipLaunchKernelGGL((k_features<F, false, 0>), dim3(blocks), dim3(128), 0, st, particles, rays, z, z_table, S, radius, \
                               K, ro, ro_per_ray ? 3 : 0, row_sample, row_nbr, n_rows, max_rows, X, blend);               \
        break;
    switch (enc_flags) {
        NF_FEAT_CASE(0) NF_FEAT_CASE(1) NF_FEAT_CASE(2) NF_FEAT_CASE(3) NF_FEAT_CASE(4) NF_FEAT_CASE(5) NF_FEAT_CASE(6)
        NF_FEAT_CASE(7) NF_FEAT_CASE(8) NF_FEAT_CASE(9) NF_FEAT_CASE(10) NF_FEAT_CASE(11) NF_FEAT_CASE(12)
        NF_FEAT_CASE(13) NF_FEAT_CASE(14) NF_FEAT_CASE(15)
    }
#undef NF_FEAT_CASE
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// ------------------------------------------------------------------------------------------------
// composite (A8): one thread per ray, sequential transmittance like torch.cumprod
// ------------------------------------------------------------------------------------------------
// 64 rays per wave; rgbsigma / z tiles of 16 samples are staged through LDS with coalesced 16-B accesses
// (a ray's 16 samples are 256 contiguous bytes), each thread then walks ITS ray sequentially — the same
// association as torch.cumprod — and the weights leave through LDS the same way.
// gate != 0 (use_mask): rgbsigma of a sample with mask = 0 IS zero (rgbsigma * mask, models/renderer.py:237) and is
// never read — nobody wrote it.  A zero sample has alpha = 0, weight 0 and leaves T unchanged bit for bit
// (1 - 0 + 1e-10 rounds to 1.f), so a 16-sample tile without any mask bit in the whole wave is skipped outright:
// 1 B per sample of traffic for the ~97 % of the volume that is empty.
#define CP_TS 16
#define CP_PITCH 17
__global__ void __launch_bounds__(64) k_composite(const float4* __restrict__ rgbsigma, const float* __restrict__ z,
                                                  const float* __restrict__ z_table, const float* __restrict__ rays,
                                                  const uint8_t* __restrict__ mask, int gate, int R, int S, int white_bg,
                                                  float* __restrict__ rgb, float* __restrict__ depth,
                                                  float* __restrict__ opacity, float* __restrict__ weights,
                                                  float* __restrict__ mask_sum, const int* __restrict__ num_nn, int k_full,
                                                  const float* __restrict__ noise)
{
    // noise (optional, R x S): added to sigma before the ReLU (models/renderer.py:193-196, noise_std > 0).  The reference adds it to
    // EVERY sample, masked ones included (their sigma is 0 * mask + noise), so no tile is skipped then; rgbsigma is still
    // read only where the mask allows.
    __shared__ float4 s_rs[64 * CP_PITCH];
    __shared__ float s_z[64 * (CP_TS + 1) + 64];
    __shared__ float s_w[64 * CP_PITCH];
    __shared__ unsigned s_m[64];            // per ray: bit k = mask of sample s0 + k
    const int t = threadIdx.x;
    const int r0 = blockIdx.x * 64, r = r0 + t;
    const bool live = r < R;
    float nrm = 0.f;
    if (live) {
        const float* ry = rays + 6 * (size_t)r;
        nrm = sqrtf(ry[3] * ry[3] + ry[4] * ry[4] + ry[5] * ry[5]);
    }
    float T = 1.f, cr = 0.f, cg = 0.f, cb = 0.f, cd = 0.f, ws = 0.f;
    int ms = 0;
    const int sub = t >> 4, col = t & 15;   // staging role: ray-in-group, sample-in-tile
    const bool vec_mask = (mask || num_nn) && (S & 15) == 0;
    const bool have_mask = mask || num_nn;      // mask bit of a sample: mask[s] != 0, or (mask == NULL) num_nn[s] == k_full
    for (int s0 = 0; s0 < S; s0 += CP_TS) {
        const int ns = min(CP_TS, S - s0);
        // ---- this ray's 16 mask bytes (one 16-B load when the row is 16-B tiled)
        unsigned mbits = 0;
        if (have_mask && live) {
            if (!mask) {        // derived from the neighbour counts (4 x 16 B per tile when the row is 16-B tiled)
                if (vec_mask) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int4 nv = *(const int4*)(num_nn + (size_t)r * S + s0 + 4 * q);
                        mbits |= ((nv.x == k_full ? 1u : 0u) | (nv.y == k_full ? 2u : 0u) | (nv.z == k_full ? 4u : 0u) |
                                  (nv.w == k_full ? 8u : 0u)) << (4 * q);
                    }
                } else {
                    for (int k = 0; k < ns; ++k) mbits |= (num_nn[(size_t)r * S + s0 + k] == k_full ? 1u : 0u) << k;
                }
            } else if (vec_mask) {
                const uint4 mv = *(const uint4*)(mask + (size_t)r * S + s0);
                const unsigned wv[4] = {mv.x, mv.y, mv.z, mv.w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int b = 0; b < 4; ++b) mbits |= (((wv[q] >> (8 * b)) & 0xffu) ? 1u : 0u) << (4 * q + b);
            } else {
                for (int k = 0; k < ns; ++k) mbits |= (mask[(size_t)r * S + s0 + k] ? 1u : 0u) << k;
            }
            ms += __popc(mbits);
        }
        const unsigned abits = gate ? mbits : (live ? 0xffffu : 0u);     // samples whose rgbsigma is read
        if (!noise && __ballot(abits != 0u) == 0ull) {
            // nothing to composite in this tile for any of the 64 rays
            if (weights) {
#pragma unroll 4
                for (int i = 0; i < 16; ++i) {
                    int rr = i * 4 + sub, gr = r0 + rr, gs = s0 + col;
                    if (gr < R && gs < S) weights[(size_t)gr * S + gs] = 0.f;
                }
            }
            continue;
        }
        s_m[t] = abits;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        // ---- stage in: 4 rays per instruction, 16 samples x 16 B each (only where the sample is live)
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            int rr = i * 4 + sub, gr = r0 + rr, gs = s0 + col;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gr < R && gs < S && ((s_m[rr] >> col) & 1u)) v = rgbsigma[(size_t)gr * S + gs];
            s_rs[rr * CP_PITCH + col] = v;
        }
        // z tile incl. one look-ahead element per ray
        for (int i = 0; i < 16; ++i) {
            int rr = i * 4 + sub, gr = r0 + rr;
            for (int cc = col; cc < CP_TS + 1; cc += 16) {
                int gs = s0 + cc;
                float zv = 0.f;
                if (gs < S) zv = z ? (gr < R ? z[(size_t)gr * S + gs] : 0.f) : z_table[gs];
                s_z[rr * (CP_TS + 1) + cc] = zv;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        // ---- sequential walk of this thread's ray
        if (live) {
            for (int k = 0; k < ns; ++k) {
                int s = s0 + k;
                float zc = s_z[t * (CP_TS + 1) + k], zn = s_z[t * (CP_TS + 1) + k + 1];
                float delta = ((s + 1 < S) ? (zn - zc) : 1e10f) * nrm;
                float4 v = s_rs[t * CP_PITCH + k];
                const float sg = noise ? v.w + noise[(size_t)r * S + s] : v.w;
                float alpha = 1.f - expf(-delta * fmaxf(sg, 0.f));
                float w = alpha * T;
                T = T * ((1.f - alpha) + 1e-10f);
                cr += w * v.x; cg += w * v.y; cb += w * v.z; cd += w * zc; ws += w;
                s_w[t * CP_PITCH + k] = w;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        // ---- stage out the weights (64 B per ray per instruction)
        if (weights) {
#pragma unroll 4
            for (int i = 0; i < 16; ++i) {
                int rr = i * 4 + sub, gr = r0 + rr, gs = s0 + col;
                if (gr < R && gs < S) weights[(size_t)gr * S + gs] = s_w[rr * CP_PITCH + col];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (!live) return;
    if (white_bg) { cr = cr + 1.f - ws; cg = cg + 1.f - ws; cb = cb + 1.f - ws; }
    rgb[3 * (size_t)r] = cr; rgb[3 * (size_t)r + 1] = cg; rgb[3 * (size_t)r + 2] = cb;
    depth[r] = cd;
    opacity[r] = ws;
    if (mask_sum) mask_sum[r] = (float)ms;
}

// The fused renderer's case (mask bits from the neighbour counts, rows 16-B tiled, S <= 256), software-pipelined: the mask
// bits of ALL tiles of the 64 rays are fetched up front (16 bits per tile, 8 registers, published once through LDS), tiles
// without a live sample in the whole wave are skipped without touching memory, and the rgbsigma / z values of the NEXT live
// tile are in flight (held in registers) while this tile is walked.  k_composite pays three dependent global round trips
// per tile (counts -> gated rgbsigma / z -> walk), 12 tiles per ray at S = 192, and its duration is that of ONE block
// (the ~750 blocks that touch the fluid are all resident at once: halving the occupancy changes nothing).  Same
// arithmetic in the same order: bit-identical outputs.
__global__ void __launch_bounds__(64) k_composite_p(const float4* __restrict__ rgbsigma, const float* __restrict__ z,
                                                    const float* __restrict__ z_table, const float* __restrict__ rays, int gate,
                                                    int R, int S, int white_bg, float* __restrict__ rgb, float* __restrict__ depth,
                                                    float* __restrict__ opacity, float* __restrict__ weights,
                                                    float* __restrict__ mask_sum, const int* __restrict__ num_nn, int k_full)
{
    __shared__ float4 s_rs[64 * CP_PITCH];
    __shared__ float s_z[64 * (CP_TS + 1) + 64];
    __shared__ float s_w[64 * CP_PITCH];
    __shared__ unsigned s_mw[64][8];        // per ray: 16 live-sample bits per tile, two tiles per word
    const int t = threadIdx.x;
    const int r0 = blockIdx.x * 64, r = r0 + t;
    const bool live = r < R;
    const int NT = S >> 4;
    float nrm = 0.f;
    if (live) {
        const float* ry = rays + 6 * (size_t)r;
        nrm = sqrtf(ry[3] * ry[3] + ry[4] * ry[4] + ry[5] * ry[5]);
    }
    // ---- mask bits of every tile of this ray: 4 x 16 B per tile, all loads independent
    unsigned mw[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    int ms = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        unsigned bits = 0u;
        if (live && w < NT) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int4 nv = *(const int4*)(num_nn + (size_t)r * S + 16 * w + 4 * q);
                bits |= ((nv.x == k_full ? 1u : 0u) | (nv.y == k_full ? 2u : 0u) | (nv.z == k_full ? 4u : 0u) |
                         (nv.w == k_full ? 8u : 0u)) << (4 * q);
            }
        }
        ms += __popc(bits);
        if (!gate) bits = (live && w < NT) ? 0xffffu : 0u;      // samples whose rgbsigma is read
        if (w & 1) mw[w >> 1] |= bits << 16; else mw[w >> 1] = bits;
    }
    unsigned tile_act = 0u;                                     // wave-uniform: tiles with a live sample in any of the 64 rays
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const unsigned bits = (w & 1) ? (mw[w >> 1] >> 16) : (mw[w >> 1] & 0xffffu);
        if (__ballot(bits != 0u) != 0ull) tile_act |= 1u << w;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) s_mw[t][k] = mw[k];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    const int sub = t >> 4, col = t & 15;   // staging role: ray-in-group, sample-in-tile
    float4 prs[16];
    float pz[16], pzx[16];
    auto fetch = [&](int w) {               // tile w -> registers (4 rays per instruction, 16 samples x 16 B each)
        const int s0 = 16 * w;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int rr = i * 4 + sub, gr = r0 + rr, gs = s0 + col;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const unsigned word = s_mw[rr][w >> 1];
            if (gr < R && ((word >> (16 * (w & 1) + col)) & 1u)) v = rgbsigma[(size_t)gr * S + gs];
            prs[i] = v;
            pz[i] = z ? (gr < R ? z[(size_t)gr * S + gs] : 0.f) : z_table[gs];
            pzx[i] = 0.f;
            if (col == 0 && s0 + CP_TS < S) pzx[i] = z ? (gr < R ? z[(size_t)gr * S + s0 + CP_TS] : 0.f) : z_table[s0 + CP_TS];
        }
    };
    float T = 1.f, cr = 0.f, cg = 0.f, cb = 0.f, cd = 0.f, ws = 0.f;
    if (tile_act) fetch(__ffs(tile_act) - 1);
    for (int w = 0; w < NT; ++w) {
        const int s0 = 16 * w;
        if (!((tile_act >> w) & 1u)) {
            // nothing to composite in this tile for any of the 64 rays
            if (weights) {
#pragma unroll 4
                for (int i = 0; i < 16; ++i) {
                    const int rr = i * 4 + sub, gr = r0 + rr, gs = s0 + col;
                    if (gr < R) weights[(size_t)gr * S + gs] = 0.f;
                }
            }
            continue;
        }
        // ---- registers -> LDS
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int rr = i * 4 + sub;
            s_rs[rr * CP_PITCH + col] = prs[i];
            s_z[rr * (CP_TS + 1) + col] = pz[i];
            if (col == 0) s_z[rr * (CP_TS + 1) + CP_TS] = pzx[i];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        // ---- the next live tile's values go in flight behind this tile's walk
        const unsigned later = tile_act & ~((2u << w) - 1u);
        if (later) fetch(__ffs(later) - 1);
        // ---- sequential walk of this thread's ray
        if (live) {
#pragma unroll 4
            for (int k = 0; k < CP_TS; ++k) {
                const int sidx = s0 + k;
                const float zc = s_z[t * (CP_TS + 1) + k], zn = s_z[t * (CP_TS + 1) + k + 1];
                const float delta = ((sidx + 1 < S) ? (zn - zc) : 1e10f) * nrm;
                const float4 v = s_rs[t * CP_PITCH + k];
                const float alpha = 1.f - expf(-delta * fmaxf(v.w, 0.f));
                const float wgt = alpha * T;
                T = T * ((1.f - alpha) + 1e-10f);
                cr += wgt * v.x; cg += wgt * v.y; cb += wgt * v.z; cd += wgt * zc; ws += wgt;
                s_w[t * CP_PITCH + k] = wgt;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        // ---- stage out the weights (64 B per ray per instruction)
        if (weights) {
#pragma unroll 4
            for (int i = 0; i < 16; ++i) {
                const int rr = i * 4 + sub, gr = r0 + rr, gs = s0 + col;
                if (gr < R) weights[(size_t)gr * S + gs] = s_w[rr * CP_PITCH + col];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (!live) return;
    if (white_bg) { cr = cr + 1.f - ws; cg = cg + 1.f - ws; cb = cb + 1.f - ws; }
    rgb[3 * (size_t)r] = cr; rgb[3 * (size_t)r + 1] = cg; rgb[3 * (size_t)r + 2] = cb;
    depth[r] = cd;
    opacity[r] = ws;
    if (mask_sum) mask_sum[r] = (float)ms;
}

// Small ray counts (training steps): a wave per ray, as k_composite_bwd_w below — the elementwise work and the global accesses
// on 64 lanes, the recurrence T_i and the five running sums walked by one lane over LDS in sample order (the results are
// those of the thread-per-ray kernel bit for bit; 1 024 rays there are 16 waves on the whole chip for 45 us).
__global__ void __launch_bounds__(256) k_composite_w(const float4* __restrict__ rgbsigma, const float* __restrict__ z,
                                                     const float* __restrict__ z_table, const float* __restrict__ rays,
                                                     const uint8_t* __restrict__ mask, int gate, int R, int S, int white_bg,
                                                     float* __restrict__ rgb, float* __restrict__ depth,
                                                     float* __restrict__ opacity, float* __restrict__ weights,
                                                     float* __restrict__ mask_sum, const int* __restrict__ num_nn, int k_full)
{
    extern __shared__ float cw_lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wv;
    if (r >= R) return;                                  // wave-uniform; no workgroup barrier below
    float* al = cw_lds + (size_t)wv * 6 * S;
    float* vx = al + S; float* vy = vx + S; float* vz = vy + S; float* zc_ = vz + S; float* wv_ = zc_ + S;
    const float* ry = rays + 6 * (size_t)r;
    const float nrm = sqrtf(ry[3] * ry[3] + ry[4] * ry[4] + ry[5] * ry[5]);
    const float* zr = z ? z + (size_t)r * S : z_table;
    const bool have_mask = mask || num_nn;
    int ms = 0;
    for (int s = lane; s < S; s += 64) {
        bool mb = false;
        if (have_mask) mb = mask ? mask[(size_t)r * S + s] != 0 : num_nn[(size_t)r * S + s] == k_full;
        ms += mb ? 1 : 0;
        const bool on = gate ? mb : true;
        const float4 v = on ? rgbsigma[(size_t)r * S + s] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float zc = zr[s];
        const float delta = ((s + 1 < S) ? (zr[s + 1] - zc) : 1e10f) * nrm;
        al[s] = 1.f - expf(-delta * fmaxf(v.w, 0.f));
        vx[s] = v.x; vy[s] = v.y; vz[s] = v.z; zc_[s] = zc;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ms += __shfl_down(ms, o, 64);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        float T = 1.f, cr = 0.f, cg = 0.f, cb = 0.f, cd = 0.f, ws = 0.f;
        for (int s = 0; s < S; ++s) {
            const float alpha = al[s];
            const float w = alpha * T;
            T = T * ((1.f - alpha) + 1e-10f);
            cr += w * vx[s]; cg += w * vy[s]; cb += w * vz[s]; cd += w * zc_[s]; ws += w;
            wv_[s] = w;
        }
        if (white_bg) { cr = cr + 1.f - ws; cg = cg + 1.f - ws; cb = cb + 1.f - ws; }
        rgb[3 * (size_t)r] = cr; rgb[3 * (size_t)r + 1] = cg; rgb[3 * (size_t)r + 2] = cb;
        depth[r] = cd;
        opacity[r] = ws;
        if (mask_sum) mask_sum[r] = (float)ms;
    }
    if (weights) {
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        for (int s = lane; s < S; s += 64) weights[(size_t)r * S + s] = wv_[s];
    }
}

extern "C" int nf_composite_fwd(const float* rgbsigma, const float* z, const float* z_table, const float* rays,
                                const uint8_t* mask, int gate_by_mask, int R, int S, int white_bg, float* rgb, float* depth,
                                float* opacity, float* weights, float* mask_sum, const int32_t* num_nn, int k_full,
                                nf_stream_t stream)
{
    NF_CHECK_ARG(rgbsigma && (z || z_table) && rays && rgb && depth && opacity, "null pointer");
    NF_CHECK_ARG(!gate_by_mask || mask || num_nn, "gate_by_mask needs the mask (or num_nn + k_full)");
    if (R == 0) return NF_OK;
    const size_t lds_w = (size_t)4 * 6 * S * sizeof(float);
    if (R <= 16384 && lds_w <= 64 * 1024) {             // few rays: a wave per ray (see k_composite_w)
        hipLaunchKernelGGL(k_composite_w, dim3((R + 3) / 4), dim3(256), lds_w, (hipStream_t)stream, (const float4*)rgbsigma, z,
                           z_table, rays, mask, gate_by_mask, R, S, white_bg, rgb, depth, opacity, weights, mask_sum, num_nn, k_full);
        NF_CHECK_LAUNCH();
        return NF_OK;
    }
    if (!mask && num_nn && (S & 15) == 0 && S <= 256) {
        hipLaunchKernelGGL(k_composite_p, dim3((R + 63) / 64), dim3(64), 0, (hipStream_t)stream, (const float4*)rgbsigma, z, z_table,
                           rays, gate_by_mask, R, S, white_bg, rgb, depth, opacity, weights, mask_sum, num_nn, k_full);
        NF_CHECK_LAUNCH();
        return NF_OK;
    }
    hipLaunchKernelGGL(k_composite, dim3((R + 63) / 64), dim3(64), 0, (hipStream_t)stream, (const float4*)rgbsigma, z,
                       z_table, rays, mask, gate_by_mask, R, S, white_bg, rgb, depth, opacity, weights, mask_sum, num_nn, k_full,
                       (const float*)nullptr);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_composite_fwd_noise(const float* rgbsigma, const float* z, const float* z_table, const float* rays,
                                      const uint8_t* mask, int gate_by_mask, int R, int S, int white_bg, const float* noise,
                                      float* rgb, float* depth, float* opacity, float* weights, float* mask_sum,
                                      const int32_t* num_nn, int k_full, nf_stream_t stream)
{
    NF_CHECK_ARG(rgbsigma && (z || z_table) && rays && rgb && depth && opacity && noise, "null pointer");
    NF_CHECK_ARG(!gate_by_mask || mask || num_nn, "gate_by_mask needs the mask (or num_nn + k_full)");
    if (R == 0) return NF_OK;
    hipLaunchKernelGGL(k_composite, dim3((R + 63) / 64), dim3(64), 0, (hipStream_t)stream, (const float4*)rgbsigma, z,
                       z_table, rays, mask, gate_by_mask, R, S, white_bg, rgb, depth, opacity, weights, mask_sum, num_nn, k_full, noise);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// ------------------------------------------------------------------------------------------------
// importance sampling (A9), det=True.  One thread per ray; cdf / new samples staged in LDS as
// [k][thread] (conflict-free).  utils/ray_utils.py:178-229.
// ------------------------------------------------------------------------------------------------
#define IS_BLOCK 64
// zero_row (optional): the output row of a ray whose weights[1:-1] are all exactly zero — the same for every such
// ray because the coarse depths are shared (computed once by this very kernel on one zero-weight ray, so the bits
// are those of the general path).  Rays that hit nothing (the large majority of an image) then cost one pass over
// their weights and a coalesced 64-lane copy of that row instead of the serial inverse-CDF walk.
// Rays whose weights[1:-1] are all exactly zero (the large majority of an image) get `zero_row`; the others get a NaN in
// their first output word, which tells k_importance to compute them.  A kernel of its own because the general path needs
// 50 KB of LDS per 64 rays (3 waves per CU): the empty rays are pure streaming work and want the whole chip's occupancy.
__global__ void __launch_bounds__(256) k_importance_zero(const float* __restrict__ w0, int R, int S0, int NI,
                                                         const float* __restrict__ zero_row, float* __restrict__ z1)
{
    const int lane = threadIdx.x & 63;
    const int r0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;       // a wave takes 64 consecutive rays
    if (r0 >= R) return;
    const int ST = S0 + NI, NW = S0 - 2;
    unsigned long long nzm = 0ull;                                    // bit i: ray r0 + i has a non-zero inner weight
    if (S0 == 64 && r0 + 64 <= R) {
        // the 64 weight rows are one contiguous 16 KB run: 16 lanes (float4) per ray, 4 rays per step
        const float4* w4 = (const float4*)(w0 + (size_t)r0 * S0);
#pragma unroll 4
        for (int step = 0; step < 16; ++step) {
            const float4 v = w4[step * 64 + lane];
            const int q = lane & 15;
            const bool nz = (v.y != 0.f) || (v.z != 0.f) || (q != 0 && v.x != 0.f) || (q != 15 && v.w != 0.f);
            const unsigned long long bm = __ballot(nz);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                if ((bm >> (16 * g4)) & 0xffffull) nzm |= 1ull << (step * 4 + g4);
        }
    } else {
        bool nz = false;
        if (r0 + lane < R) {
            const float* wz = w0 + (size_t)(r0 + lane) * S0;
            for (int k = 0; k < NW; ++k) nz = nz || (wz[k + 1] != 0.f);
        }
        nzm = __ballot(nz);
    }
    const int nr = min(64, R - r0);
    for (int i = 0; i < nr; ++i) {
        float* out = z1 + (size_t)(r0 + i) * ST;
        if ((nzm >> i) & 1ull) {
            if (lane == 0) out[0] = __int_as_float(0x7fc00000);
        } else {
            for (int k = lane; k < ST; k += 64) out[k] = zero_row[k];
        }
    }
}

#define IS_PITCH (IS_BLOCK + 1)       // LDS rows [k][thread] with an odd pitch: conflict-free by thread (own column) AND by k (write-out)
__global__ void __launch_bounds__(IS_BLOCK) k_importance(const float* __restrict__ z0, const float* __restrict__ w0,
                                                         const float* __restrict__ u_table, int R, int S0, int NI,
                                                         const float* __restrict__ zero_row, float* __restrict__ z1)
{
    // rows [0, NI): the new samples; rows [NI, NI + S0 - 1): the cdf (dead once the samples exist).  The merge with the coarse
    // depths runs BACKWARDS in place over rows [0, S0 + NI) (row k is written when at most k samples are still unread), and
    // the merged rows leave through a cooperative copy: 64 lanes on consecutive addresses of one ray instead of every thread
    // storing its own row 4 bytes at a time (192 stores, each into 64 different lines).
    extern __shared__ float sm[];
    float* zn = sm;                          // [NI][IS_PITCH], later the merged row [S0 + NI][IS_PITCH]
    float* cdf = sm + NI * IS_PITCH;         // [(S0-1)][IS_PITCH]
    const int tid = threadIdx.x;
    const int r = blockIdx.x * IS_BLOCK + tid;
    const int NB = S0 - 1;   // bins (mid points): 63
    const int NW = S0 - 2;   // weights[1:-1]: 62
    const int ST = S0 + NI;
    bool act = r < R;
    if (zero_row && act) act = isnan(z1[(size_t)r * ST]);      // k_importance_zero ran first: empty rays already hold their row
    if (__ballot(act) == 0ull) return;
    if (act) {
        const float* w = w0 + (size_t)r * S0;
        float tot = 0.f;
        for (int k = 0; k < NW; ++k) tot += (w[k + 1] + 1e-5f);
        float c = 0.f;
        cdf[0 * IS_PITCH + tid] = 0.f;
        for (int k = 0; k < NW; ++k) {
            c += (w[k + 1] + 1e-5f) / tot;
            cdf[(k + 1) * IS_PITCH + tid] = c;
        }
        // inverse CDF; u ascending -> the searchsorted(right=True) pointer only moves forward
        int ind = 0;  // number of cdf entries <= u
        for (int k = 0; k < NI; ++k) {
            float u = u_table[k];
            while (ind < NB && cdf[ind * IS_PITCH + tid] <= u) ++ind;
            int below = ind - 1 < 0 ? 0 : ind - 1;
            int above = ind > NB - 1 ? NB - 1 : ind;
            float c0 = cdf[below * IS_PITCH + tid], c1 = cdf[above * IS_PITCH + tid];
            float b0 = 0.5f * (z0[below + 1] + z0[below]);
            float b1 = 0.5f * (z0[above + 1] + z0[above]);
            float denom = c1 - c0;
            if (denom < 1e-5f) denom = 1.f;
            float t = (u - c0) / denom;
            zn[k * IS_PITCH + tid] = b0 + t * (b1 - b0);
        }
        // torch.sort(cat(z0, z_new)): make z_new sorted (it is, up to rounding), then merge
        for (int k = 1; k < NI; ++k) {
            float v = zn[k * IS_PITCH + tid];
            int m = k;
            while (m > 0 && zn[(m - 1) * IS_PITCH + tid] > v) { zn[m * IS_PITCH + tid] = zn[(m - 1) * IS_PITCH + tid]; --m; }
            zn[m * IS_PITCH + tid] = v;
        }
        // backward merge, in place: a coarse depth goes BEHIND an equal new sample exactly when the forward merge puts it in
        // front (equal values: the rows agree bit for bit either way)
        int a = S0 - 1, b = NI - 1;
        for (int k = ST - 1; k >= 0; --k) {
            const float va = a >= 0 ? z0[a] : -INFINITY;
            const float vb = b >= 0 ? zn[b * IS_PITCH + tid] : -INFINITY;
            if (a < 0 || (b >= 0 && !(va > vb))) { zn[k * IS_PITCH + tid] = vb; --b; } else { zn[k * IS_PITCH + tid] = va; --a; }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    unsigned long long am = __ballot(act);
    while (am) {
        const int src = __ffsll((long long)am) - 1;
        am &= am - 1ull;
        float* out = z1 + (size_t)(blockIdx.x * IS_BLOCK + src) * ST;
        for (int k = tid; k < ST; k += IS_BLOCK) out[k] = zn[k * IS_PITCH + src];
    }
}

// Small ray counts: a wave per ray.  The elementwise parts (pdf terms, the inverse-CDF samples) and the global accesses run on
// the 64 lanes; the order-dependent parts (total, running CDF, insertion sort, merge) are walked by one lane over LDS exactly
// as the thread-per-ray kernel walks them, so the samples are the same bit for bit (the count of CDF entries <= u is found by
// the same forward scan, per sample instead of carried over: the CDF is strictly increasing, the count is the same).
__global__ void __launch_bounds__(256) k_importance_w(const float* __restrict__ z0, const float* __restrict__ w0,
                                                      const float* __restrict__ u_table, int R, int S0, int NI,
                                                      const float* __restrict__ zero_row, float* __restrict__ z1)
{
    extern __shared__ float smw[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wv;
    if (r >= R) return;                                  // wave-uniform; no workgroup barrier below
    const int NB = S0 - 1, NW = S0 - 2, ST = S0 + NI;
    float* pw = smw + (size_t)wv * (S0 + S0 + NI + ST);      // pdf terms w[k+1] + 1e-5 -> (..)/tot
    float* cdf = pw + S0;                                      // [NB]
    float* zn = cdf + S0;                                      // [NI]
    float* mg = zn + NI;                                       // [ST] merged row
    const float* w = w0 + (size_t)r * S0;
    float* out = z1 + (size_t)r * ST;
    bool nz = false;
    for (int k = lane; k < NW; k += 64) { const float v = w[k + 1]; pw[k] = v + 1e-5f; nz = nz || (v != 0.f); }
    if (zero_row && __ballot(nz) == 0ull) {
        for (int k = lane; k < ST; k += 64) out[k] = zero_row[k];
        return;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    float tot = 0.f;
    if (lane == 0) {
        for (int k = 0; k < NW; ++k) tot += pw[k];
    }
    tot = __shfl(tot, 0, 64);
    for (int k = lane; k < NW; k += 64) pw[k] = pw[k] / tot;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        float c = 0.f;
        cdf[0] = 0.f;
        for (int k = 0; k < NW; ++k) { c += pw[k]; cdf[k + 1] = c; }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < NI; k += 64) {
        const float u = u_table[k];
        int ind = 0;
        while (ind < NB && cdf[ind] <= u) ++ind;
        const int below = ind - 1 < 0 ? 0 : ind - 1;
        const int above = ind > NB - 1 ? NB - 1 : ind;
        const float c0 = cdf[below], c1 = cdf[above];
        const float b0 = 0.5f * (z0[below + 1] + z0[below]);
        const float b1 = 0.5f * (z0[above + 1] + z0[above]);
        float denom = c1 - c0;
        if (denom < 1e-5f) denom = 1.f;
        const float t = (u - c0) / denom;
        zn[k] = b0 + t * (b1 - b0);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        for (int k = 1; k < NI; ++k) {
            const float v = zn[k];
            int m = k;
            while (m > 0 && zn[m - 1] > v) { zn[m] = zn[m - 1]; --m; }
            zn[m] = v;
        }
        int a = 0, b = 0;
        for (int k = 0; k < ST; ++k) {
            const float va = a < S0 ? z0[a] : INFINITY;
            const float vb = b < NI ? zn[b] : INFINITY;
            if (b >= NI || (a < S0 && va <= vb)) { mg[k] = va; ++a; } else { mg[k] = vb; ++b; }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < ST; k += 64) out[k] = mg[k];
}

extern "C" int nf_importance_sample(const float* z_table0, const float* weights0, const float* u_table, int R, int S0,
                                    int N_imp, const float* zero_row, float* z1, nf_stream_t stream)
{
    NF_CHECK_ARG(z_table0 && weights0 && u_table && z1, "null pointer");
    NF_CHECK_ARG(S0 >= 3 && N_imp >= 1, "bad S0/N_imp");
    size_t lds = (size_t)(S0 + N_imp) * IS_PITCH * sizeof(float);
    NF_CHECK_ARG(lds <= 160 * 1024, "S0 + N_imp too large for LDS staging");
    if (R == 0) return NF_OK;
    const size_t lds_w = (size_t)4 * (3 * S0 + 2 * N_imp) * sizeof(float);
    if (R <= 16384 && lds_w <= 64 * 1024) {             // few rays: a wave per ray (see k_importance_w)
        hipLaunchKernelGGL(k_importance_w, dim3((R + 3) / 4), dim3(256), lds_w, (hipStream_t)stream, z_table0, weights0, u_table, R,
                           S0, N_imp, zero_row, z1);
        NF_CHECK_LAUNCH();
        return NF_OK;
    }
    if (lds > 64 * 1024)
        hipFuncSetAttribute((const void*)k_importance, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (zero_row)
        hipLaunchKernelGGL(k_importance_zero, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, weights0, R, S0, N_imp,
                           zero_row, z1);
    hipLaunchKernelGGL(k_importance, dim3((R + IS_BLOCK - 1) / IS_BLOCK), dim3(IS_BLOCK), lds, (hipStream_t)stream,
                       z_table0, weights0, u_table, R, S0, N_imp, zero_row, z1);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// ------------------------------------------------------------------------------------------------
// perturb > 0 (utils/ray_utils.py:186-190, :247-252; models/renderer.py:225, :250): per-ray coarse depths jittered inside
// their intervals, and the inverse CDF evaluated at per-ray uniform draws instead of the shared linspace.  The draws come
// from the caller (RenderNet.draw_perturb = torch.rand on the rays' device, in the reference's order), so a test can hand
// the same numbers to the oracle.  Not on any reference caller's path (trainer/basetrainer.py:284-289 never passes it):
// written for exactness, not tuned.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_coarse_perturb(const float* __restrict__ zt, const float* __restrict__ rnd, float perturb,
                                                        int R, int S, float* __restrict__ z)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)R * S) return;
    const int k = (int)(i % S);
    // mid-points of the shared depth table; lower = [z0, mid...], upper = [mid..., z_last]   (:248-251)
    const float zk = zt[k];
    const float lower = k == 0 ? zk : 0.5f * (zt[k - 1] + zk);
    const float upper = k == S - 1 ? zk : 0.5f * (zk + zt[k + 1]);
    const float pr = perturb * rnd[i];
    z[i] = lower + (upper - lower) * pr;          // (-ffp-contract=off: mul, then add, as torch evaluates it)
}

extern "C" int nf_coarse_perturb(const float* z_table, const float* rnd, float perturb, int R, int S, float* z, nf_stream_t stream)
{
    NF_CHECK_ARG(z_table && rnd && z, "null pointer");
    NF_CHECK_ARG(S >= 2 && R >= 0, "bad R/S");
    if (R == 0) return NF_OK;
    const long n = (long)R * S;
    hipLaunchKernelGGL(k_coarse_perturb, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z_table, rnd, perturb, R, S, z);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// A wave per ray, per-ray depths z0[r][S0] and draws u[r][NI] (any order: the samples are sorted with the coarse depths
// afterwards, utils/ray_utils.py:225, so only the multiset matters).  Elementwise parts on the 64 lanes, the order-dependent
// sums (total, running CDF) and the merge walked by one lane in k_importance's order; the NI samples are ordered by a rank
// sort (every lane counts the elements in front of its own: stable, deterministic) instead of a serial insertion sort,
// which is quadratic on unsorted input.
__global__ void __launch_bounds__(256) k_importance_r(const float* __restrict__ z0g, const float* __restrict__ w0,
                                                      const float* __restrict__ ug, int R, int S0, int NI, float* __restrict__ z1)
{
    extern __shared__ float smr[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wv;
    if (r >= R) return;                                  // wave-uniform; no workgroup barrier below
    const int NB = S0 - 1, NW = S0 - 2, ST = S0 + NI;
    float* pw = smr + (size_t)wv * (3 * S0 + 2 * NI + ST);   // pdf terms
    float* cdf = pw + S0;                                      // [NB]
    float* zc = cdf + S0;                                      // [S0] this ray's coarse depths
    float* zn = zc + S0;                                       // [NI] samples as drawn
    float* zs = zn + NI;                                       // [NI] samples sorted
    float* mg = zs + NI;                                       // [ST] merged row
    const float* w = w0 + (size_t)r * S0;
    const float* z0 = z0g + (size_t)r * S0;
    const float* u_row = ug + (size_t)r * NI;
    float* out = z1 + (size_t)r * ST;
    for (int k = lane; k < NW; k += 64) pw[k] = w[k + 1] + 1e-5f;
    for (int k = lane; k < S0; k += 64) zc[k] = z0[k];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    float tot = 0.f;
    if (lane == 0) {
        for (int k = 0; k < NW; ++k) tot += pw[k];
    }
    tot = __shfl(tot, 0, 64);
    for (int k = lane; k < NW; k += 64) pw[k] = pw[k] / tot;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        float c = 0.f;
        cdf[0] = 0.f;
        for (int k = 0; k < NW; ++k) { c += pw[k]; cdf[k + 1] = c; }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < NI; k += 64) {
        const float u = u_row[k];
        int ind = 0;                                      // searchsorted(cdf, u, right=True): entries <= u
        while (ind < NB && cdf[ind] <= u) ++ind;
        const int below = ind - 1 < 0 ? 0 : ind - 1;
        const int above = ind > NB - 1 ? NB - 1 : ind;
        const float c0 = cdf[below], c1 = cdf[above];
        const float b0 = 0.5f * (zc[below + 1] + zc[below]);
        const float b1 = 0.5f * (zc[above + 1] + zc[above]);
        float denom = c1 - c0;
        if (denom < 1e-5f) denom = 1.f;
        const float t = (u - c0) / denom;
        zn[k] = b0 + t * (b1 - b0);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < NI; k += 64) {
        const float v = zn[k];
        int rank = 0;
        for (int j = 0; j < NI; ++j) {
            const float o = zn[j];
            rank += (o < v || (o == v && j < k)) ? 1 : 0;
        }
        zs[rank] = v;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        int a = 0, b = 0;
        for (int k = 0; k < ST; ++k) {
            const float va = a < S0 ? zc[a] : INFINITY;
            const float vb = b < NI ? zs[b] : INFINITY;
            if (b >= NI || (a < S0 && va <= vb)) { mg[k] = va; ++a; } else { mg[k] = vb; ++b; }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < ST; k += 64) out[k] = mg[k];
}

extern "C" int nf_importance_sample_rays(const float* z0, const float* weights0, const float* u, int R, int S0, int N_imp, float* z1,
                                         nf_stream_t stream)
{
    NF_CHECK_ARG(z0 && weights0 && u && z1, "null pointer");
    NF_CHECK_ARG(S0 >= 3 && N_imp >= 1, "bad S0/N_imp");
    const size_t lds = (size_t)4 * (3 * S0 + 2 * N_imp + S0 + N_imp) * sizeof(float);
    NF_CHECK_ARG(lds <= 64 * 1024, "S0 + N_imp too large for LDS staging");
    if (R == 0) return NF_OK;
    hipLaunchKernelGGL(k_importance_r, dim3((R + 3) / 4), dim3(256), lds, (hipStream_t)stream, z0, weights0, u, R, S0, N_imp, z1);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// ------------------------------------------------------------------------------------------------
// composite backward (A12, compositing part): dL/d(rgbsigma) from dL/d(rgb).
//   w_i = a_i T_i,  T_i = prod_{j<i} (1 - a_j + 1e-10),  a_i = 1 - exp(-delta_i relu(sigma_i))
//   rgb = sum w_i c_i (+ 1 - sum w_i)
// One thread per ray: forward sweep rebuilds T_i into `scratch`, reverse sweep carries
// suffix = sum_{k>i} dL/dw_k * w_k.  (z is detached in the reference: utils/ray_utils.py:224.)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_composite_bwd(const float4* __restrict__ rgbsigma, const float* __restrict__ z,
                                                      const float* __restrict__ z_table, const float* __restrict__ rays,
                                                      const float* __restrict__ d_rgb, const uint8_t* __restrict__ mask,
                                                      int gate, int R, int S, int white_bg, float* __restrict__ scratch,
                                                      float4* __restrict__ d_rgbsigma, const int* __restrict__ num_nn, int k_full,
                                                      const float* __restrict__ noise)
{
    int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= R) return;
    const float* ry = rays + 6 * (size_t)r;
    float nrm = sqrtf(ry[3] * ry[3] + ry[4] * ry[4] + ry[5] * ry[5]);
    const float* zr = z ? z + (size_t)r * S : z_table;
    const float g0 = d_rgb[3 * (size_t)r], g1 = d_rgb[3 * (size_t)r + 1], g2 = d_rgb[3 * (size_t)r + 2];
    const float gsum = white_bg ? (g0 + g1 + g2) : 0.f;
    float* Tr = scratch + (size_t)r * S;
    float T = 1.f;
    for (int s = 0; s < S; ++s) {
        float delta = ((s + 1 < S) ? (zr[s + 1] - zr[s]) : 1e10f) * nrm;
        const bool on = !gate || (mask ? mask[(size_t)r * S + s] != 0 : num_nn[(size_t)r * S + s] == k_full);      // rgbsigma is defined (written by the MLP) only there
        float alpha = on ? 1.f - expf(-delta * fmaxf(rgbsigma[(size_t)r * S + s].w + (noise ? noise[(size_t)r * S + s] : 0.f), 0.f)) : 0.f;
        if (noise && !on) alpha = 1.f - expf(-delta * fmaxf(noise[(size_t)r * S + s], 0.f));      // a masked sample: sigma = 0 + noise
        Tr[s] = T;
        T = T * ((1.f - alpha) + 1e-10f);
    }
    float suffix = 0.f;
    for (int s = S - 1; s >= 0; --s) {
        const bool on = !gate || (mask ? mask[(size_t)r * S + s] != 0 : num_nn[(size_t)r * S + s] == k_full);
        float4 v = on ? rgbsigma[(size_t)r * S + s] : make_float4(0.f, 0.f, 0.f, 0.f);
        float delta = ((s + 1 < S) ? (zr[s + 1] - zr[s]) : 1e10f) * nrm;
        const float sg = noise ? v.w + noise[(size_t)r * S + s] : v.w;
        float e = expf(-delta * fmaxf(sg, 0.f));
        float alpha = 1.f - e;
        float Ti = Tr[s];
        float w = alpha * Ti;
        float dw = g0 * v.x + g1 * v.y + g2 * v.z - gsum;
        float dalpha = Ti * dw - suffix / ((1.f - alpha) + 1e-10f);
        suffix += dw * w;
        float dsigma = sg > 0.f ? dalpha * delta * e : 0.f;
        d_rgbsigma[(size_t)r * S + s] = make_float4(w * g0, w * g1, w * g2, dsigma);
    }
}

// Small ray counts (the training steps: 1 024 - 4 096 rays): a thread per ray is a 2 x S chain of dependent, uncoalesced
// global loads per thread and takes ~170 us however few rays there are (16 - 64 waves on the whole chip).  Here a WAVE owns a
// ray: the elementwise parts (exp, alpha, dL/dw, the output rows) run on the 64 lanes with coalesced accesses, and only the
// two recurrences — T_i (forward) and the suffix sum (reverse) — are walked by one lane over LDS, in the same order as above
// (bit-identical results).  4 arrays of S floats per wave in LDS.
__global__ void __launch_bounds__(256) k_composite_bwd_w(const float4* __restrict__ rgbsigma, const float* __restrict__ z,
                                                         const float* __restrict__ z_table, const float* __restrict__ rays,
                                                         const float* __restrict__ d_rgb, const uint8_t* __restrict__ mask,
                                                         int gate, int R, int S, int white_bg,
                                                         float4* __restrict__ d_rgbsigma, const int* __restrict__ num_nn, int k_full)
{
    extern __shared__ float cbw_lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wv;
    if (r >= R) return;                                  // wave-uniform; no workgroup barrier below
    float* fac = cbw_lds + (size_t)wv * 4 * S;          // (1 - alpha_s) + 1e-10
    float* Tv = fac + S;                                 // T_s
    float* dww = Tv + S;                                 // dL/dw_s * w_s
    float* sfx = dww + S;                                // sum_{k>s} dL/dw_k * w_k
    const float* ry = rays + 6 * (size_t)r;
    const float nrm = sqrtf(ry[3] * ry[3] + ry[4] * ry[4] + ry[5] * ry[5]);
    const float* zr = z ? z + (size_t)r * S : z_table;
    const float g0 = d_rgb[3 * (size_t)r], g1 = d_rgb[3 * (size_t)r + 1], g2 = d_rgb[3 * (size_t)r + 2];
    const float gsum = white_bg ? (g0 + g1 + g2) : 0.f;
    for (int s = lane; s < S; s += 64) {
        const float delta = ((s + 1 < S) ? (zr[s + 1] - zr[s]) : 1e10f) * nrm;
        const bool on = !gate || (mask ? mask[(size_t)r * S + s] != 0 : num_nn[(size_t)r * S + s] == k_full);
        const float alpha = on ? 1.f - expf(-delta * fmaxf(rgbsigma[(size_t)r * S + s].w, 0.f)) : 0.f;
        fac[s] = (1.f - alpha) + 1e-10f;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        float T = 1.f;
        for (int s = 0; s < S; ++s) { Tv[s] = T; T = T * fac[s]; }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    for (int s = lane; s < S; s += 64) {
        const bool on = !gate || (mask ? mask[(size_t)r * S + s] != 0 : num_nn[(size_t)r * S + s] == k_full);
        const float4 v = on ? rgbsigma[(size_t)r * S + s] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float delta = ((s + 1 < S) ? (zr[s + 1] - zr[s]) : 1e10f) * nrm;
        const float alpha = 1.f - expf(-delta * fmaxf(v.w, 0.f));
        const float w = alpha * Tv[s];
        const float dw = g0 * v.x + g1 * v.y + g2 * v.z - gsum;
        dww[s] = dw * w;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        float suffix = 0.f;
        for (int s = S - 1; s >= 0; --s) { sfx[s] = suffix; suffix += dww[s]; }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    for (int s = lane; s < S; s += 64) {
        const bool on = !gate || (mask ? mask[(size_t)r * S + s] != 0 : num_nn[(size_t)r * S + s] == k_full);
        const float4 v = on ? rgbsigma[(size_t)r * S + s] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float delta = ((s + 1 < S) ? (zr[s + 1] - zr[s]) : 1e10f) * nrm;
        const float e = expf(-delta * fmaxf(v.w, 0.f));
        const float alpha = 1.f - e;
        const float Ti = Tv[s];
        const float w = alpha * Ti;
        const float dw = g0 * v.x + g1 * v.y + g2 * v.z - gsum;
        const float dalpha = Ti * dw - sfx[s] / ((1.f - alpha) + 1e-10f);
        const float dsigma = v.w > 0.f ? dalpha * delta * e : 0.f;
        d_rgbsigma[(size_t)r * S + s] = make_float4(w * g0, w * g1, w * g2, dsigma);
    }
}

extern "C" int nf_composite_bwd(const float* rgbsigma, const float* z, const float* z_table, const float* rays,
                                const float* d_rgb, const uint8_t* mask, int gate_by_mask, int R, int S, int white_bg,
                                float* scratch, float* d_rgbsigma, const int32_t* num_nn, int k_full, nf_stream_t stream)
{
    NF_CHECK_ARG(rgbsigma && (z || z_table) && rays && d_rgb && scratch && d_rgbsigma, "null pointer");
    NF_CHECK_ARG(!gate_by_mask || mask || num_nn, "gate_by_mask needs the mask (or num_nn + k_full)");
    if (R == 0) return NF_OK;
    const size_t lds_w = (size_t)4 * 4 * S * sizeof(float);
    if (R <= 16384 && lds_w <= 64 * 1024) {             // few rays: a wave per ray (see k_composite_bwd_w)
        hipLaunchKernelGGL(k_composite_bwd_w, dim3((R + 3) / 4), dim3(256), lds_w, (hipStream_t)stream, (const float4*)rgbsigma, z,
                           z_table, rays, d_rgb, mask, gate_by_mask, R, S, white_bg, (float4*)d_rgbsigma, num_nn, k_full);
        NF_CHECK_LAUNCH();
        return NF_OK;
    }
    hipLaunchKernelGGL(k_composite_bwd, dim3((R + 63) / 64), dim3(64), 0, (hipStream_t)stream, (const float4*)rgbsigma, z,
                       z_table, rays, d_rgb, mask, gate_by_mask, R, S, white_bg, scratch, (float4*)d_rgbsigma, num_nn, k_full,
                       (const float*)nullptr);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_composite_bwd_noise(const float* rgbsigma, const float* z, const float* z_table, const float* rays,
                                      const float* d_rgb, const uint8_t* mask, int gate_by_mask, int R, int S, int white_bg,
                                      const float* noise, float* scratch, float* d_rgbsigma, const int32_t* num_nn, int k_full,
                                      nf_stream_t stream)
{
    NF_CHECK_ARG(rgbsigma && (z || z_table) && rays && d_rgb && scratch && d_rgbsigma && noise, "null pointer");
    NF_CHECK_ARG(!gate_by_mask || mask || num_nn, "gate_by_mask needs the mask (or num_nn + k_full)");
    if (R == 0) return NF_OK;
    hipLaunchKernelGGL(k_composite_bwd, dim3((R + 63) / 64), dim3(64), 0, (hipStream_t)stream, (const float4*)rgbsigma, z,
                       z_table, rays, d_rgb, mask, gate_by_mask, R, S, white_bg, scratch, (float4*)d_rgbsigma, num_nn, k_full, noise);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// ------------------------------------------------------------------------------------------------
// A0: per-pixel rays on device (utils/ray_utils.py:85-130): dir_cam = ((i-W/2)/f, -(j-H/2)/f, -1),
// d = normalise(dir_cam @ c2w[:, :3]^T), o = c2w[:, 3].  Rows [row0, row0+nrows) of the image.
// ------------------------------------------------------------------------------------------------
__global__ void k_get_rays(int H, int W, float focal, const float* __restrict__ c2w, int row0, int nrows,
                           float* __restrict__ rays)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nrows * W) return;
    int j = row0 + t / W, i = t % W;
    float dx = ((float)i - (float)W / 2) / focal, dy = -((float)j - (float)H / 2) / focal, dz = -1.f;
    float r[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) r[k] = dx * c2w[4 * k] + dy * c2w[4 * k + 1] + dz * c2w[4 * k + 2];
    float n = sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    float* o = rays + 6 * (size_t)t;
    o[0] = c2w[3]; o[1] = c2w[7]; o[2] = c2w[11];
    o[3] = r[0] / n; o[4] = r[1] / n; o[5] = r[2] / n;
}

extern "C" int nf_get_rays(int H, int W, float focal, const float* c2w, int row0, int nrows, float* rays, nf_stream_t stream)
{
    NF_CHECK_ARG(c2w && rays, "null pointer");
    NF_CHECK_ARG(H > 0 && W > 0 && focal > 0.f && row0 >= 0 && nrows >= 0 && row0 + nrows <= H, "bad image geometry");
    if (nrows == 0) return NF_OK;
    int n = nrows * W;
    hipLaunchKernelGGL(k_get_rays, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, H, W, focal, c2w, row0, nrows, rays);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// The rays of ONE RANK's chunks, in ownership order, in one launch ("rays generated on-device per tile so no ray tensor
// is scattered"): chunk j of this rank is the image's chunk first + j * stride (chunk k -> rank k mod world: first = rank, stride =
// world), a chunk = `chunk` consecutive rays in row-major pixel order; the image's last chunk may be ragged.  Same arithmetic as
// k_get_rays (utils/ray_utils.py:85-130), so a rank's rays are bit-identical to the rows of the full tensor it used to index.
__global__ void k_get_rays_chunks(int H, int W, float focal, const float* __restrict__ c2w, int chunk, int first, int stride,
                                  int n_own_rays, float* __restrict__ rays)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_own_rays) return;
    const long long g = (long long)(first + (t / chunk) * stride) * chunk + t % chunk;      // index of the ray in the image
    int j = (int)(g / W), i = (int)(g % W);
    float dx = ((float)i - (float)W / 2) / focal, dy = -((float)j - (float)H / 2) / focal, dz = -1.f;
    float r[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) r[k] = dx * c2w[4 * k] + dy * c2w[4 * k + 1] + dz * c2w[4 * k + 2];
    float n = sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    float* o = rays + 6 * (size_t)t;
    o[0] = c2w[3]; o[1] = c2w[7]; o[2] = c2w[11];
    o[3] = r[0] / n; o[4] = r[1] / n; o[5] = r[2] / n;
}

extern "C" int nf_get_rays_chunks(int H, int W, float focal, const float* c2w, int chunk, int first, int stride, int n_own_rays,
                                  float* rays, nf_stream_t stream)
{
    NF_CHECK_ARG(c2w && (rays || n_own_rays == 0), "null pointer");
    NF_CHECK_ARG(H > 0 && W > 0 && focal > 0.f && chunk > 0 && first >= 0 && stride > 0 && n_own_rays >= 0, "bad image geometry");
    if (n_own_rays == 0) return NF_OK;
    const long long last = (long long)(first + ((n_own_rays - 1) / chunk) * stride) * chunk + (n_own_rays - 1) % chunk;
    NF_CHECK_ARG(last < (long long)H * W, "the rank's chunks reach beyond the image");
    hipLaunchKernelGGL(k_get_rays_chunks, dim3((n_own_rays + 255) / 256), dim3(256), 0, (hipStream_t)stream, H, W, focal, c2w, chunk,
                       first, stride, n_own_rays, rays);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// ------------------------------------------------------------------------------------------------
// features backward (A12, e2e): dL/d(particles) from dL/d(feature row) through density, smoothed
// position, variance and smoothed direction (the only particle-dependent columns; gradients reach the
// particles ONLY through the gathered neighbour positions — `dists` is used in != 0 tests only and z is
// detached, utils/ray_utils.py:224).  One thread per active row, float atomics into dparticles.
// ------------------------------------------------------------------------------------------------
template <int C, int NF>
__device__ __forceinline__ void pe_backward(const float* __restrict__ g, const float (&v)[C], float (&dv)[C])
{
#pragma unroll
    for (int c = 0; c < C; ++c) {
        float acc = g[c];
        double sd, cd;                       // same double-precision angle doubling as the forward encoding
        sincos((double)v[c], &sd, &cd);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const float fr = (float)(1 << f);
            const float s = (float)sd, co = (float)cd;
            acc += fr * (g[C * (1 + 2 * f) + c] * co - g[C * (2 + 2 * f) + c] * s);
            const double s2 = 2.0 * sd * cd, c2 = 1.0 - 2.0 * sd * sd;
            sd = s2; cd = c2;
        }
        dv[c] = acc;
    }
}

template <int FLAGS>
__global__ void __launch_bounds__(128) k_features_bwd(const float* __restrict__ particles, const float* __restrict__ rays,
                                                      const float* __restrict__ z, const float* __restrict__ z_table,
                                                      int S, float radius, int K, const float* __restrict__ ro_base,
                                                      int ro_stride, const int* __restrict__ row_sample,
                                                      const int* __restrict__ row_nbr, const int* __restrict__ n_rows,
                                                      int max_rows, const float* __restrict__ dX /*row-major*/,
                                                      float* __restrict__ dparticles, int blend)
{
    constexpr int CX = 63 + ((FLAGS & 1) ? 9 : 0) + ((FLAGS & 2) ? 63 : 0) + ((FLAGS & 4) ? 63 : 0);
    constexpr int CD = 27 + ((FLAGS & 8) ? 27 : 0);
    constexpr int OFF_DEN = 63, OFF_SM = OFF_DEN + ((FLAGS & 1) ? 9 : 0), OFF_VAR = OFF_SM + ((FLAGS & 2) ? 63 : 0);
    constexpr int OFF_SDIR = CX + 27;
    int nrows = min(*n_rows, max_rows);
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < nrows; row += gridDim.x * blockDim.x) {
        int sample = row_sample[row];
        float px[3], zv;
        sample_xyz(rays, z, z_table, S, sample, px[0], px[1], px[2], zv);
        const float* ro = ro_base + (size_t)ro_stride * (sample / S);
        const float* g = dX + (size_t)row * (CX + CD);
        // ---- forward recompute
        float sw = 0.f, swp[3] = {0.f, 0.f, 0.f}, sd[3] = {0.f, 0.f, 0.f};
        int nvalid = 0;
        for (int k = 0; k < K; ++k) {
            int j = row_nbr[(size_t)row * K + k];
            float n[3] = {0.f, 0.f, 0.f};
            bool valid = false;
            if (j >= 0) {
                n[0] = particles[3 * (size_t)j]; n[1] = particles[3 * (size_t)j + 1]; n[2] = particles[3 * (size_t)j + 2];
                valid = nf_dist2(px[0], px[1], px[2], n[0], n[1], n[2]) != 0.f;
            }
            float d[3] = {n[0] - px[0], n[1] - px[1], n[2] - px[2]};
            float t = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) / radius;
            float w = fmaxf(1.f - t * t * t, 0.f);
            sw += w; swp[0] += w * n[0]; swp[1] += w * n[1]; swp[2] += w * n[2];
            if (valid) { sd[0] += d[0]; sd[1] += d[1]; sd[2] += d[2]; ++nvalid; }
        }
        const float den = sw + 1e-12f, nn_f = (float)nvalid + 1e-12f;
        const float smw[3] = {swp[0] / den, swp[1] / den, swp[2] / den};      // weighted neighbour mean
        // exclude_ray=False (k_features): the encoded position is ray_pos * (1 - alpha) + smw * alpha; d smw = alpha * d pos
        const float alpha = blend ? ((blend == 2 || nvalid > 20) ? 0.9f : 0.1f) : 1.f;
        float sm[3] = {smw[0], smw[1], smw[2]};
        if (blend) {
            const float oma = 1.f - alpha;
#pragma unroll
            for (int c = 0; c < 3; ++c) sm[c] = px[c] * oma + smw[c] * alpha;
        }
        float mean[3] = {sd[0] / nn_f, sd[1] / nn_f, sd[2] / nn_f};
        float var[3] = {0.f, 0.f, 0.f}, resid[3] = {0.f, 0.f, 0.f};
        if (FLAGS & 4)
            for (int k = 0; k < K; ++k) {
                int j = row_nbr[(size_t)row * K + k];
                if (j < 0) continue;
                float n[3] = {particles[3 * (size_t)j], particles[3 * (size_t)j + 1], particles[3 * (size_t)j + 2]};
                if (nf_dist2(px[0], px[1], px[2], n[0], n[1], n[2]) == 0.f) continue;
#pragma unroll
                for (int c = 0; c < 3; ++c) { float e = (n[c] - px[c]) - mean[c]; var[c] += e * e; resid[c] += e; }
            }
        var[0] /= nn_f; var[1] /= nn_f; var[2] /= nn_f;
        // ---- gradients of the scalar/vector features
        float d_den = 0.f, d_sm[3] = {0.f, 0.f, 0.f}, d_var[3] = {0.f, 0.f, 0.f};
        if (FLAGS & 1) { float v1[1] = {sw}, o1[1]; pe_backward<1, 4>(g + OFF_DEN, v1, o1); d_den = o1[0]; }
        if (FLAGS & 2) pe_backward<3, 10>(g + OFF_SM, sm, d_sm);
        if (FLAGS & 4) pe_backward<3, 10>(g + OFF_VAR, var, d_var);
        if (FLAGS & 8) {
            float u[3] = {sm[0] - ro[0], sm[1] - ro[1], sm[2] - ro[2]};
            float nu = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
            float sdir[3] = {u[0] / nu, u[1] / nu, u[2] / nu}, d_sdir[3];
            pe_backward<3, 4>(g + OFF_SDIR, sdir, d_sdir);
            float dot = sdir[0] * d_sdir[0] + sdir[1] * d_sdir[1] + sdir[2] * d_sdir[2];
#pragma unroll
            for (int c = 0; c < 3; ++c) d_sm[c] += (d_sdir[c] - sdir[c] * dot) / nu;
        }
        // smw = N / den, density = sw
        if (blend) { d_sm[0] *= alpha; d_sm[1] *= alpha; d_sm[2] *= alpha; }
        float dN[3] = {d_sm[0] / den, d_sm[1] / den, d_sm[2] / den};
        float d_sw = d_den - (d_sm[0] * smw[0] + d_sm[1] * smw[1] + d_sm[2] * smw[2]) / den;
        // ---- scatter to the neighbours
        for (int k = 0; k < K; ++k) {
            int j = row_nbr[(size_t)row * K + k];
            if (j < 0) continue;
            float n[3] = {particles[3 * (size_t)j], particles[3 * (size_t)j + 1], particles[3 * (size_t)j + 2]};
            float d[3] = {n[0] - px[0], n[1] - px[1], n[2] - px[2]};
            float dist = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            float t = dist / radius;
            float w = fmaxf(1.f - t * t * t, 0.f);
            float dw = dN[0] * n[0] + dN[1] * n[1] + dN[2] * n[2] + d_sw;
            // dw/dn = -3 t^2 / radius * d/dist  (w > 0), written without the division by dist
            float coef = (1.f - t * t * t) > 0.f ? dw * (-3.f * dist / (radius * radius * radius)) : 0.f;
            float gp[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) gp[c] = w * dN[c] + coef * d[c];
            if ((FLAGS & 4) && nf_dist2(px[0], px[1], px[2], n[0], n[1], n[2]) != 0.f) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    gp[c] += d_var[c] * (2.f / nn_f) * ((d[c] - mean[c]) - resid[c] / nn_f);
            }
            atomicAdd(dparticles + 3 * (size_t)j, gp[0]);
            atomicAdd(dparticles + 3 * (size_t)j + 1, gp[1]);
            atomicAdd(dparticles + 3 * (size_t)j + 2, gp[2]);
        }
    }
}

// Wave per row (K <= 64; every e2e launch: 5 000 - 50 000 rows).  With a thread per row a launch is a few hundred waves,
// each walking 252 strided loads of its dX row, ten double-precision sincos chains and three neighbour loops one after the
// other (203 us for 19 500 rows, however idle the chip).  Here lane k owns neighbour k (gathers, weights, atomics side by side,
// the sums over neighbours as wave reductions), the dX row is read with four coalesced loads into LDS, and the ten encoded
// scalars (density, smoothed position, variance, smoothed direction) are differentiated by ten lanes at once — one sincos
// chain deep instead of ten.  Same formulas as k_features_bwd; the neighbour sums associate differently (tree instead of
// k = 0..K-1), a relative 1e-7 on the recomputed forward values.
// (round 4: the sums over neighbours on the DPP path — nf_wave_sum, nf_common.h; as __shfl_xor butterflies, thirteen per row, they were
// a fifth of this kernel: 78.1 -> 61.7 us per launch)
__device__ __forceinline__ float wave_sum(float v) { return nf_wave_sum(v); }

template <int FLAGS>
__global__ void __launch_bounds__(256) k_features_bwd_w(const float* __restrict__ particles, const float* __restrict__ rays,
                                                        const float* __restrict__ z, const float* __restrict__ z_table,
                                                        int S, float radius, int K, const float* __restrict__ ro_base,
                                                        int ro_stride, const int* __restrict__ row_sample,
                                                        const int* __restrict__ row_nbr, const int* __restrict__ n_rows,
                                                        int max_rows, const float* __restrict__ dX /*row-major*/,
                                                        float* __restrict__ dparticles, int blend)
{
    constexpr int CX = 63 + ((FLAGS & 1) ? 9 : 0) + ((FLAGS & 2) ? 63 : 0) + ((FLAGS & 4) ? 63 : 0);
    constexpr int CD = 27 + ((FLAGS & 8) ? 27 : 0);
    constexpr int OFF_DEN = 63, OFF_SM = OFF_DEN + ((FLAGS & 1) ? 9 : 0), OFF_VAR = OFF_SM + ((FLAGS & 2) ? 63 : 0);
    constexpr int OFF_SDIR = CX + 27;
    __shared__ float gs[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* g = gs[wave];
    const int nrows = min(*n_rows, max_rows);
    for (int row = blockIdx.x * 4 + wave; row < nrows; row += gridDim.x * 4) {
        const int sample = row_sample[row];
        float px[3], zv;
        sample_xyz(rays, z, z_table, S, sample, px[0], px[1], px[2], zv);
        const float* ro = ro_base + (size_t)ro_stride * (sample / S);
        const float* grow = dX + (size_t)row * (CX + CD);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (lane + 64 * u < CX + CD) g[lane + 64 * u] = grow[lane + 64 * u];
        // ---- forward recompute: lane k = neighbour k
        const int j = lane < K ? row_nbr[(size_t)row * K + lane] : -1;
        float n[3] = {0.f, 0.f, 0.f};
        bool valid = false;
        if (j >= 0) {
            n[0] = particles[3 * (size_t)j]; n[1] = particles[3 * (size_t)j + 1]; n[2] = particles[3 * (size_t)j + 2];
            valid = nf_dist2(px[0], px[1], px[2], n[0], n[1], n[2]) != 0.f;
        }
        const float d[3] = {n[0] - px[0], n[1] - px[1], n[2] - px[2]};
        const float dist = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        const float t = dist / radius;
        // padded entries (j < 0) take part in the sums with n = 0, as in the forward (and in k_features_bwd); lanes >= K do not
        const float w = lane < K ? fmaxf(1.f - t * t * t, 0.f) : 0.f;
        const float sw = wave_sum(w);
        const float swp[3] = {wave_sum(w * n[0]), wave_sum(w * n[1]), wave_sum(w * n[2])};
        const float sd[3] = {wave_sum(valid ? d[0] : 0.f), wave_sum(valid ? d[1] : 0.f), wave_sum(valid ? d[2] : 0.f)};
        const int nvalid = __popcll(__ballot(valid));
        const float den = sw + 1e-12f, nn_f = (float)nvalid + 1e-12f;
        const float smw[3] = {swp[0] / den, swp[1] / den, swp[2] / den};
        const float alpha = blend ? ((blend == 2 || nvalid > 20) ? 0.9f : 0.1f) : 1.f;   // exclude_ray=False, see k_features_bwd
        float sm[3] = {smw[0], smw[1], smw[2]};
        if (blend) {
            const float oma = 1.f - alpha;
#pragma unroll
            for (int c = 0; c < 3; ++c) sm[c] = px[c] * oma + smw[c] * alpha;
        }
        const float mean[3] = {sd[0] / nn_f, sd[1] / nn_f, sd[2] / nn_f};
        float var[3] = {0.f, 0.f, 0.f}, resid[3] = {0.f, 0.f, 0.f};
        if (FLAGS & 4) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float e = valid ? d[c] - mean[c] : 0.f;
                var[c] = wave_sum(e * e) / nn_f;
                resid[c] = wave_sum(e);
            }
        }
        float u3[3] = {sm[0] - ro[0], sm[1] - ro[1], sm[2] - ro[2]};
        const float nu = sqrtf(u3[0] * u3[0] + u3[1] * u3[1] + u3[2] * u3[2]);
        const float sdir[3] = {u3[0] / nu, u3[1] / nu, u3[2] / nu};
        // ---- the ten encoded scalars, one per lane: 0 density | 1-3 smoothed position | 4-6 variance | 7-9 smoothed direction
        float acc = 0.f;
        {
            const int grp = lane == 0 ? 0 : (lane < 4 ? 1 : (lane < 7 ? 2 : 3));
            const int c = lane == 0 ? 0 : (lane - 1) % 3;
            const bool on = lane < 10 && ((grp == 0 && (FLAGS & 1)) || (grp == 1 && (FLAGS & 2)) || (grp == 2 && (FLAGS & 4)) ||
                                          (grp == 3 && (FLAGS & 8)));
            const int base = grp == 0 ? OFF_DEN : (grp == 1 ? OFF_SM : (grp == 2 ? OFF_VAR : OFF_SDIR));
            const int C = grp == 0 ? 1 : 3, NF = (grp == 0 || grp == 3) ? 4 : 10;
            const float v = grp == 0 ? sw : (grp == 1 ? sm[c] : (grp == 2 ? var[c] : sdir[c]));
            if (on) {
                acc = g[base + c];
                double sdb, cdb;                 // same double-precision angle doubling as the forward encoding
                sincos((double)v, &sdb, &cdb);
                for (int f = 0; f < NF; ++f) {
                    const float fr = (float)(1 << f);
                    const float sf = (float)sdb, cf = (float)cdb;
                    acc += fr * (g[base + C * (1 + 2 * f) + c] * cf - g[base + C * (2 + 2 * f) + c] * sf);
                    const double s2 = 2.0 * sdb * cdb, c2 = 1.0 - 2.0 * sdb * sdb;
                    sdb = s2; cdb = c2;
                }
            }
        }
        const float d_den = __shfl(acc, 0, 64);
        float d_sm[3] = {__shfl(acc, 1, 64), __shfl(acc, 2, 64), __shfl(acc, 3, 64)};
        const float d_var[3] = {__shfl(acc, 4, 64), __shfl(acc, 5, 64), __shfl(acc, 6, 64)};
        if (FLAGS & 8) {
            const float d_sdir[3] = {__shfl(acc, 7, 64), __shfl(acc, 8, 64), __shfl(acc, 9, 64)};
            const float dot = sdir[0] * d_sdir[0] + sdir[1] * d_sdir[1] + sdir[2] * d_sdir[2];
#pragma unroll
            for (int c = 0; c < 3; ++c) d_sm[c] += (d_sdir[c] - sdir[c] * dot) / nu;
        }
        if (blend) { d_sm[0] *= alpha; d_sm[1] *= alpha; d_sm[2] *= alpha; }
        const float dN[3] = {d_sm[0] / den, d_sm[1] / den, d_sm[2] / den};
        const float d_sw = d_den - (d_sm[0] * smw[0] + d_sm[1] * smw[1] + d_sm[2] * smw[2]) / den;
        // ---- scatter: lane k -> neighbour k
        if (j >= 0) {
            const float dw = dN[0] * n[0] + dN[1] * n[1] + dN[2] * n[2] + d_sw;
            const float coef = (1.f - t * t * t) > 0.f ? dw * (-3.f * dist / (radius * radius * radius)) : 0.f;
            float gp[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) gp[c] = w * dN[c] + coef * d[c];
            if ((FLAGS & 4) && valid) {
#pragma unroll
                for (int c = 0; c < 3; ++c) gp[c] += d_var[c] * (2.f / nn_f) * ((d[c] - mean[c]) - resid[c] / nn_f);
            }
            atomicAdd(dparticles + 3 * (size_t)j, gp[0]);
            atomicAdd(dparticles + 3 * (size_t)j + 1, gp[1]);
            atomicAdd(dparticles + 3 * (size_t)j + 2, gp[2]);
        }
    }
}

extern "C" int nf_render_features_bwd(const float* particles, const float* rays, const float* z, const float* z_table,
                                      int R, int S, float radius, int K, int enc_flags, const float* ro, int ro_per_ray,
                                      const int32_t* row_sample, const int32_t* row_nbr, const int32_t* n_rows,
                                      int max_rows, const float* dX, float* dparticles, nf_stream_t stream)
{
    NF_CHECK_ARG(particles && rays && (z || z_table) && ro && row_sample && row_nbr && n_rows && dX && dparticles,
                 "null pointer");
    NF_CHECK_ARG(enc_flags >= 0 && enc_flags < 64 && (enc_flags >> 4) != 2, "bad enc_flags");
    const int blend = (enc_flags & 16) ? ((enc_flags & 32) ? 2 : 1) : 0;
    enc_flags &= 15;
    if (max_rows <= 0) return NF_OK;
    int blocks = (max_rows + 127) / 128;
    if (blocks > 4096) blocks = 4096;
    int blocks_w = (max_rows + 3) / 4;
    if (blocks_w > 16384) blocks_w = 16384;
    hipStream_t st = (hipStream_t)stream;
#define NF_FB_CASE(F)                                                                                                   \
    case F:                                                                                                             \
        if (K <= 64)                                                                                                    \
            hipLaunchKernelGGL(k_features_bwd_w<F>, dim3(blocks_w), dim3(256), 0, st, particles, rays, z, z_table, S, radius, \
                               K, ro, ro_per_ray ? 3 : 0, row_sample, row_nbr, n_rows, max_rows, dX, dparticles, blend); \
        else                                                                                                            \
            hipLaunchKernelGGL(k_features_bwd<F>, dim3(blocks), dim3(128), 0, st, particles, rays, z, z_table, S, radius, K, \
                               ro, ro_per_ray ? 3 : 0, row_sample, row_nbr, n_rows, max_rows, dX, dparticles, blend);    \
        break;
    switch (enc_flags) {
        NF_FB_CASE(0) NF_FB_CASE(1) NF_FB_CASE(2) NF_FB_CASE(3) NF_FB_CASE(4) NF_FB_CASE(5) NF_FB_CASE(6) NF_FB_CASE(7)
        NF_FB_CASE(8) NF_FB_CASE(9) NF_FB_CASE(10) NF_FB_CASE(11) NF_FB_CASE(12) NF_FB_CASE(13) NF_FB_CASE(14) NF_FB_CASE(15)
    }
#undef NF_FB_CASE
    NF_CHECK_LAUNCH();
    return NF_OK;
}
