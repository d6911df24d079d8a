/*
 * neurofluid_hip.h — C ABI of libneurofluid_hip.so (gfx950 / MI355X).
 *
 * The reference (syguan96/NeuroFluid) has no FFI of its own: its hot path sits behind three
 * Python-level operator boundaries.  This header is the drop-in boundary underneath
 * them; each entry point cites the reference interface it replaces.  All pointers are DEVICE
 * pointers unless marked [host]; the caller owns every buffer (torch caching allocator); the
 * library allocates nothing, enqueues all work on the hipStream_t argument and never synchronises.
 * State it keeps: a thread-local error string, and per launch site a per-device "launch attribute
 * set" flag (the dynamic-LDS limit of a kernel is set once per device; the flags are accessed
 * atomically and setting the attribute twice is harmless, so the entry points are re-entrant).
 * Nothing a result depends on lives in the library.  Return 0 on success, <0 on error
 * (nf_last_error() gives the message).  No torch types cross this boundary.
 */
#ifndef NEUROFLUID_HIP_H
#define NEUROFLUID_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* nf_stream_t; /* hipStream_t */

#define NF_VERSION 100
#define NF_OK 0
#define NF_EINVAL (-1)
#define NF_ELAUNCH (-2)

int nf_version(void);
const char* nf_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Uniform cell grid over a point cloud (shared by renderer and transition model).
 * Replaces the spatial structures inside pytorch3d.ops.ball_query (brute force there) and
 * Open3D FixedRadiusSearch's hash table (reference: models/transmodel.py:86-95).
 * bbox = {xmin,ymin,zmin,xmax,ymax,zmax} [host]; points outside are clamped into border cells
 * (search stays exact).  Cell edge = max(cell, extent/NF_GRID_MAX_DIM).  Within a cell the points
 * are stored in ASCENDING ORIGINAL INDEX (needed for first-K-by-index).
 * ------------------------------------------------------------------------------------------ */
#define NF_GRID_MAX_DIM 128
size_t nf_grid_workspace_bytes(int n_points, float cell, const float bbox[6]);
/* number of cells the grid of (n_points, cell, bbox) has (0 = bad parameters): the same header code the builders run */
int nf_grid_cells(int n_points, float cell, const float bbox[6]);
/* Byte offset, inside a built workspace, of the EXACT axis-aligned bounds of the points: 6 x uint32 (lo xyz, hi xyz) in the
 * order-preserving encoding u = bits(f) ^ (bits(f) >> 31 ? 0xffffffff : 0x80000000), written by nf_grid_build whatever bbox
 * the grid was given (points outside the bbox are clamped into its boundary cells; results do not depend on the bbox).
 * A caller that rebuilds the grid every frame reads them back with data it fetches anyway and passes them, padded by a
 * cell, as the next frame's bbox instead of reducing the cloud on the host side of a device sync. */
size_t nf_grid_points_aabb_offset(void);
/* with_firstk_lists != 0 also builds what the first-K-by-index search and the renderer's classify
 * stage need (per-cell particle AABBs, the dilated index-sorted lists and their chunk boxes);
 * 0 builds the cell lists only — enough for nf_radius_count / nf_radius_fill (the transition
 * model rebuilds its grid every step and never runs a first-K query on it). */
int nf_grid_build(const float* pts /*n*3*/, int n_points, float cell, const float bbox[6],
                  void* grid_ws, size_t grid_ws_bytes, int with_firstk_lists, nf_stream_t stream);

/* pytorch3d.ops.ball_query(p1, p2, radius, K) for one cloud — reference call site
 * models/renderer.py:116-118.  First K points IN INDEX ORDER with sum_d (q_d-p_d)^2 < radius^2
 * (fp32, mul+add, no FMA).  dists2/idx/nn padded with 0 / -1 / 0.  idx is int64 like pytorch3d. */
int nf_ball_query_firstk(const void* grid_ws, const float* pts /*the cloud the grid was built from*/,
                         const float* queries /*nq*3*/, int nq, float radius, int K,
                         float* dists2 /*nq*K*/, int64_t* idx /*nq*K*/, float* nn /*nq*K*3*/,
                         nf_stream_t stream);

/* Open3D FixedRadiusSearch (L2, d^2 <= r^2, optional skip of identical positions) — reference:
 * models/transmodel.py:92 (radius_search_ignore_query_points=True), results read at :136-138.
 * Two calls: counts -> row_splits (inclusive scan done on device), then fill.
 * nf_radius_count writes row_splits[0..nq] (int64, row_splits[nq] = nnz).
 * nf_radius_fill writes the entries whose position is below nnz_capacity (deterministic order:
 * cell-major, ascending point index inside a cell; Open3D's own order is hash-bucket order); a caller
 * that sizes the arrays by a bound instead of row_splits[nq] must compare the two and clamp
 * row_splits before handing it to a consumer. */
int nf_radius_count(const void* grid_ws, const float* queries, int nq, float radius, int ignore_same_pos,
                    int64_t* row_splits /*nq+1*/, void* scan_ws, size_t scan_ws_bytes, nf_stream_t stream);
size_t nf_radius_scan_workspace_bytes(int nq);
int nf_radius_fill(const void* grid_ws, const float* queries, int nq, float radius, int ignore_same_pos,
                   const int64_t* row_splits, int32_t* idx, float* dist2, int64_t nnz_capacity,
                   nf_stream_t stream);
/* The clamp named above, for callers that run against a learnt capacity without a host round trip: total_out[0] (optional) =
 * row_splits[n_rows] as counted (saturated to int32), then row_splits[i] = min(row_splits[i], capacity) in place.  The caller
 * compares total_out with the capacity once its work is enqueued and redoes the call when it was exceeded. */
int nf_csr_clamp(int64_t* row_splits /*n_rows+1*/, int n_rows, int64_t capacity, int32_t* total_out /*or NULL*/,
                 nf_stream_t stream);

/* Exact nearest neighbour of every query among pts (FluidErrors' gt -> prediction metric,
 * utils/point_eval.py:36-58, where the reference calls scipy.spatial.cKDTree(pred).query(gt) on the
 * host).  dist[i] = Euclidean distance in double of the fp32 coordinates, idx[i] (optional) the
 * winner's index (ties: lowest index). */
int nf_nearest(const float* pts, int n_pts, const float* queries, int nq, double* dist /*nq*/,
               int32_t* idx /*nq or NULL*/, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Renderer (RenderNet.forward, models/renderer.py:211-270), fused stages.
 * Sample index = ray * S + s.  "Active rows" are the samples the MLP must evaluate:
 * all-K-slots-filled samples when use_mask (models/renderer.py:233-237), every sample otherwise.
 * ------------------------------------------------------------------------------------------ */

/* A0: rays of image rows [row0, row0+nrows) of an H x W pinhole camera (utils/ray_utils.py:74-131,
 * get_ray_directions + get_rays): rays[t] = (origin xyz, unit direction xyz), t = (j-row0)*W + i.
 * c2w: 3x4 row-major camera-to-world on the device. */
int nf_get_rays(int H, int W, float focal, const float* c2w /*12*/, int row0, int nrows, float* rays /*nrows*W*6*/,
                nf_stream_t stream);
/* The rays of one rank's chunks (image chunk first + j * stride for j = 0, 1, ...; `chunk` consecutive rays each, row-major pixel
 * order), in ownership order, n_own_rays of them in all (the image's last chunk may be ragged): the ray-tile sharding of SURVEY 8e
 * without a full (H*W, 6) tensor per rank.  Bit-identical to the corresponding rows of nf_get_rays' output. */
int nf_get_rays_chunks(int H, int W, float focal, const float* c2w /*3x4 row-major*/, int chunk, int first, int stride,
                       int n_own_rays, float* rays /*n_own_rays*6*/, nf_stream_t stream);


/* A1 + cell test: xyz = o + d*z (utils/ray_utils.py:232-256 / :227); num_nn and mask are cleared
 * for EVERY sample here (the search overwrites the candidates); samples whose 27-cell neighbourhood
 * holds a particle box within the radius are appended to cand[] (count in cand_count[0], must be
 * zeroed by the caller).  The (R*S*4) rgbsigma array is NOT touched by classify/search: with use_mask
 * its value at a sample with mask = 0 is zero by definition (rgbsigma * mask, models/renderer.py:237)
 * and nf_composite_fwd/bwd (gate_by_mask) never read it there; the MLP writes the mask = 1 rows.
 * z_table (S) is used when z == NULL (coarse pass: same depths for every ray). */
int nf_render_classify(const void* grid_ws, const float* rays /*R*6*/, const float* z /*R*S or NULL*/,
                       const float* z_table /*S or NULL*/, int R, int S, float radius, int use_mask,
                       int32_t* num_nn /*R*S*/, uint8_t* mask /*R*S or NULL*/,
                       int32_t* cand /*R*S*/, int32_t* cand_count /*1*/, nf_stream_t stream);

/* A2 + A7: first-K search for every candidate; writes num_nn (and mask when it is not NULL: the mask of a sample is
 * num_nn == K; the fused renderer passes NULL and lets nf_composite_* derive the bit from num_nn), appends active rows:
 * row_sample[row] = sample index, row_nbr[row*K + k] = neighbour index or -1,
 * n_rows[0] = number of active rows (zeroed by the caller). */
int nf_render_search(const void* grid_ws, const float* rays, const float* z, const float* z_table, int R, int S,
                     float radius, int K, int use_mask, const int32_t* cand, const int32_t* cand_count,
                     int32_t* num_nn, uint8_t* mask /*or NULL*/,
                     int32_t* row_sample, int32_t* row_nbr, int32_t* n_rows, nf_stream_t stream);

/* A3 + A4 + A5: local-geometry features + positional encodings for every active row, written in
 * the MLP operand layout X[tile][q][lane][4] (tile = row/32; lane = 32*h + row%32; float e of
 * group q holds feature 8q+4h+e; pos-like features first (padded to 8*QX), then dir-like (8*QD)).
 * enc_flags: bit0 density, bit1 smoothed_pos, bit2 var, bit3 smoothed_dir (models/renderer.py:30-44); bit4 =
 * encoding.exclude_ray=False (models/renderer.py:100-106: the smoothed position is ray_pos * (1 - alpha) + weighted_nn * alpha,
 * alpha = 0.9, or 0.1 where num_nn <= 20), bit5 (with bit4) = encoding.same_smooth_factor (alpha = 0.9 everywhere). */
int nf_render_features(const float* particles /*Np*3*/, const float* rays, const float* z, const float* z_table,
                       int R, int S, float radius, int K, int enc_flags,
                       const float* ro /*3, or R*3 when ro_per_ray (several views batched in one call)*/, int ro_per_ray,
                       const int32_t* row_sample, const int32_t* row_nbr, const int32_t* n_rows, int max_rows,
                       void* X, int x_fp16, nf_stream_t stream);
/* x_fp16 != 0 (operand of nf_nerf_mlp_fwd_h2): Xh[tile][t][lane] x 16 B = the lane's 8 halves of K-step t, i.e.
 * features 16t+4h+e (e < 4) and 16t+8+4h+e, round-to-nearest-even — half the bytes of the fp32 layout. */
/* A12 (feature part, e2e training): dparticles[j] += dL/d(particle j) given dX (row-major, n_rows x (cx+cd)) =
 * dL/d(feature row).  Gradients flow only through the gathered neighbour positions (models/renderer.py:96-109,
 * :163-169); float atomics (order-dependent in the last bits).  dparticles (Np*3) must be zero-initialised. */
int nf_render_features_bwd(const float* particles, const float* rays, const float* z, const float* z_table,
                           int R, int S, float radius, int K, int enc_flags, const float* ro, int ro_per_ray,
                           const int32_t* row_sample, const int32_t* row_nbr, const int32_t* n_rows, int max_rows,
                           const float* dX, float* dparticles, nf_stream_t stream);
int nf_render_feature_dims(int enc_flags, int* cx, int* cd, int* qx, int* qd);

/* A6: NeRF MLP (models/nerf.py:83-124) on fp32 MFMA.  `packed` comes from nf_nerf_pack.
 * Writes rgbsigma[row_sample[row]*4 .. +3] = (r,g,b,sigma) for every active row.
 * If acts != NULL (training) also stores per-row activations for the backward pass:
 * acts[row][NF_ACT_STRIDE] = h1..h8 (8*256) | final (256) | dir hidden (128). */
#define NF_ACT_STRIDE 2432
typedef struct {
    const float* w[12]; /* xyz_encoding_1..8, xyz_encoding_final, dir_encoding, sigma, rgb  ([out][in], torch Linear) */
    const float* b[12];
} nf_nerf_params_t;
size_t nf_nerf_packed_floats(int cx, int cd);
int nf_nerf_pack(const nf_nerf_params_t* params /*[host]*/, int cx, int cd, float* packed, nf_stream_t stream);
int nf_nerf_mlp_fwd(const float* packed, int cx, int cd, const float* X, const int32_t* n_rows, int max_rows,
                    const int32_t* row_sample, float* rgbsigma, float* acts /*or NULL*/, nf_stream_t stream);

/* A6, same fp32 arithmetic with the weight stream shared by the 4 waves of a workgroup through an LDS ring
 * (inference; default encodings 198 + 54 only — nf_nerf_pack_stream reports anything else as an error and the
 * caller keeps nf_nerf_mlp_fwd).  wstream = nf_nerf_pack_stream(packed): the K-steps of `packed` re-ordered into the
 * order of consumption, nf_nerf_stream_floats(cx, cd) floats.  Results differ from nf_nerf_mlp_fwd only by the
 * position of the bias in the fp32 summation order (last instead of first K-step). */
size_t nf_nerf_stream_floats(int cx, int cd);
int nf_nerf_pack_stream(const float* packed, int cx, int cd, float* wstream, nf_stream_t stream);
int nf_nerf_mlp_fwd_l(const float* packed, const float* wstream, int cx, int cd, const float* X,
                      const int32_t* n_rows, int max_rows, const int32_t* row_sample, float* rgbsigma,
                      nf_stream_t stream);
/* The same computation, hand-scheduled (nf_mlp_a.hip: the kernel body is one asm statement generated by gen_mlp_a.py).  Its weight
 * stream carries no padding slots (nf_nerf_pack_stream_a / nf_nerf_stream_a_floats: 1 312 slots of 2 KB for 198 + 54 features);
 * results are bit-identical to nf_nerf_mlp_fwd_l (same K order, bias as the last K-step, the heads' mul/add order and the compiler's
 * expansion of 1 / (1 + expf(-c))).  Inference, default encodings only. */
size_t nf_nerf_stream_a_floats(int cx, int cd);
int nf_nerf_pack_stream_a(const float* packed, int cx, int cd, float* wstream, nf_stream_t stream);
int nf_nerf_mlp_fwd_a(const float* packed, const float* wstream, int cx, int cd, const float* X,
                      const int32_t* n_rows, int max_rows, const int32_t* row_sample, float* rgbsigma, nf_stream_t stream);


/* A6 for small launches (models/nerf.py:83-124, the forward of a training step; nf_mlp_n.hip): one 32-row tile per WORKGROUP, a layer's 8 output blocks split over its 4 waves,
 * activations exchanged through an LDS image — a quarter of nf_nerf_mlp_fwd's per-tile latency (it keeps a tile in one
 * wave for 0.35 ms: a launch costs ceil(tiles / 1024) such rounds however empty the last one is).  packed_n = nf_nerf_pack_n
 * of the standard blob; same operand X; the saved activations (acts, when not NULL) equal nf_nerf_mlp_fwd's BIT FOR BIT; round 6: the
 * sigma / rgb heads are summed by the four waves (a quarter of each head's products per wave, then the four partials) instead of one
 * chain per lane by one wave while three wait, so rgbsigma agrees with nf_nerf_mlp_fwd's to the last bits (1e-6 relative), not bit for bit. */
int nf_nerf_mlp_fwd_n(const float* packed_n, int cx, int cd, const float* X, const int32_t* n_rows, int max_rows,
                      const int32_t* row_sample, float* rgbsigma, float* acts /*or NULL*/, nf_stream_t stream);
/* The training forward proper (round 6): nf_nerf_mlp_fwd_n + the ReLU masks of every hidden unit as BITS — amask[tile][10][4 waves][64 lanes]
 * uint32 (nf_nerf_amask_words(max_rows) words; slot = activation slot 0..7, 9; bit 31 - (16 i + r) of a word = [pre-activation > 0] of the
 * wave's block i, accumulator register r; the view branch's single block: bit 15 - r) — which is all nf_nerf_mlp_bwd_n2 needs of the forward: it reads one word per lane and layer
 * instead of 128 B of saved activations (acts are still written: the weight gradients' operand). */
size_t nf_nerf_amask_words(int max_rows);
int nf_nerf_mlp_fwd_n2(const float* packed_n, int cx, int cd, const float* X, const int32_t* n_rows, int max_rows,
                       const int32_t* row_sample, float* rgbsigma, float* acts, uint32_t* amask, nf_stream_t stream);

/* fp16-MFMA forward of A6 (models/nerf.py:83-124), version 3 (nf_mlp_h2.hip): two 32-row tiles per wave, out-block-major, packed fp16 activations
 * between layers, sigma / rgb heads on the matrix pipe.  Operand X: the fp16 layout written
 * by nf_render_features(x_fp16 = 1)), which must be allocated for an EVEN number of 32-row tiles; its own weight stream. */
size_t nf_nerf_packed_h2_bytes(void);
int nf_nerf_pack_h2(const nf_nerf_params_t* params, int cx, int cd, void* stream_h2, nf_stream_t stream);
int nf_nerf_mlp_fwd_h2(const void* stream_h2, int cx, int cd, const void* X, const int32_t* n_rows, int max_rows,
                       const int32_t* row_sample, float* rgbsigma, nf_stream_t stream);
/* The same forward with bit-identical results as a hand-scheduled instruction stream (nf_mlp_ha.hip; body generated by csrc/gen_mlp_ha.py): the
 * kernel the fp16 path runs.  Same X operand; its weight stream is nf_nerf_pack_h2's WITHOUT the 78 bias K-steps (their value — fp32(hi) + fp32(lo),
 * exact — is the C operand of each output block's first MFMA instead) followed by that bias table: nf_nerf_pack_ha re-packs an h2 stream on the
 * device (stream_ha: nf_nerf_packed_ha_bytes() bytes). */
size_t nf_nerf_packed_ha_bytes(void);
int nf_nerf_pack_ha(const void* stream_h2, void* stream_ha, nf_stream_t stream);
int nf_nerf_mlp_fwd_ha(const void* stream_ha, int cx, int cd, const void* X, const int32_t* n_rows, int max_rows,
                       const int32_t* row_sample, float* rgbsigma, nf_stream_t stream);

/* Split-precision forward of A6 (models/nerf.py:83-124; nf_mlp_s.hip): every operand as hi + lo fp16, three fp16 MFMAs per product, fp32 accumulate —
 * fp32-level accuracy (max-abs <= 2e-4 on RGB vs the fp32 path) at a multiple of the fp32-MFMA kernel's speed.  Takes
 * the fp32 operand layout X of nf_render_features (x_fp16 = 0); its own weight stream.  Inference only. */
size_t nf_nerf_packed_s_bytes(void);
int nf_nerf_pack_s(const nf_nerf_params_t* params, int cx, int cd, void* stream_s, nf_stream_t stream);
int nf_nerf_mlp_fwd_s(const void* stream_s, int cx, int cd, const float* X, const int32_t* n_rows, int max_rows,
                      const int32_t* row_sample, float* rgbsigma, nf_stream_t stream);

/* A5 standalone: Embedding.forward (models/nerf.py:21-38), for callers that drive the reference's modules one by one.
 * out[b][c] = x[b][c]; out[b][C(1+2f)+c] = sin(2^f x[b][c]); out[b][C(2+2f)+c] = cos(2^f x[b][c]), f < n_freqs (logscale
 * freq_bands, :16-17).  Each value is the correctly rounded sin / cos of the reference's exact fp32 argument (one
 * double-precision sincos + angle doubling, as the fused feature kernel does).  nf_embed_bwd: what autograd derives,
 * d_x[b][c] = g[b][c] + sum_f 2^f (g_sin cos - g_cos sin). */
int nf_embed_fwd(const float* x /*n_rows*channels*/, int64_t n_rows, int channels, int n_freqs,
                 float* out /*n_rows*channels*(2 n_freqs+1)*/, nf_stream_t stream);
int nf_embed_bwd(const float* x, const float* d_out, int64_t n_rows, int channels, int n_freqs, float* d_x,
                 nf_stream_t stream);

/* Plain strided fp32 GEMM on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32; exact fp32 products and sums): the dense
 * products that torch autograd runs for the nn.Linear / ContinuousConv layers of models/nerf.py:83-124 and
 * models/transmodel.py:116-131 under loss.backward() (trainer/trainer_e2e.py:219-277) and that are not fused elsewhere:
 * dX = dpre W, dB = relu(x)^T dG, dx = dG B^T.   C[m*ldc + n] (+)= sum_k opA(A[m*sa_m + k*sa_k]) * B[k*sb_k + n*sb_n];
 * one of (sa_m, sa_k) and one of (sb_k, sb_n) must be 1; relu_a: opA = max(., 0); splits > 1: split-K with a
 * deterministic slice reduction through `workspace` (nf_gemm_f32_workspace_floats floats). */
size_t nf_gemm_f32_workspace_floats(int M, int N, int splits);
int nf_gemm_f32(int M, int N, int K, const float* A, int64_t sa_m, int64_t sa_k, int relu_a, const float* B, int64_t sb_k,
                int64_t sb_n, float* C, int64_t ldc, int accumulate, int splits, float* workspace, nf_stream_t stream);

/* A12 (MLP part; what torch autograd derives for models/nerf.py:83-124 under loss.backward(),
 * trainer/trainer_renderer.py:96): data gradient of the MLP on fp32 MFMA with transposed packed weights.
 * Reads d_rgbsigma[row_sample[row]] (gradient w.r.t. the MLP output (rgb after sigmoid, sigma)) and the
 * activations saved by nf_nerf_mlp_fwd; writes, per row, the pre-activation gradients of every layer:
 * dpre[row][NF_DPRE_STRIDE] = dpre_1..8 (8*256) | dpre_final (256) | dpre_dir (128) | dz_rgb (3) | dsigma (1).
 * Weight gradients are then plain GEMMs dW_l = dpre_l^T * input_l over all rows. */
#define NF_DPRE_STRIDE 2436
size_t nf_nerf_packed_bwd_floats(void);
int nf_nerf_pack_bwd(const nf_nerf_params_t* params /*[host]*/, int cx, int cd, float* packed_t, nf_stream_t stream);
int nf_nerf_mlp_bwd(const float* packed, const float* packed_t, int cx, int cd, const float* acts,
                    const int32_t* n_rows, int max_rows, const int32_t* row_sample, const float* rgbsigma,
                    const float* d_rgbsigma, float* dpre, nf_stream_t stream);
/* Weight blobs of the tile-per-workgroup kernels (nf_nerf_mlp_fwd_n / nf_nerf_mlp_bwd_n): the standard blobs of nf_nerf_pack /
 * nf_nerf_pack_bwd re-arranged (same size, out of place) so that one 16-byte load per lane holds a wave's operands of two
 * K-steps.  Re-run after every nf_nerf_pack / nf_nerf_pack_bwd. */
int nf_nerf_pack_n(const float* packed, int cx, int cd, float* packed_n, nf_stream_t stream);
int nf_nerf_pack_bwd_n(const float* packed_t, float* packed_tn, nf_stream_t stream);
/* The same data gradient, bit for bit, with a 32-row tile per WORKGROUP instead of per wave; packed = nf_nerf_pack (the heads),
 * packed_t = nf_nerf_pack_bwd_n (the training steps' launches of a
 * few hundred to a few thousand tiles: a quarter of the per-tile latency, no whole round lost to a partly filled last one). */
int nf_nerf_mlp_bwd_n(const float* packed, const float* packed_t, int cx, int cd, const float* acts,
                      const int32_t* n_rows, int max_rows, const int32_t* row_sample, const float* rgbsigma,
                      const float* d_rgbsigma, float* dpre, nf_stream_t stream);
/* The same again with the ReLU masks taken from nf_nerf_mlp_fwd_n2's mask words instead of the saved activations (same dpre, bit for bit). */
int nf_nerf_mlp_bwd_n2(const float* packed, const float* packed_t, int cx, int cd, const uint32_t* amask,
                       const int32_t* n_rows, int max_rows, const int32_t* row_sample, const float* rgbsigma,
                       const float* d_rgbsigma, float* dpre, nf_stream_t stream);
/* nf_nerf_mlp_bwd_n2 + the gradient of the feature row in the same launch (the end-to-end step's dL/dX, models/nerf.py:85-122 under
 * autograd towards models/renderer.py:55-180): dX[row][0:cx] = dpre_1 W_1[:, :cx] + dpre_5 W_5[:, :cx], dX[row][cx:cx+cd] =
 * dpre_dir W_dir[:, 256:], row-major with pitch cx + cd, for rows < *n_rows — what nf_render_features_bwd scatters to the particles.
 * Replaces three nf_gemm_f32 calls per pass over dpre. */
int nf_nerf_mlp_bwd_n3(const float* packed, const float* packed_t, int cx, int cd, const uint32_t* amask,
                       const int32_t* n_rows, int max_rows, const int32_t* row_sample, const float* rgbsigma,
                       const float* d_rgbsigma, float* dpre, float* dX, nf_stream_t stream);

/* A12 (weight gradients of the nn.Linear layers of models/nerf.py:55-81): all 15 GEMMs dW_l = dpre_l^T * input_l of one NeRF in one batched fp32-MFMA launch +
 * one deterministic slice reduction.  X = the MLP operand of nf_render_features (tile layout) that the forward consumed.
 * dweights = one flat blob holding the 12 weight gradients in nf_nerf_params_t order, each [out][in] row-major
 * (nf_nerf_wgrad_floats floats); workspace = nf_nerf_wgrad_workspace_floats(cx, cd, nslices) floats.
 * dbias (NF_DPRE_STRIDE floats, or NULL) = the column sums of dpre over the n_rows rows = the 12 bias gradients in dpre's
 * column order, added up from the operand slabs the GEMMs stage anyway (no separate pass over dpre). */
size_t nf_nerf_wgrad_floats(int cx, int cd);
size_t nf_nerf_wgrad_workspace_floats(int cx, int cd, int nslices);
int nf_nerf_wgrad(const float* dpre, const float* acts, const float* X, int cx, int cd, int n_rows, int nslices,
                  float* workspace, float* dweights, float* dbias, nf_stream_t stream);
/* The same with the row count in device memory (*n_rows, clamped to max_rows; the grid is sized for nslices slices and the kernels
 * derive nf_nerf_wgrad's slicing for the actual count: bit-identical sums).  For a training step replayed as a HIP graph, whose
 * launch arguments are frozen (trainer/trainer_renderer.py:94-99 is the step). */
int nf_nerf_wgrad_dev(const float* dpre, const float* acts, const float* X, int cx, int cd, const int32_t* n_rows, int max_rows,
                      int nslices, float* workspace, float* dweights, float* dbias, nf_stream_t stream);

/* A8: alpha compositing (models/renderer.py:182-208), one thread per ray, sequential products.
 * gate_by_mask != 0 (use_mask): rgbsigma is read only where mask != 0 and taken as zero elsewhere
 * (bit-identical to compositing rgbsigma * mask); 0: every sample is read.  mask (optional when not
 * gating) also yields mask_sum; weights may be NULL when the caller does not resample from them. */
int nf_composite_fwd(const float* rgbsigma /*R*S*4*/, const float* z, const float* z_table, const float* rays,
                     const uint8_t* mask, int gate_by_mask, int R, int S, int white_bg,
                     float* rgb /*R*3*/, float* depth /*R*/, float* opacity /*R*/, float* weights /*R*S or NULL*/,
                     float* mask_sum /*R or NULL*/, const int32_t* num_nn /*R*S or NULL*/, int k_full, nf_stream_t stream);
/* mask == NULL and num_nn != NULL: the mask bit of a sample is (num_nn[sample] == k_full). */

/* A12 (compositing part): d_rgbsigma (R*S*4) from d_rgb (R*3); scratch = R*S floats.  Depth/opacity are
 * not differentiated (the reference losses use rgb0/rgb1 only: trainer/trainer_renderer.py:127-130). */
int nf_composite_bwd(const float* rgbsigma, const float* z, const float* z_table, const float* rays,
                     const float* d_rgb, const uint8_t* mask, int gate_by_mask, int R, int S, int white_bg,
                     float* scratch /*R*S*/, float* d_rgbsigma /*R*S*4*/, const int32_t* num_nn /*or NULL*/, int k_full,
                     nf_stream_t stream);

/* noise_std > 0 (models/renderer.py:193-196): `noise` (R*S, = noise_std * randn, drawn by the caller in the reference's order: coarse
 * pass, then fine pass) is added to sigma before the ReLU — at EVERY sample: a masked sample's sigma is 0 * mask + noise, so
 * nothing is skipped; rgbsigma is still read only where the mask allows. */
int nf_composite_fwd_noise(const float* rgbsigma, const float* z, const float* z_table, const float* rays, const uint8_t* mask,
                           int gate_by_mask, int R, int S, int white_bg, const float* noise, float* rgb, float* depth,
                           float* opacity, float* weights, float* mask_sum, const int32_t* num_nn, int k_full, nf_stream_t stream);
int nf_composite_bwd_noise(const float* rgbsigma, const float* z, const float* z_table, const float* rays, const float* d_rgb,
                           const uint8_t* mask, int gate_by_mask, int R, int S, int white_bg, const float* noise, float* scratch,
                           float* d_rgbsigma, const int32_t* num_nn, int k_full, nf_stream_t stream);

/* A9: ImportanceSampling(det=True) (utils/ray_utils.py:178-229): z1 = sort(cat(z0, inverse-CDF samples)).
 * u_table = torch.linspace(0,1,N_imp) supplied by the caller (bit-identical to the reference).
 * zero_row (optional, S0+N_imp floats): this function's own output for a ray with all-zero weights
 * (call it once with R = 1, zero weights, zero_row = NULL); rays whose weights[1:-1] are all zero
 * then take a copy of it instead of the serial inverse-CDF walk — same bits, they share z_table0. */
int nf_importance_sample(const float* z_table0 /*S0*/, const float* weights0 /*R*S0*/, const float* u_table /*N_imp*/,
                         int R, int S0, int N_imp, const float* zero_row, float* z1 /*R*(S0+N_imp)*/,
                         nf_stream_t stream);

/* perturb > 0 (models/renderer.py:225, :250).  The uniform draws are the CALLER's (torch.rand in the reference:
 * utils/ray_utils.py:252 for the coarse jitter, :190 for the inverse-CDF u), so that results are reproducible against a seed.
 * nf_coarse_perturb: coarse_sample_ray's perturb branch (utils/ray_utils.py:247-253):
 *   z[r][k] = lower[k] + (upper[k] - lower[k]) * (perturb * rnd[r][k]),  lower / upper from the mid-points of z_table.
 * nf_importance_sample_rays: ImportanceSampling(det=False) (utils/ray_utils.py:178-229) with PER-RAY coarse depths z0 (R*S0)
 *   and per-ray draws u (R*N_imp, any order): z1 = sort(cat(z0, inverse-CDF(u))). */
int nf_coarse_perturb(const float* z_table /*S*/, const float* rnd /*R*S*/, float perturb, int R, int S, float* z /*R*S*/,
                      nf_stream_t stream);
int nf_importance_sample_rays(const float* z0 /*R*S0*/, const float* weights0 /*R*S0*/, const float* u /*R*N_imp*/, int R,
                              int S0, int N_imp, float* z1 /*R*(S0+N_imp)*/, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Transition model (ParticleNet.forward, models/transmodel.py:151-163).
 * ------------------------------------------------------------------------------------------ */

/* B1: integrate_pos_vel (models/transmodel.py:100-104); also emits fluid_feats = [1, vel_new] (:111-114).
 * gravity is a [host] float[3]. */
int nf_trans_integrate(const float* pos, const float* vel, const float gravity[3], float dt, int n,
                       float* pos_new, float* vel_new, float* feats4 /*n*4 or NULL*/, nf_stream_t stream);
/* B1: update_pos_vel with pos_delta = scale * y3 (models/transmodel.py:141, :144-148). */
int nf_trans_update(const float* pos, const float* pos_new, const float* y3, float scale, float dt, int n,
                    float* pos_corrected, float* vel_corrected, nf_stream_t stream);

/* B3 + coordinate mapping of Open3D continuous_conv (ball_to_cube_volume_preserving, align_corners,
 * linear interpolation, 4x4x4 filter), once per CSR entry: pair_w[p*8+c] = window(d2/r^2) * trilinear
 * weight, pair_cell[p*8+c] = (z*4+y)*4+x filter cell.  Shared by every layer that uses the same
 * (inp_positions, out_positions) pair (the reference recomputes it inside each of its 5 convs). */
int nf_cconv_pairs(const float* inp_pos, const float* out_pos, const int64_t* row_splits, const int32_t* nbr,
                   const float* dist2, int n_out, float extent, int use_window,
                   int negate /*1: relative position negated = the pair as seen from the neighbour (backward)*/,
                   float* pair_w /*nnz*8*/, uint8_t* pair_cell /*nnz*8*/, nf_stream_t stream);

/* B4 for Cin in {3,4}, Cout = 32 (conv0_fluid, conv0_obstacle; models/transmodel.py:116,:118):
 * out[i*ld_out + col_off + co] = ContinuousConv(feats)[i][co] + bias[co]; optional Linear branch
 * (dense0_fluid, :117) on self_feats written at dense_col_off.  kernel = (4,4,4,Cin,32) as stored by Open3D. */
int nf_cconv_small(const float* feats, int cin, const int64_t* row_splits, const int32_t* nbr,
                   const float* pair_w, const uint8_t* pair_cell, const float* kernel, const float* bias,
                   int n_out, float* out, int ld_out, int col_off,
                   const float* self_feats, const float* dense_w, const float* dense_b, int dense_col_off,
                   nf_stream_t stream);

/* B4/B5 general layers, step 1: G (M x 65*Cout) = act(A (M x Cin)) * [filter as (Cin x 64*Cout) | dense_w^T]
 * on fp32 MFMA; relu != 0 applies inp_feats = relu(prev) (models/transmodel.py:124). */
int nf_cconv_transform(const float* A, int M, int cin, int cout, int relu, const float* kernel /*(4,4,4,Cin,Cout)*/,
                       const float* dense_w /*[Cout][Cin]*/, float* G, nf_stream_t stream);
/* step 2: y[i] = sum_pairs sum_corners pair_w * G[j][cell] + G[i][dense] + bias_conv + bias_dense (+ residual[i])
 * (models/transmodel.py:125-130) over CSR neighbour rows.  (Training path and oversize clouds; the inference step uses the
 * G-free layers below.) */
int nf_cconv_gather(const float* G, int cout, const int64_t* row_splits, const int32_t* nbr, const float* pair_w,
                    const uint8_t* pair_cell, const float* bias_conv, const float* bias_dense,
                    const float* residual /*n_out*Cout or NULL*/, int n_out, float* out, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * First launch of the inference transition step (nf_trans.hip): B1 + the fluid cell grid of the integrated positions in ONE
 * workgroup (same grid workspace layout as nf_grid_build with with_firstk_lists = 0; bbox as there).  NF_EINVAL when the
 * cloud / grid exceeds nf_trans_prepare_limits.  The rest of the step: nf_trans_front / nf_cconv_gf_layer / nf_cconv3_layer /
 * nf_trans_step at the end of this header.
 * ------------------------------------------------------------------------------------------ */
int nf_trans_prepare_limits(int* max_points, int* max_cells);
int nf_trans_prepare(const float* pos, const float* vel, const float gravity[3], float dt, int n, float cell,
                     const float bbox[6], void* grid_ws, size_t ws_bytes, float* pos_new, float* vel_new, float* feats4,
                     nf_stream_t stream);

/* B8: backward of the continuous convolutions.  Open3D differentiates continuous_conv w.r.t. filter and input
 * features only (not positions); the reference trains through it at trainer/trainer_e2e.py:277.
 * nf_cconv_gather_bwd: dG (n x 65*Cout) from dy (n x Cout) with the TRANSPOSED pair cache (nf_cconv_pairs negate=1;
 *   fluid<->fluid neighbourhoods are symmetric, so the forward CSR serves both directions).  The caller then does
 *   the plain GEMMs  d[filter|dense_w] = x^T dG  and  dx = dG [filter|dense_w]^T.
 * nf_cconv_small_bwd_filter / _feat: filter / input-feature gradient of the direct Cin<=4 convs.  The filter
 *   gradient is dK += A^T dy with the patch matrix A (n_out x 64*cin) built in `workspace`
 *   (nf_cconv_small_bwd_filter_workspace_floats floats); deterministic; dkernel is accumulated into. */
int nf_cconv_gather_bwd(const float* dy, int cout, const int64_t* row_splits, const int32_t* nbr,
                        const float* pair_w_t, const uint8_t* pair_cell_t, int n, float* dG, nf_stream_t stream);
size_t nf_cconv_small_bwd_filter_workspace_floats(int cin, int n_out);
int nf_cconv_small_bwd_filter(const float* feats, int cin, const int64_t* row_splits, const int32_t* nbr,
                              const float* pair_w, const uint8_t* pair_cell, const float* dy, int ld_dy, int col_off,
                              int n_out, float* workspace, float* dkernel /*64*cin*32*/, nf_stream_t stream);
int nf_cconv_small_bwd_feat(const float* kernel, int cin, const int64_t* row_splits, const int32_t* nbr,
                            const float* pair_w_t, const uint8_t* pair_cell_t, const float* dy, int ld_dy, int col_off,
                            int n, float* dfeat /*n*cin*/, nf_stream_t stream);

/* C1 / C2 callers, host side (no device work): the pixel draw np.random.choice(n, size, replace=False) of
 * trainer/trainer_renderer.py:119 and trainer/basetrainer.py:186-190 for numpy's legacy RandomState, callable without the
 * interpreter lock.  key[624] / *pos are the MT19937 state (RandomState.get_state()[1:3]) and are advanced exactly as numpy
 * advances them; out[size] receives permutation(n)[:size].  1 <= n < 2^31. */
int nf_host_choice_mt19937(uint32_t* key /*624, in/out*/, int* pos /*in/out*/, int64_t n, int64_t size, int64_t* out);

/* SSIM of utils/evaluate_images.ipynb cell 5 (class SSIM): 11x11 gaussian window given as its 1-D
 * factor window[11] (normalised), no padding, dynamic range L; pred / gt are B x C x H x W on the device, H, W >= 11.
 * ssim_per_image[b] = mean over channels and valid positions (the notebook's size_average=False result; its default is the
 * mean of these).  workspace: nf_image_ssim_workspace_floats floats. */
size_t nf_image_ssim_workspace_floats(int B, int C, int H, int W);
int nf_image_ssim(const float* pred, const float* gt, int B, int C, int H, int W, const float window[11], float L,
                  float* workspace, float* ssim_per_image, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Round 3: G-free inference step (ParticleNet.forward, models/transmodel.py:151-163; the Open3D ContinuousConv layers of
 * :86-95 evaluated in Open3D's own order: gather a (64 x Cin) patch per output point, then one contraction with the
 * filter on the fp32 matrix pipe — nothing of size n x 64 x C is ever materialised).
 *   nf_trans_front     search of both clouds + the row-entry lists of the fluid pairs + layer 0 (conv0_obstacle, conv0_fluid,
 *                      dense0_fluid, :116-120).  Row entry = {j | cx << 30, w(cx), w(cx + 1)} of a pair in filter row
 *                      (cz + dz) * 4 + (cy + dy); entries[i][roff[i][rho] .. roff[i][rho + 1]) (roff: 20 uint16 per particle).
 *                      Rows keep a pitch (<= nf_trans_front_max_pitch()); overflow2[w] = largest count above its pitch.
 *   nf_cconv_gf_layer  y = cconv(act(x)) + Linear(act(x)) + biases (+ residual) for Cin in {96, 64}, Cout <= 64 (:121-131)
 *                      (relu = 0 when x already holds the activated values, as inside nf_trans_step),
 *                      optionally followed by pos_correction / update_pos_vel (:141-148).  packed = nf_cconv_gf_pack(kernel
 *                      (4,4,4,Cin,Cout), dense_w (Cout,Cin)); scratch = nf_cconv_gf_plan(...) floats; max_wg = CUs.
 *   nf_trans_step      prepare + front + three layers behind one call; the front kernel reports through the pinned words
 *                      host_flag3 (overflow, completion) while the layers are still running. */
int nf_trans_front_max_pitch(void);
int nf_trans_front(const void* fluid_grid, const void* box_grid, const float* queries, const float* fluid_feats,
                   const float* box_feats, int n, float radius, float extent, int use_window, int pitch_fluid,
                   int pitch_box, int32_t* counts2, float* num_fluid_nbrs, int32_t* idx_f, float* d2_f, uint16_t* roff,
                   uint32_t* entries, const float* kernel_fluid, const float* bias_fluid, const float* kernel_obstacle,
                   const float* bias_obstacle, const float* dense_w, const float* dense_b, float* out96,
                   int relu_out /* store max(a0, 0): what conv1 reads (models/transmodel.py:124) */, int64_t* overflow2,
                   int32_t* host_flag3 /* or NULL: device address (nf_pinned_device_ptr) of three pinned host words: [0], [1] are
                                         written with an overflowing count ONLY when a row overflows; [2] = step_id once the
                                         LAST workgroup has finished (the host spins on it: a HIP event recorded in the middle
                                         of a batch of launches completes with the batch, not behind this kernel) */,
                   uint32_t* done_counter /* device word, zero between launches */, int step_id, nf_stream_t stream);
void* nf_pinned_device_ptr(void* host_ptr /*[host] page-locked*/);
/* Diagnostics: the calling thread's pending HIP error code (hipPeekAtLastError; 0 = none), left in place. */
int nf_hip_peek_error(void);
/* [host code] Spin until the pinned host word holds `expected` (0) or `timeout_s` seconds have passed (1): the wait for nf_trans_step's
 * completion word (host_flag3[2]) outside the interpreter — a ctypes call releases the GIL. */
int nf_host_wait_word(const volatile int32_t* word /*[host] page-locked*/, int32_t expected, double timeout_s);

/* The pixel gathers of one training step (trainer/basetrainer.py:186-193: rays[v][ys, xs], rgbs[v][ys, xs] of every view + the view's camera
 * position c2w[v][:, 3] per ray) in one launch.  rays / rgb / c2w: HOST arrays of n_views (<= 16) device pointers to (H*W, 6), (H*W, rgb_c), (3, 4)
 * row-major tensors; flat: n_views * per_view pixel indices in [0, n_pixels) (device, view-major; validated by the caller on the host, where they
 * are drawn — the kernel clamps strays to pixel 0); outputs view-major (n_views * per_view, 6 / rgb_c / 3). */
int nf_gather_view_pixels(int n_views, const float* const* rays, const float* const* rgb, const float* const* c2w, int per_view, int rgb_c,
                          int64_t n_pixels, const int64_t* flat, float* rays_out, float* rgb_out, float* ro_out, nf_stream_t stream);

/* The same gather with the views' pointers in DEVICE memory: table[3 v + {0, 1, 2}] = the 64-bit device addresses of view v's rays, rgb, c2w.
 * A step replayed as a HIP graph bakes its launch arguments; the frame it draws from changes per step (trainer/trainer_e2e.py:189-236
 * walks the sequence's frames), so the caller rewrites the table instead. */
int nf_gather_view_pixels_tab(int n_views, const uint64_t* table /*device, 3 * n_views*/, int per_view, int rgb_c, int64_t n_pixels,
                              const int64_t* flat, float* rays_out, float* rgb_out, float* ro_out, nf_stream_t stream);

/* out_x = x * *scale for up to three contiguous float buffers (any of them empty; out_x == x allowed), *scale one float in device memory: the fused
 * loss's backward scales its three gradients by the upstream gradient in one launch. */
int nf_scale3(const float* a, int64_t na, const float* b, int64_t nb, const float* c, int64_t nc, const float* scale, float* out_a, float* out_b,
              float* out_c, nf_stream_t stream);

/* B8 glue (round 5): the elementwise / reduction steps between the launches of the transition model's backward.
 * nf_relu_bwd_add: out = (prev > 0 ? dx : 0) + (res or 0) — the ReLU in front of a layer back-propagated (+ the residual branch), n floats.
 * nf_colsum: out[c] (and out2[c] if given) = sum_r a[r * lda + c] — bias gradients; deterministic.
 * nf_cconv_split_db: dB (Cin x 65*Cout) -> dK (64, Cin, Cout) filter gradient + dW (Cout, Cin) Linear gradient (models/transmodel.py:116-131). */
int nf_relu_bwd_add(const float* dx, const float* prev, const float* res /*or NULL*/, float* out, int64_t n, nf_stream_t stream);
int nf_colsum(const float* a, int rows, int cols, int lda, float* out, float* out2 /*or NULL*/, nf_stream_t stream);
int nf_cconv_split_db(const float* dB, int cin, int cout, float* dK, float* dW, nf_stream_t stream);
size_t nf_cconv_gf_packed_floats(int cin, int cout);
int nf_cconv_gf_pack(const float* kernel, const float* dense_w, int cin, int cout, float* packed, nf_stream_t stream);
/* split-precision form of the same contraction (nf_cconv_gf_layer(split = 1)): every operand as hi + lo fp16, three fp16 MFMAs
 * per product block, fp32 accumulate — fp32-LEVEL accuracy (products carry 22 bits) on the fp16 matrix pipe, which, unlike the
 * fp32 MFMA on gfx950, does not share its ALUs with the gather arithmetic.  Not the reference's arithmetic: opt-in. */
size_t nf_cconv_gf_packed_split_bytes(int cin, int cout);
int nf_cconv_gf_pack_split(const float* kernel, const float* dense_w, int cin, int cout, void* packed, nf_stream_t stream);
int nf_cconv_gf_plan(int n, int cout, int max_wg, int* tiles, int* nwg, int* maxseg, size_t* scratch_floats);
int nf_cconv_gf_layer(const float* x, int n, int cin, int cout, int relu, const uint16_t* roff, const uint32_t* entries,
                      int pitch, const void* packed, int split /* 0: fp32 MFMA, packed = nf_cconv_gf_pack; 1: nf_cconv_gf_pack_split */,
                      const float* bias_conv, const float* bias_dense,
                      const float* residual, float* out /*or NULL*/, float* out_relu /*max(y, 0), or NULL*/, float* scratch,
                      int max_wg, const float* pos,
                      const float* pos_new, float scale, float dt, float* pos_c, float* vel_c, nf_stream_t stream);
/* The 3-channel last layer (conv3 + dense3, :121-131 at i = 3, and the update of :141-148): transform (G3[j][node][co], 780 bytes
 * per particle) then gather over the row entries.  x_act = relu(a2) (n x 64); workspace = nf_cconv3_workspace_floats(n). */
size_t nf_cconv3_workspace_floats(int n);
size_t nf_cconv3_packed_floats(void);
int nf_cconv3_pack(const float* kernel /*(4,4,4,64,3)*/, const float* dense_w /*(3,64)*/, float* packed, nf_stream_t stream);
int nf_cconv3_layer(const float* x_act, int n, const uint16_t* roff, const uint32_t* entries, int pitch, const float* packed,
                    const float* bias_conv, const float* bias_dense, float* workspace, float* y3,
                    const float* pos /*or NULL: no update*/, const float* pos_new, float scale, float dt, float* pos_c,
                    float* vel_c, nf_stream_t stream);
/* The two halves of nf_cconv3_layer as the fused step runs them: nf_cconv_gf_layer for a 64-channel layer whose epilogue ALSO
 * transforms relu(y) with the packed 3-channel filter (g3 = nf_cconv3_workspace_floats(n) floats; out / out_relu optional — the
 * step passes neither: conv2's output is read by conv3 only), and the gather + update over g3. */
int nf_cconv_gf_layer_g3(const float* x, int n, int cin, int relu, const uint16_t* roff, const uint32_t* entries, int pitch,
                         const void* packed, int split, const float* bias_conv, const float* bias_dense, const float* residual,
                         float* out /*or NULL*/, float* out_relu /*or NULL*/, float* scratch, int max_wg,
                         const float* packed3 /*nf_cconv3_pack*/, float* g3, nf_stream_t stream);
int nf_cconv3_gather(const float* g3, int n, const uint16_t* roff, const uint32_t* entries, int pitch, const float* bias_conv,
                     const float* bias_dense, float* y3, const float* pos /*or NULL: no update*/, const float* pos_new,
                     float scale, float dt, float* pos_c, float* vel_c, nf_stream_t stream);
/* The optimiser step of the training callers (trainer/trainer_renderer.py:96-99, trainer/trainer_e2e.py:277-283: torch.optim.Adam.step):
 * ONE launch over a whole parameter list.  [host] arrays of `count` device pointers / sizes; step_size[t] = lr / (1 - beta1^step_t),
 * bc2_sqrt[t] = sqrt(1 - beta2^step_t) computed by the caller (torch keeps `step` per tensor).  torch.optim.Adam's default
 * arithmetic, operation by operation: g' = g + wd p; m = lerp(m, g', 1 - beta1); v = v beta2 + (1 - beta2) g'^2;
 * p = p - step_size m / (sqrt(v) / bc2_sqrt + eps). */
int nf_adam_step(int count, float* const* params /*[host]*/, const float* const* grads /*[host]*/, float* const* exp_avg /*[host]*/,
                 float* const* exp_avg_sq /*[host]*/, const int64_t* sizes /*[host]*/, const float* step_size /*[host]*/,
                 const float* bc2_sqrt /*[host]*/, double beta1, double beta2 /* (1 - beta) is formed in double, as torch forms its scalars */,
                 float eps, float weight_decay, nf_stream_t stream);
/* The same step for a training step replayed as a HIP graph: sched[0] = step_size, sched[1] = bc2_sqrt of THIS step in device memory
 * (all tensors have taken the same number of steps); *skip != 0 (skip may be NULL) makes the launch a no-op. */
int nf_adam_step_dev(int count, float* const* params /*[host]*/, const float* const* grads /*[host]*/, float* const* exp_avg /*[host]*/,
                     float* const* exp_avg_sq /*[host]*/, const int64_t* sizes /*[host]*/, const float* sched /*[2], device*/,
                     const int32_t* skip /*device or NULL*/, double beta1, double beta2, float eps, float weight_decay, nf_stream_t stream);
/* Overflow bookkeeping of a replayed training step: state (5 device words) = {poisoned, first poisoned step, step counter, count0,
 * count1}; the counter advances per call, the poison word is set — and stays set until the host clears it — when *count0 > cap0 or
 * *count1 > cap1 (either pointer may be NULL).  nf_adam_step_dev's `skip` points at state[0].  host_ring (or NULL): the device
 * address (nf_pinned_device_ptr) of 64 mapped host words = 8 records; step s writes record s & 7, word 2 (= s + 1) last. */
int nf_note_overflow(const int32_t* count0, int cap0, const int32_t* count1, int cap1, int32_t* state, int32_t* host_ring,
                     nf_stream_t stream);

/* nf_note_overflow over up to four (count, capacity) pairs (the end-to-end step: two render passes' rows + the transition step's two
 * pair totals).  counts: HOST array of 4 device pointers (NULL = unused), caps: HOST array of 4; state: 7 device words = {poisoned,
 * first poisoned step, step counter, count[0..3]}; host_ring as above (words 3..6 = the counts). */
int nf_note_overflow4(const int32_t* const* counts /*[host] 4*/, const int32_t* caps /*[host] 4*/, int32_t* state, int32_t* host_ring,
                      nf_stream_t stream);

/* The loss of the end-to-end training step (trainer/trainer_e2e.py:264-280; trainer/basetrainer.py:108-116, :136 for the boundary
 * term) and its gradients for a unit upstream gradient, ONE launch:
 *   loss = (sum (rgb0 - rgb)^2 [+ sum (rgb1 - rgb)^2]) / denom + w_boundary * mean |pos - clamp(pos, lo, hi)|
 * n_rgb values per colour array (views x rays x 3), denom = values per view (the views' MSE means share it); rgb1 / g_rgb1 NULL: no
 * fine pass; n_points = 0: no boundary term.  lo / hi: [host]. */
int nf_e2e_loss(const float* rgb0, const float* rgb1 /*or NULL*/, const float* rgb, int n_rgb, int denom, const float* pos /*(n_points, 3)*/,
                int n_points, const float lo[3] /*[host]*/, const float hi[3] /*[host]*/, float w_boundary, float* loss /*[1]*/,
                float* g_rgb0, float* g_rgb1 /*or NULL*/, float* g_pos, nf_stream_t stream);

typedef struct {
    /* model (device pointers; wpK = nf_cconv_gf_pack of convK / denseK) */
    const float *k_fluid, *b_fluid, *k_obst, *b_obst, *dense0_w, *dense0_b;
    const void *wp1; const float *bc1, *bd1; const void *wp2; const float *bc2, *bd2, *wp3 /*nf_cconv3_pack*/, *bc3, *bd3;
    /* scene */
    const void* box_grid; const float* box_feats;
    /* workspace of one particle count */
    void* grid_ws; size_t grid_ws_bytes;
    float *pos_new, *vel_new, *feats; int32_t* counts2; int32_t* idx_f; float* d2_f; uint16_t* roff; uint32_t* ent;
    float *a0 /*relu(layer 0)*/, *a1, *a1r /*relu(a1)*/, *g3 /*nf_cconv3_workspace_floats(n): conv3's transformed array*/, *y3, *scratch; int64_t* overflow2;
    uint32_t* done_counter;
    int n, pitch_f, pitch_b, use_window, max_wg;
    float radius, extent, dt, scale;
    float gravity[3]; float bbox[6];
    int split;          /* arithmetic of conv1 / conv2: 0 fp32 MFMA (wp1, wp2 from nf_cconv_gf_pack), 1 split fp16 (.._pack_split) */
    int search;         /* fixed-radius search of the fluid cloud: 0 auto (all pairs up to nf_trans_all_pairs_max_points() particles,
                           the cell grid beyond), 1 cell grid (grid_ws, bbox; limits: nf_trans_prepare_limits), 2 all pairs (no grid:
                           grid_ws / bbox unused; rows in ascending neighbour index).  Same neighbour sets and counts either way */
} nf_trans_step_t;
int nf_trans_all_pairs_max_points(void);
int nf_trans_step(const nf_trans_step_t* s /*[host]*/, const float* pos, const float* vel, float* num_fluid_nbrs, float* pos_c,
                  float* vel_c, int32_t* host_flag3 /* as nf_trans_front's; inside the step [2] = step_id is raised by the first layer's launch as it
                                                       starts (the searches and their overflow words [0], [1] are complete then) */, int step_id, nf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NEUROFLUID_HIP_H */
