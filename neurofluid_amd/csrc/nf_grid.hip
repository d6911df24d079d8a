// nf_grid.hip — uniform cell grid, first-K-by-index ball query, fixed-radius CSR search.
//
// Replaces pytorch3d.ops.ball_query (reference call site models/renderer.py:116-118) and Open3D
// FixedRadiusSearch (models/transmodel.py:86-95, :136-138).  Design notes:
//  * cell edge >= search radius, so a query only visits its 27-cell neighbourhood;
//  * cells keep their points in ascending ORIGINAL index (counting sort + in-cell rank), so
//    "first K by index" is a K-bounded sorted insertion with an early break per cell;
//  * the per-thread K-list lives in LDS, laid out [k][thread] so every access is conflict-free.
#include "nf_common.h"
#include <stddef.h>
#include <math.h>
#include <string.h>

// ------------------------------------------------------------------------------------------------
// error string
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void nf_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" int nf_version(void) { return NF_VERSION; }
extern "C" const char* nf_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------------
// generic exclusive scan  int32 counts[n] -> OutT out[n+1]   (3 launches, any n)
// ------------------------------------------------------------------------------------------------
#define SCAN_BLOCK 1024

template <typename OutT>
__device__ __forceinline__ OutT block_exclusive_scan(OutT v, OutT* lds, OutT* total)
{
    // wave scan then cross-wave
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    OutT x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        OutT y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) lds[w] = x;
    __syncthreads();
    if (w == 0) {
        OutT s = lane < (SCAN_BLOCK / 64) ? lds[lane] : 0;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            OutT y = __shfl_up(s, o, 64);
            if (lane >= o) s += y;
        }
        if (lane < (SCAN_BLOCK / 64)) lds[lane] = s;
    }
    __syncthreads();
    OutT base = w ? lds[w - 1] : 0;
    *total = lds[SCAN_BLOCK / 64 - 1];
    return base + x - v;
}

template <typename OutT>
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_local(const int* __restrict__ in, OutT* __restrict__ out,
                                                           OutT* __restrict__ block_sums, int n)
{
    __shared__ OutT lds[SCAN_BLOCK / 64];
    int i = blockIdx.x * SCAN_BLOCK + threadIdx.x;
    OutT v = i < n ? (OutT)in[i] : 0;
    OutT tot;
    OutT ex = block_exclusive_scan<OutT>(v, lds, &tot);
    if (i < n) out[i] = ex;
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

template <typename OutT>
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_sums(OutT* __restrict__ block_sums, int nb, OutT* __restrict__ out, int n)
{
    __shared__ OutT lds[SCAN_BLOCK / 64];
    __shared__ OutT carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += SCAN_BLOCK) {
        int i = base + threadIdx.x;
        OutT v = i < nb ? block_sums[i] : 0;
        OutT tot;
        OutT ex = block_exclusive_scan<OutT>(v, lds, &tot);
        OutT c = carry;
        if (i < nb) block_sums[i] = ex + c;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[n] = carry;
}

template <typename OutT>
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_add(OutT* __restrict__ out, const OutT* __restrict__ block_sums, int n)
{
    int i = blockIdx.x * SCAN_BLOCK + threadIdx.x;
    if (i < n) out[i] += block_sums[blockIdx.x];
}

// small inputs (a particle cloud, a grid of a few thousand cells): one block walks the array with a carry —
// one launch instead of three on the latency-bound transition step
#define SCAN_ITEMS 8
#define SCAN_ROUNDS 4     // SCAN_SINGLE_MAX = SCAN_BLOCK * SCAN_ITEMS * SCAN_ROUNDS (register budget: 128 VGPRs at 1024 threads)
template <typename OutT>
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_single(const int* __restrict__ in, OutT* __restrict__ out, int n)
{
    __shared__ OutT lds[SCAN_BLOCK / 64];
    // all loads are issued up front (one memory latency for the whole array), the rounds then only pay the two
    // block barriers of the scan
    int v[SCAN_ROUNDS][SCAN_ITEMS];
#pragma unroll
    for (int rd = 0; rd < SCAN_ROUNDS; ++rd) {
        const int i0 = (rd * SCAN_BLOCK + threadIdx.x) * SCAN_ITEMS;      // each thread owns SCAN_ITEMS consecutive elements
#pragma unroll
        for (int u = 0; u < SCAN_ITEMS; ++u) v[rd][u] = (i0 + u) < n ? in[i0 + u] : 0;
    }
    OutT carry = 0;
#pragma unroll
    for (int rd = 0; rd < SCAN_ROUNDS; ++rd) {
        if (rd * SCAN_BLOCK * SCAN_ITEMS < n) {      // block-uniform; no `break`, so the loop unrolls and v[][] stays in registers
            const int i0 = (rd * SCAN_BLOCK + threadIdx.x) * SCAN_ITEMS;
            OutT tsum = 0;
#pragma unroll
            for (int u = 0; u < SCAN_ITEMS; ++u) tsum += (OutT)v[rd][u];
            OutT tot;
            OutT run = block_exclusive_scan<OutT>(tsum, lds, &tot) + carry;
#pragma unroll
            for (int u = 0; u < SCAN_ITEMS; ++u) { if (i0 + u < n) out[i0 + u] = run; run += (OutT)v[rd][u]; }
            carry += tot;
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) out[n] = carry;
}

#define SCAN_SINGLE_MAX 32768

template <typename OutT>
static void launch_scan(const int* in, OutT* out, OutT* block_sums, int n, hipStream_t st)
{
    if (n <= SCAN_SINGLE_MAX) {
        hipLaunchKernelGGL(k_scan_single<OutT>, dim3(1), dim3(SCAN_BLOCK), 0, st, in, out, n);
        return;
    }
    int nb = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(k_scan_local<OutT>, dim3(nb), dim3(SCAN_BLOCK), 0, st, in, out, block_sums, n);
    hipLaunchKernelGGL(k_scan_sums<OutT>, dim3(1), dim3(SCAN_BLOCK), 0, st, block_sums, nb, out, n);
    hipLaunchKernelGGL(k_scan_add<OutT>, dim3(nb), dim3(SCAN_BLOCK), 0, st, out, block_sums, n);
}

// ------------------------------------------------------------------------------------------------
// grid build
// ------------------------------------------------------------------------------------------------
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

int nf_grid_make_header(int n, float cell, const float bbox[6], NfGridHeader* h, size_t* total)
{
    if (n < 0 || !(cell > 0.f)) return NF_EINVAL;
    memset(h, 0, sizeof(*h));
    long cells = 1;
    for (int d = 0; d < 3; ++d) {
        float lo = bbox[d], hi = bbox[3 + d];
        if (!(hi >= lo)) return NF_EINVAL;
        float ext = hi - lo;
        float c = cell;
        if (ext / c > (float)(NF_GRID_MAX_DIM - 1)) c = ext / (float)(NF_GRID_MAX_DIM - 1);
        int dim = (int)floorf(ext / c) + 1;
        if (dim < 1) dim = 1;
        if (dim > NF_GRID_MAX_DIM) dim = NF_GRID_MAX_DIM;
        h->origin[d] = lo;
        h->inv_cell[d] = 1.0f / c;
        h->dims[d] = dim;
        h->sub0[d] = 0;
        h->subd[d] = dim;
        cells *= dim;
    }
    h->n_points = n;
    h->n_cells = (int)cells;
    size_t off = align_up(sizeof(NfGridHeader), 256);
    h->off_cell_start = (int)off; off = align_up(off + sizeof(int) * (cells + 1), 256);
    h->off_cell_dil = (int)off;   off = align_up(off + sizeof(int) * cells, 256);
    h->off_sorted_idx = (int)off; off = align_up(off + sizeof(int) * (size_t)(n > 0 ? n : 1), 256);
    h->off_sorted_pos = (int)off; off = align_up(off + sizeof(float4) * (size_t)(n > 0 ? n : 1), 256);
    h->off_tmp_cell = (int)off;   off = align_up(off + sizeof(int) * (size_t)(n > 0 ? n : 1), 256);
    h->off_tmp_list = (int)off;   off = align_up(off + sizeof(int) * (size_t)(n > 0 ? n : 1), 256);
    h->off_cell_fill = (int)off;  off = align_up(off + sizeof(int) * (cells + SCAN_BLOCK + 2048), 256);
    h->off_cell_aabb = (int)off;  off = align_up(off + sizeof(float) * 6 * cells, 256);
    h->off_cell_rec = (int)off;   off = align_up(off + sizeof(float4) * 3 * cells, 256);
    h->off_dil_start = (int)off;  off = align_up(off + sizeof(int) * (cells + 1), 256);
    h->off_dil_pos = (int)off;    off = align_up(off + sizeof(float4) * 27 * (size_t)(n > 0 ? n : 1), 256);
    h->off_dil_box = (int)off;    off = align_up(off + sizeof(float4) * 2 * (27 * (size_t)(n > 0 ? n : 1) / NF_DIL_CHUNK + 2), 256);
    if (off > (size_t)0x7fffffff) return NF_EINVAL;
    *total = off;
    return NF_OK;
}

extern "C" size_t nf_grid_points_aabb_offset(void) { return offsetof(NfGridHeader, pt_lo); }

extern "C" int nf_grid_cells(int n_points, float cell, const float bbox[6])
{
    NfGridHeader h;
    size_t tot = 0;
    if (nf_grid_make_header(n_points, cell, bbox, &h, &tot) != NF_OK) return 0;
    return h.n_cells;
}

extern "C" size_t nf_grid_workspace_bytes(int n_points, float cell, const float bbox[6])
{
    NfGridHeader h;
    size_t tot = 0;
    if (nf_grid_make_header(n_points, cell, bbox, &h, &tot) != NF_OK) return 0;
    return tot;
}

__global__ void k_grid_init(NfGridHeader h, void* ws)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        for (int d = 0; d < 3; ++d) { h.pt_lo[d] = nf_f2ord(INFINITY); h.pt_hi[d] = nf_f2ord(-INFINITY); }
        *(NfGridHeader*)ws = h;
    }
    int* fill = (int*)((char*)ws + h.off_cell_fill);
    if (i < h.n_cells) fill[i] = 0;
}

__global__ void k_grid_count(NfGridHeader h, void* ws, const float* __restrict__ pts)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < h.n_points;
    float p[3] = {0.f, 0.f, 0.f};
    if (live) {
        p[0] = pts[3 * i]; p[1] = pts[3 * i + 1]; p[2] = pts[3 * i + 2];
        int cx = nf_cell_coord(p[0], h.origin[0], h.inv_cell[0], h.dims[0]);
        int cy = nf_cell_coord(p[1], h.origin[1], h.inv_cell[1], h.dims[1]);
        int cz = nf_cell_coord(p[2], h.origin[2], h.inv_cell[2], h.dims[2]);
        int c = (cz * h.dims[1] + cy) * h.dims[0] + cx;
        ((int*)((char*)ws + h.off_tmp_cell))[i] = c;
        atomicAdd((int*)((char*)ws + h.off_cell_fill) + c, 1);
    }
    // exact AABB of the points: wave reduction, then 6 atomics per wave into the header copy in the workspace
    NfGridHeader* hw = (NfGridHeader*)ws;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float lo = live ? p[d] : INFINITY, hi = live ? p[d] : -INFINITY;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o, 64)); hi = fmaxf(hi, __shfl_xor(hi, o, 64)); }
        if ((threadIdx.x & 63) == 0 && lo <= hi) { atomicMin(&hw->pt_lo[d], nf_f2ord(lo)); atomicMax(&hw->pt_hi[d], nf_f2ord(hi)); }
    }
}

// Small clouds (n <= 16 384, n_cells + n <= 39 936: every scene of the configs): the cell lists in ONE workgroup — counters
// and the scatter list in LDS (LDS atomics), block scan, in-cell rank for the stable order — instead of k_grid_init,
// k_grid_count, two scan launches, k_grid_zero_fill, k_grid_scatter and k_grid_rank: seven dependent launches of a few
// microseconds of work each, with 4 913 global atomics on ~100 addresses in two of them (58 -> ~15 us per build).  Same
// outputs: header with the exact point bounds, cell_start, tmp_cell, sorted_idx / sorted_pos in ascending original index.
#define GB_BLOCK 1024
#define GB_MAX_PER_THREAD 16
#define GB_MAX_LDS_INTS 39936
__global__ void __launch_bounds__(GB_BLOCK) k_grid_cells_1wg(NfGridHeader h, void* __restrict__ ws, const float* __restrict__ pts)
{
    extern __shared__ int gb_cells[];       // n_cells counters -> starts -> ends, then the scatter list (n_points)
    __shared__ int s_scan[GB_BLOCK / 64];
    __shared__ float s_lo[GB_BLOCK / 64][3], s_hi[GB_BLOCK / 64][3];
    char* b = (char*)ws;
    int* cell_start = (int*)(b + h.off_cell_start);
    int* tmp_cell = (int*)(b + h.off_tmp_cell);
    int* tmp_list = gb_cells + h.n_cells;
    int* sorted_idx = (int*)(b + h.off_sorted_idx);
    float4* sorted_pos = (float4*)(b + h.off_sorted_pos);
    const int n = h.n_points, nc = h.n_cells, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int c = tid; c < nc; c += GB_BLOCK) gb_cells[c] = 0;
    __syncthreads();
    // ---- cell of every point, counts, exact bounds
    int mycell[GB_MAX_PER_THREAD];
    float px[GB_MAX_PER_THREAD], py[GB_MAX_PER_THREAD], pz[GB_MAX_PER_THREAD];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int u = 0; u < GB_MAX_PER_THREAD; ++u) {
        const int i = u * GB_BLOCK + tid;
        mycell[u] = -1;
        if (i < n) {
            const float p0 = pts[3 * i], p1 = pts[3 * i + 1], p2 = pts[3 * i + 2];
            const int cx = nf_cell_coord(p0, h.origin[0], h.inv_cell[0], h.dims[0]);
            const int cy = nf_cell_coord(p1, h.origin[1], h.inv_cell[1], h.dims[1]);
            const int cz = nf_cell_coord(p2, h.origin[2], h.inv_cell[2], h.dims[2]);
            mycell[u] = (cz * h.dims[1] + cy) * h.dims[0] + cx;
            tmp_cell[i] = mycell[u];
            px[u] = p0; py[u] = p1; pz[u] = p2;
            lo[0] = fminf(lo[0], p0); lo[1] = fminf(lo[1], p1); lo[2] = fminf(lo[2], p2);
            hi[0] = fmaxf(hi[0], p0); hi[1] = fmaxf(hi[1], p1); hi[2] = fmaxf(hi[2], p2);
            atomicAdd(&gb_cells[mycell[u]], 1);
        }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { lo[d] = fminf(lo[d], __shfl_xor(lo[d], o, 64)); hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o, 64)); }
        if (lane == 0) { s_lo[w][d] = lo[d]; s_hi[w][d] = hi[d]; }
    }
    __syncthreads();
    if (tid == 0) {
        for (int d = 0; d < 3; ++d) {
            float l = INFINITY, g = -INFINITY;
            for (int k = 0; k < GB_BLOCK / 64; ++k) { l = fminf(l, s_lo[k][d]); g = fmaxf(g, s_hi[k][d]); }
            h.pt_lo[d] = nf_f2ord(l); h.pt_hi[d] = nf_f2ord(g);
        }
        *(NfGridHeader*)ws = h;
    }
    // ---- exclusive scan of the cell counts (each thread a contiguous run, block scan of the run sums)
    const int per = (nc + GB_BLOCK - 1) / GB_BLOCK;
    const int c0 = tid * per, c1 = min(c0 + per, nc);
    int run = 0;
    for (int c = c0; c < c1; ++c) run += gb_cells[c];
    {
        int x = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
        if (lane == 63) s_scan[w] = x;
        __syncthreads();
        if (w == 0) {
            int sc = lane < GB_BLOCK / 64 ? s_scan[lane] : 0;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { const int y = __shfl_up(sc, o, 64); if (lane >= o) sc += y; }
            if (lane < GB_BLOCK / 64) s_scan[lane] = sc;
        }
        __syncthreads();
        int base = (w ? s_scan[w - 1] : 0) + x - run;
        for (int c = c0; c < c1; ++c) { const int cnt = gb_cells[c]; gb_cells[c] = base; cell_start[c] = base; base += cnt; }
        if (tid == GB_BLOCK - 1) cell_start[nc] = s_scan[GB_BLOCK / 64 - 1];
    }
    __syncthreads();
    // ---- scatter (arrival order); gb_cells[] ends up holding the END of every cell
#pragma unroll
    for (int u = 0; u < GB_MAX_PER_THREAD; ++u)
        if (mycell[u] >= 0) tmp_list[atomicAdd(&gb_cells[mycell[u]], 1)] = u * GB_BLOCK + tid;
    __syncthreads();
    // ---- stable order inside each cell: rank = number of same-cell points with a smaller original index
#pragma unroll
    for (int u = 0; u < GB_MAX_PER_THREAD; ++u) {
        const int c = mycell[u];
        if (c < 0) continue;
        const int i = u * GB_BLOCK + tid;
        const int s = c ? gb_cells[c - 1] : 0, e = gb_cells[c];
        int rank = 0;
        for (int t = s; t < e; ++t) rank += (tmp_list[t] < i);
        sorted_idx[s + rank] = i;
        sorted_pos[s + rank] = make_float4(px[u], py[u], pz[u], __int_as_float(i));
    }
}

__global__ void k_grid_zero_fill(NfGridHeader h, void* ws)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < h.n_cells) ((int*)((char*)ws + h.off_cell_fill))[i] = 0;
}

__global__ void k_grid_scatter(NfGridHeader h, void* ws)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= h.n_points) return;
    char* b = (char*)ws;
    int c = ((int*)(b + h.off_tmp_cell))[i];
    int pos = ((int*)(b + h.off_cell_start))[c] + atomicAdd((int*)(b + h.off_cell_fill) + c, 1);
    ((int*)(b + h.off_tmp_list))[pos] = i;
}

// stable order inside each cell: rank = number of same-cell points with a smaller original index
__global__ void k_grid_rank(NfGridHeader h, void* ws, const float* __restrict__ pts)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= h.n_points) return;
    char* b = (char*)ws;
    const int* list = (const int*)(b + h.off_tmp_list);
    int i = list[t];
    int c = ((const int*)(b + h.off_tmp_cell))[i];
    const int* cs = (const int*)(b + h.off_cell_start);
    int s = cs[c], e = cs[c + 1];
    int rank = 0;
    for (int u = s; u < e; ++u) rank += (list[u] < i);
    int dst = s + rank;
    ((int*)(b + h.off_sorted_idx))[dst] = i;
    ((float4*)(b + h.off_sorted_pos))[dst] = make_float4(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], __int_as_float(i));
}

// A WAVE per cell: the cell's particles are spread over the lanes for its AABB (min / max are exact in any order).  A thread
// per cell left a typical renderer grid (a few hundred cells of ~80 particles) on four waves, each lane walking its cell's
// particles one dependent load after the other: 40 us per build.
__global__ void __launch_bounds__(256) k_grid_dilate(NfGridHeader h, void* ws)
{
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= h.n_cells) return;
    char* b = (char*)ws;
    const int* cs = (const int*)(b + h.off_cell_start);
    const int s0 = cs[c], s1 = cs[c + 1];
    // particle AABB of this cell
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    const float4* sp = (const float4*)(b + h.off_sorted_pos);
    for (int t = s0 + lane; t < s1; t += 64) {
        float4 p = sp[t];
        lo[0] = fminf(lo[0], p.x); lo[1] = fminf(lo[1], p.y); lo[2] = fminf(lo[2], p.z);
        hi[0] = fmaxf(hi[0], p.x); hi[1] = fmaxf(hi[1], p.y); hi[2] = fmaxf(hi[2], p.z);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], o, 64));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], o, 64));
        }
    if (lane != 0) return;
    int cx = c % h.dims[0], cy = (c / h.dims[0]) % h.dims[1], cz = c / (h.dims[0] * h.dims[1]);
    int tot = 0;
    for (int z = max(cz - 1, 0); z <= min(cz + 1, h.dims[2] - 1); ++z)
        for (int y = max(cy - 1, 0); y <= min(cy + 1, h.dims[1] - 1); ++y) {
            int r0 = (z * h.dims[1] + y) * h.dims[0];
            int x0 = max(cx - 1, 0), x1 = min(cx + 1, h.dims[0] - 1);
            tot += cs[r0 + x1 + 1] - cs[r0 + x0];  // cells of one x-row are contiguous
        }
    ((int*)(b + h.off_cell_dil))[c] = tot;
    float* bb = (float*)(b + h.off_cell_aabb) + 6 * c;
    bb[0] = lo[0]; bb[1] = lo[1]; bb[2] = lo[2]; bb[3] = hi[0]; bb[4] = hi[1]; bb[5] = hi[2];
    float4* rec = (float4*)(b + h.off_cell_rec) + 3 * c;
    int mn = s1 > s0 ? ((const int*)(b + h.off_sorted_idx))[s0] : 0x7fffffff;
    rec[0] = make_float4(__int_as_float(s0), __int_as_float(s1), __int_as_float(mn), 0.f);
    rec[1] = make_float4(lo[0], lo[1], lo[2], hi[0]);
    rec[2] = make_float4(hi[1], hi[2], 0.f, 0.f);
}

// Dilated lists: point (sorted slot t) goes into the list of each of the <= 27 cells around its own cell, at the
// rank given by its original index among all points of that target cell's neighbourhood (sum of binary searches
// over the 27 index-sorted source lists) — a 27-way merge without any sorting pass.
__global__ void k_grid_dil_fill(NfGridHeader h, void* ws)
{
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= h.n_points * 27) return;
    int t = gid / 27, o = gid % 27;
    char* b = (char*)ws;
    const int* cs = (const int*)(b + h.off_cell_start);
    const int* sidx = (const int*)(b + h.off_sorted_idx);
    const float4* spos = (const float4*)(b + h.off_sorted_pos);
    const int* tcell = (const int*)(b + h.off_tmp_cell);
    const int me = sidx[t];
    const int c = tcell[me];
    int cx = c % h.dims[0], cy = (c / h.dims[0]) % h.dims[1], cz = c / (h.dims[0] * h.dims[1]);
    int tx = cx + (o % 3) - 1, ty = cy + ((o / 3) % 3) - 1, tz = cz + (o / 9) - 1;
    if (tx < 0 || tx >= h.dims[0] || ty < 0 || ty >= h.dims[1] || tz < 0 || tz >= h.dims[2]) return;
    int rank = 0;
    for (int z = max(tz - 1, 0); z <= min(tz + 1, h.dims[2] - 1); ++z)
        for (int y = max(ty - 1, 0); y <= min(ty + 1, h.dims[1] - 1); ++y)
            for (int x = max(tx - 1, 0); x <= min(tx + 1, h.dims[0] - 1); ++x) {
                int cc = (z * h.dims[1] + y) * h.dims[0] + x;
                int lo = cs[cc], hi = cs[cc + 1];      // lower_bound(me) in the index-sorted list of cell cc
                while (lo < hi) {
                    int mid = (lo + hi) >> 1;
                    if (sidx[mid] < me) lo = mid + 1; else hi = mid;
                }
                rank += lo - cs[cc];
            }
    const int target = (tz * h.dims[1] + ty) * h.dims[0] + tx;
    const int* ds = (const int*)(b + h.off_dil_start);
    ((float4*)(b + h.off_dil_pos))[ds[target] + rank] = spos[t];
}

// AABB of every aligned run of NF_DIL_CHUNK consecutive dilated-list entries (a run may straddle two lists: the
// box is then merely looser, never wrong)
__global__ void k_grid_dil_box(NfGridHeader h, void* ws)
{
    int gch = blockIdx.x * blockDim.x + threadIdx.x;
    char* b = (char*)ws;
    const int total = ((const int*)(b + h.off_dil_start))[h.n_cells];
    if (gch * NF_DIL_CHUNK >= total) return;
    const float4* dp = (const float4*)(b + h.off_dil_pos);
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    const int t1 = min(gch * NF_DIL_CHUNK + NF_DIL_CHUNK, total);
    for (int t = gch * NF_DIL_CHUNK; t < t1; ++t) {
        float4 p = dp[t];
        lo[0] = fminf(lo[0], p.x); lo[1] = fminf(lo[1], p.y); lo[2] = fminf(lo[2], p.z);
        hi[0] = fmaxf(hi[0], p.x); hi[1] = fmaxf(hi[1], p.y); hi[2] = fmaxf(hi[2], p.z);
    }
    float4* box = (float4*)(b + h.off_dil_box) + 2 * gch;
    box[0] = make_float4(lo[0], lo[1], lo[2], 0.f);
    box[1] = make_float4(hi[0], hi[1], hi[2], 0.f);
}

extern "C" int nf_grid_build(const float* pts, int n, float cell, const float bbox[6], void* ws, size_t ws_bytes,
                             int with_firstk_lists, nf_stream_t stream)
{
    NfGridHeader h;
    size_t tot = 0;
    NF_CHECK_ARG(nf_grid_make_header(n, cell, bbox, &h, &tot) == NF_OK, "bad grid parameters");
    NF_CHECK_ARG(ws != nullptr && ws_bytes >= tot, "workspace too small");
    NF_CHECK_ARG(n == 0 || pts != nullptr, "null points");
    hipStream_t st = (hipStream_t)stream;
    const int B = 256;
    int gc = (h.n_cells + B - 1) / B, gp = (n + B - 1) / B;
    if (gp < 1) gp = 1;
    int* fill = (int*)((char*)ws + h.off_cell_fill);
    int* cstart = (int*)((char*)ws + h.off_cell_start);
    if (n > 0 && n <= GB_BLOCK * GB_MAX_PER_THREAD && (long)h.n_cells + n <= GB_MAX_LDS_INTS) {
        static bool attr_set[64] = {};          // per DEVICE: the attribute belongs to the device's copy of the function
        if (nf_first_use_on_device(attr_set))
            hipFuncSetAttribute((const void*)k_grid_cells_1wg, hipFuncAttributeMaxDynamicSharedMemorySize,
                                GB_MAX_LDS_INTS * (int)sizeof(int));
        hipLaunchKernelGGL(k_grid_cells_1wg, dim3(1), dim3(GB_BLOCK), sizeof(int) * ((size_t)h.n_cells + n), st, h, ws, pts);
    } else {
        hipLaunchKernelGGL(k_grid_init, dim3(gc), dim3(B), 0, st, h, ws);
        hipLaunchKernelGGL(k_grid_count, dim3(gp), dim3(B), 0, st, h, ws, pts);
        launch_scan<int>(fill, cstart, fill + h.n_cells + 8, h.n_cells, st);  // block sums live past the fill array
        hipLaunchKernelGGL(k_grid_zero_fill, dim3(gc), dim3(B), 0, st, h, ws);
        hipLaunchKernelGGL(k_grid_scatter, dim3(gp), dim3(B), 0, st, h, ws);
        hipLaunchKernelGGL(k_grid_rank, dim3(gp), dim3(B), 0, st, h, ws, pts);
    }
    if (!with_firstk_lists) {       // radius search only: the cell lists are complete here
        NF_CHECK_LAUNCH();
        return NF_OK;
    }
    hipLaunchKernelGGL(k_grid_dilate, dim3((h.n_cells + 3) / 4), dim3(256), 0, st, h, ws);      // a wave per cell
    launch_scan<int>((const int*)((char*)ws + h.off_cell_dil), (int*)((char*)ws + h.off_dil_start), fill + h.n_cells + 8,
                     h.n_cells, st);
    if (n > 0) {
        hipLaunchKernelGGL(k_grid_dil_fill, dim3((n * 27 + B - 1) / B), dim3(B), 0, st, h, ws);
        const int nchunks = n * 27 / NF_DIL_CHUNK + 1;
        hipLaunchKernelGGL(k_grid_dil_box, dim3((nchunks + B - 1) / B), dim3(B), 0, st, h, ws);
    }
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// ------------------------------------------------------------------------------------------------
// first-K-by-index ball query (standalone op with pytorch3d's output contract)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BQ_BLOCK) k_ball_query(const void* __restrict__ ws, const float* __restrict__ pts,
                                                         const float* __restrict__ q, int nq, float r2, int K,
                                                         float* __restrict__ dists2, int64_t* __restrict__ idx,
                                                         float* __restrict__ nn)
{
    extern __shared__ int lds[];
    int* li = lds;
    int i = blockIdx.x * BQ_BLOCK + threadIdx.x;
    if (i >= nq) return;
    NfGridView g = nf_grid_view(ws);
    float qx = q[3 * i], qy = q[3 * i + 1], qz = q[3 * i + 2];
    unsigned nzmask;
    int cnt = firstk_search(g, qx, qy, qz, r2, K, li, threadIdx.x, nzmask);
    for (int k = 0; k < K; ++k) {
        size_t o = (size_t)i * K + k;
        float d = 0.f, x = 0.f, y = 0.f, z = 0.f;
        int64_t j = -1;
        if (k < cnt) {
            j = li[k * BQ_BLOCK + threadIdx.x];
            x = pts[3 * j]; y = pts[3 * j + 1]; z = pts[3 * j + 2];
            d = nf_dist2(qx, qy, qz, x, y, z);   // identical bits to the value tested during the search
        }
        dists2[o] = d;
        idx[o] = j;
        if (nn) { nn[3 * o] = x; nn[3 * o + 1] = y; nn[3 * o + 2] = z; }
    }
}

extern "C" int nf_ball_query_firstk(const void* ws, const float* pts, const float* queries, int nq, float radius, int K,
                                    float* dists2, int64_t* idx, float* nn, nf_stream_t stream)
{
    NF_CHECK_ARG(ws && pts && (nq == 0 || (queries && dists2 && idx)), "null pointer");
    NF_CHECK_ARG(K >= 1 && K <= 32, "K must be in [1,32]");
    NF_CHECK_ARG(radius > 0.f, "radius must be positive");
    if (nq == 0) return NF_OK;
    hipStream_t st = (hipStream_t)stream;
    const float r2 = radius * radius;  // fp32 product, like pytorch3d's `radius2`
    size_t lds = (size_t)BQ_LDS_INTS(K) * 4;
    hipLaunchKernelGGL(k_ball_query, dim3((nq + BQ_BLOCK - 1) / BQ_BLOCK), dim3(BQ_BLOCK), lds, st, ws, pts, queries, nq,
                       r2, K, dists2, idx, nn);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// ------------------------------------------------------------------------------------------------
// fixed-radius search -> CSR  (Open3D FixedRadiusSearch contract: d2 <= r2, optional skip of points
// at the identical position; models/transmodel.py:92, :136-138)
// ------------------------------------------------------------------------------------------------
// One WAVE per query: the 3 x-adjacent cells of each of the 9 (z, y) rows are one contiguous range of the
// cell-sorted point array, swept 64 candidates at a time; hits are counted with ballot/popcount and placed with
// the lane-prefix popcount, so a row's neighbours keep the cell-major / ascending order.  (A thread per query
// leaves 4 913 queries on 77 waves — a quarter of the CUs, each with one latency-bound wave.)
#define RS_QPB 4     // queries (waves) per block
template <bool FILL>
__global__ void __launch_bounds__(64 * RS_QPB) k_radius(const void* __restrict__ ws, const float* __restrict__ q, int nq, float r2,
                                                        int ignore_same, int* __restrict__ counts,
                                                        const int64_t* __restrict__ row_splits, int32_t* __restrict__ idx,
                                                        float* __restrict__ dist2, int64_t cap)
{
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * RS_QPB + (threadIdx.x >> 6);
    if (i >= nq) return;
    NfGridView g = nf_grid_view(ws);
    const float qx = q[3 * i], qy = q[3 * i + 1], qz = q[3 * i + 2];
    const int cx = nf_cell_coord(qx, g.ox, g.icx, g.dx);
    const int cy = nf_cell_coord(qy, g.oy, g.icy, g.dy);
    const int cz = nf_cell_coord(qz, g.oz, g.icz, g.dz);
    int cnt = 0;
    const int64_t o = FILL ? row_splits[i] : 0;
    if (FILL && o >= cap) return;           // nothing of this row fits (the caller detects row_splits[nq] > capacity)
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int z = max(cz - 1, 0); z <= min(cz + 1, g.dz - 1); ++z)
        for (int y = max(cy - 1, 0); y <= min(cy + 1, g.dy - 1); ++y) {
            const int r0 = (z * g.dy + y) * g.dx;
            const int s = g.cell_start[r0 + max(cx - 1, 0)], e = g.cell_start[r0 + min(cx + 1, g.dx - 1) + 1];
            for (int t0 = s; t0 < e; t0 += 64) {
                const int t = t0 + lane;
                bool hit = false;
                float d2 = 0.f;
                int pi = 0;
                if (t < e) {
                    const float4 p = g.sorted_pos[t];
                    d2 = nf_dist2(qx, qy, qz, p.x, p.y, p.z);
                    hit = d2 <= r2 && !(ignore_same && p.x == qx && p.y == qy && p.z == qz);
                    pi = __float_as_int(p.w);
                }
                const unsigned long long m = __ballot(hit);
                if (FILL && hit) {
                    const int64_t w = o + cnt + __popcll(m & lt);
                    if (w < cap) { idx[w] = pi; dist2[w] = d2; }
                }
                cnt += __popcll(m);
            }
        }
    if (!FILL && lane == 0) counts[i] = cnt;
}

extern "C" size_t nf_radius_scan_workspace_bytes(int nq)
{
    size_t nb = (size_t)(nq + SCAN_BLOCK - 1) / SCAN_BLOCK + 1;
    return align_up(sizeof(int) * (size_t)(nq > 0 ? nq : 1), 256) + align_up(sizeof(int64_t) * nb, 256);
}

extern "C" int nf_radius_count(const void* ws, const float* queries, int nq, float radius, int ignore_same_pos,
                               int64_t* row_splits, void* scan_ws, size_t scan_ws_bytes, nf_stream_t stream)
{
    NF_CHECK_ARG(ws && row_splits && scan_ws, "null pointer");
    NF_CHECK_ARG(nq >= 0 && radius > 0.f, "bad nq/radius");
    NF_CHECK_ARG(scan_ws_bytes >= nf_radius_scan_workspace_bytes(nq), "scan workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int* counts = (int*)scan_ws;
    int64_t* sums = (int64_t*)((char*)scan_ws + align_up(sizeof(int) * (size_t)(nq > 0 ? nq : 1), 256));
    const float r2 = radius * radius;
    if (nq > 0)
        hipLaunchKernelGGL(k_radius<false>, dim3((nq + RS_QPB - 1) / RS_QPB), dim3(64 * RS_QPB), 0, st, ws, queries, nq, r2, ignore_same_pos,
                           counts, (const int64_t*)nullptr, (int32_t*)nullptr, (float*)nullptr, (int64_t)0);
    launch_scan<int64_t>(counts, row_splits, sums, nq, st);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// row_splits of a search whose arrays were sized by a BOUND (a capacity learnt from earlier calls) instead of row_splits[nq]:
// report the true total, then clamp every split to the capacity, so that every consumer of the CSR stays inside the arrays
// (rows beyond the capacity are truncated; the caller compares total with the capacity and redoes the call when it was exceeded).
__global__ void __launch_bounds__(256) k_csr_clamp(int64_t* __restrict__ rs, int n_rows, int64_t cap, int32_t* __restrict__ total)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i > n_rows) return;
    const int64_t v = rs[i];
    if (i == n_rows && total) total[0] = v > 0x7fffffff ? 0x7fffffff : (int32_t)v;
    if (v > cap) rs[i] = cap;
}

extern "C" int nf_csr_clamp(int64_t* row_splits, int n_rows, int64_t capacity, int32_t* total_out, nf_stream_t stream)
{
    NF_CHECK_ARG(row_splits && n_rows >= 0 && capacity >= 0, "bad arguments");
    hipLaunchKernelGGL(k_csr_clamp, dim3((n_rows + 1 + 255) / 256), dim3(256), 0, (hipStream_t)stream, row_splits, n_rows, capacity, total_out);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_radius_fill(const void* ws, const float* queries, int nq, float radius, int ignore_same_pos,
                              const int64_t* row_splits, int32_t* idx, float* dist2, int64_t nnz_capacity,
                              nf_stream_t stream)
{
    NF_CHECK_ARG(ws && row_splits && (nnz_capacity == 0 || (idx && dist2)), "null pointer");
    if (nq == 0 || nnz_capacity == 0) return NF_OK;
    hipStream_t st = (hipStream_t)stream;
    const float r2 = radius * radius;
    hipLaunchKernelGGL(k_radius<true>, dim3((nq + RS_QPB - 1) / RS_QPB), dim3(64 * RS_QPB), 0, st, ws, queries, nq, r2, ignore_same_pos,
                       (int*)nullptr, row_splits, idx, dist2, nnz_capacity);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// ------------------------------------------------------------------------------------------------
// exact 1-nearest neighbour (FluidErrors' gt -> prediction distance, utils/point_eval.py:36-58: the
// reference runs scipy's cKDTree on the host per frame).  Clouds are a few thousand points, so the
// exact answer is a brute-force sweep with the point set staged through LDS in 256-point tiles:
// 25 M distance tests per 5 k x 5 k frame, no tree, no host round trip.  The winner is chosen on the
// fp32 squared distance (ties -> lowest index) and its distance re-evaluated in double, like cKDTree's
// double arithmetic on the fp32 coordinates.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_nearest(const float* __restrict__ pts, int np, const float* __restrict__ q, int nq,
                                                 double* __restrict__ dist, int32_t* __restrict__ idx)
{
    __shared__ float sx[256], sy[256], sz[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < nq;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (live) { qx = q[3 * i]; qy = q[3 * i + 1]; qz = q[3 * i + 2]; }
    float best = INFINITY;
    int bi = -1;
    for (int base = 0; base < np; base += 256) {
        const int j = base + threadIdx.x;
        if (j < np) { sx[threadIdx.x] = pts[3 * j]; sy[threadIdx.x] = pts[3 * j + 1]; sz[threadIdx.x] = pts[3 * j + 2]; }
        __syncthreads();
        const int n = min(256, np - base);
        for (int t = 0; t < n; ++t) {
            const float d2 = nf_dist2(qx, qy, qz, sx[t], sy[t], sz[t]);
            if (d2 < best) { best = d2; bi = base + t; }
        }
        __syncthreads();
    }
    if (live) {
        double d = 0.0;
        if (bi >= 0) {
            const double dx = (double)qx - (double)pts[3 * bi], dy = (double)qy - (double)pts[3 * bi + 1],
                         dz = (double)qz - (double)pts[3 * bi + 2];
            d = sqrt(dx * dx + dy * dy + dz * dz);
        }
        dist[i] = d;
        if (idx) idx[i] = bi;
    }
}

extern "C" int nf_nearest(const float* pts, int n_pts, const float* queries, int nq, double* dist, int32_t* idx,
                          nf_stream_t stream)
{
    NF_CHECK_ARG(n_pts >= 1 && pts, "nearest-neighbour search needs at least one point");
    NF_CHECK_ARG(nq >= 0 && (nq == 0 || (queries && dist)), "null pointer");
    if (nq == 0) return NF_OK;
    hipLaunchKernelGGL(k_nearest, dim3((nq + 255) / 256), dim3(256), 0, (hipStream_t)stream, pts, n_pts, queries, nq, dist, idx);
    NF_CHECK_LAUNCH();
    return NF_OK;
}
