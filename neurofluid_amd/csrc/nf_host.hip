// nf_host.hip — host-only helper of the training callers (no device code).
//
// nf_host_choice_mt19937: the pixel draw of the warm-up / end-to-end trainers,
//     np.random.choice(n, size=[ray_chunk], replace=False)          (/root/reference/trainer/trainer_renderer.py:119,
//                                                                     trainer/basetrainer.py:186-190)
// restated for numpy's legacy RandomState so that it can run OUTSIDE the interpreter lock: the draw is a full Fisher-Yates
// shuffle of arange(n) (choice without replacement = permutation(n)[:size]; n = 160 000 per view, 4 views per step = 5.6 ms of
// numpy time per 5.7 ms GPU step), and numpy holds the GIL for all of it — a read-ahead THREAD then starves the thread that
// feeds the GPU (measured: 7.5-8.0 ms per step with the numpy draw on a thread, 5.7 ms with a free draw).
// Same generator (MT19937, the published reference algorithm), same bounded-integer rule (masked rejection on 32-bit
// outputs), same swap order as RandomState.shuffle on a 1-D array, so the indices AND the generator state afterwards are
// those numpy would produce (tests/test_host_logic.py checks both against numpy).
#include "nf_common.h"
#include <time.h>
#include <stdlib.h>

namespace {

struct Mt {
    uint32_t* key;   // [624]
    int pos;
};

inline void mt_refill(Mt& s)
{
    const uint32_t A = 0x9908b0dfu, UP = 0x80000000u, LO = 0x7fffffffu;
    uint32_t* k = s.key;
    uint32_t y;
    int i = 0;
    for (; i < 624 - 397; ++i) { y = (k[i] & UP) | (k[i + 1] & LO); k[i] = k[i + 397] ^ (y >> 1) ^ ((0u - (y & 1u)) & A); }
    for (; i < 623; ++i) { y = (k[i] & UP) | (k[i + 1] & LO); k[i] = k[i + (397 - 624)] ^ (y >> 1) ^ ((0u - (y & 1u)) & A); }
    y = (k[623] & UP) | (k[0] & LO);
    k[623] = k[396] ^ (y >> 1) ^ ((0u - (y & 1u)) & A);
    s.pos = 0;
}

}  // namespace

extern "C" int nf_host_choice_mt19937(uint32_t* key, int* pos, int64_t n, int64_t size, int64_t* out)
{
    NF_CHECK_ARG(key && pos && out, "null pointer");
    NF_CHECK_ARG(n >= 1 && n <= 0x7fffffffLL && size >= 0 && size <= n, "need 1 <= n < 2^31 and 0 <= size <= n");
    NF_CHECK_ARG(*pos >= 0 && *pos <= 624, "bad generator position");
    int32_t* x = (int32_t*)malloc((size_t)n * sizeof(int32_t));
    if (!x) { nf_set_error("nf_host_choice_mt19937: out of host memory"); return NF_EINVAL; }
    for (int64_t i = 0; i < n; ++i) x[i] = (int32_t)i;
    Mt s = {key, *pos};
    // numpy: for i = n-1 .. 1: j = random_interval(i) (32-bit outputs masked to the smallest 2^k - 1 >= i, rejected while
    // > i); swap(x[i], x[j]).  Same draws in the same order, but branch-free: the mask is constant over i in [2^k, 2^(k+1))
    // and a rejected output swaps x[i] with itself (numpy's loop mispredicts on every fourth output: 1.7 -> 0.8 ms per draw)
    int64_t i = n - 1;
    while (i >= 1) {
        const uint32_t mask = 0xffffffffu >> __builtin_clz((uint32_t)i);
        const int64_t lo = (int64_t)(mask >> 1) + 1;
        while (i >= lo) {
            if (s.pos >= 624) mt_refill(s);
            int p = s.pos;
            while (p < 624 && i >= lo) {
                uint32_t y = key[p++];
                y ^= y >> 11;
                y ^= (y << 7) & 0x9d2c5680u;
                y ^= (y << 15) & 0xefc60000u;
                y ^= y >> 18;
                const uint32_t v = y & mask;
                const bool ok = v <= (uint32_t)i;
                const int64_t j = ok ? (int64_t)v : i;
                const int32_t t = x[j]; x[j] = x[i]; x[i] = t;
                i -= ok;
            }
            s.pos = p;
        }
    }
    for (int64_t i = 0; i < size; ++i) out[i] = x[i];
    *pos = s.pos;
    free(x);
    return NF_OK;
}


// Device-side address of a page-locked host allocation (hipHostMalloc / torch pinned memory), or NULL when the block is not
// mapped into the device's address space: nf_trans_step's overflow words live there (written by a kernel only when a
// neighbour row overflows, read by the host after the event behind that kernel).
extern "C" void* nf_pinned_device_ptr(void* host_ptr)
{
    void* d = nullptr;
    if (!host_ptr || hipHostGetDevicePointer(&d, host_ptr, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return d;
}

// Diagnostics: the calling thread's pending HIP error code (hipPeekAtLastError; 0 = none), not cleared.
extern "C" int nf_hip_peek_error(void) { return (int)hipPeekAtLastError(); }

// Host code: wait until a pinned host word (written by a kernel through its device address, nf_pinned_device_ptr) holds `expected`.
// The caller of nf_trans_step waits for the front kernel's completion word while the convolutions behind it are still running; round 5
// moved the spin out of the Python interpreter (a Python loop polled the word with the interpreter lock held on every rollout frame):
// a foreign call through ctypes releases the lock, and the loop here is plain volatile loads with a pause, the clock read every 1 024 polls.
// Returns 0 when the word arrived, 1 on timeout (the caller then synchronises the device to surface what went wrong).
extern "C" int nf_host_wait_word(const volatile int32_t* word, int32_t expected, double timeout_s)
{
    if (!word) return 1;
    unsigned spins = 0;
    bool armed = false;
    struct timespec t0 = {0, 0};
    while (*word != expected) {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
        if ((++spins & 0x3ff) == 0) {
            struct timespec now;
            clock_gettime(CLOCK_MONOTONIC, &now);
            if (!armed) { t0 = now; armed = true; }
            else if ((double)(now.tv_sec - t0.tv_sec) + 1e-9 * (double)(now.tv_nsec - t0.tv_nsec) > timeout_s) return *word == expected ? 0 : 1;
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Adam step of a whole parameter list in ONE launch (round 4).  torch's fused Adam walks the 48 + 22 tensors of the two models
// through multi_tensor_apply: 2-3 launches of ~44 us for 2.0 M parameters (57 MB of traffic = ~10 us at HBM speed) on every
// training step.  Here the tensor table (pointers, sizes, the per-tensor step size and bias correction) travels in the kernel
// arguments and a block finds its (tensor, chunk) by a scan over the table's chunk prefix.  The arithmetic is torch.optim.Adam's
// default (single-tensor) formulation, operation by operation:
//     g' = g + wd p;  m = m + (g' - m)(1 - b1)  [lerp];  v = v b2 + (1 - b2) g' g';  p = p - step_size * m / (sqrt(v) / bc2_sqrt + eps)
// ------------------------------------------------------------------------------------------------
#define NA_MAX 64
#define NA_CHUNK 4096
struct NfAdamArgs {
    float* p[NA_MAX]; const float* g[NA_MAX]; float* m[NA_MAX]; float* v[NA_MAX];
    int n[NA_MAX]; int chunk0[NA_MAX + 1];
    float step_size[NA_MAX], bc2_sqrt[NA_MAX];
    int count;
    float beta2, w1, w2, eps, weight_decay;      // w1 = 1 - beta1, w2 = 1 - beta2, rounded from the DOUBLE differences (as torch's scalars are)
    const float* sched;     // nf_adam_step_dev: {step_size, bc2_sqrt} of THIS step in device memory (a replayed graph's arguments are frozen)
    const int* skip;        // nf_adam_step_dev: a non-zero word makes the step a no-op (a training step whose forward was truncated)
};

__global__ void __launch_bounds__(256) k_adam(NfAdamArgs A)
{
    int t = 0;
    const int b = blockIdx.x;
    while (t + 1 < A.count && b >= A.chunk0[t + 1]) ++t;
    const int base = (b - A.chunk0[t]) * NA_CHUNK, n = A.n[t];
    float* __restrict__ p = A.p[t];
    const float* __restrict__ g = A.g[t];
    float* __restrict__ m = A.m[t];
    float* __restrict__ v = A.v[t];
    if (A.skip && *A.skip) return;
    const float ss = A.sched ? A.sched[0] : A.step_size[t], bc = A.sched ? A.sched[1] : A.bc2_sqrt[t], w1 = A.w1, w2 = A.w2;
#pragma unroll 4
    for (int k = 0; k < NA_CHUNK / 256; ++k) {
        const int i = base + k * 256 + threadIdx.x;
        if (i >= n) break;
        float gi = g[i];
        const float pi = p[i];
        if (A.weight_decay != 0.f) gi = gi + A.weight_decay * pi;
        const float mi = m[i], vi = v[i];
        const float mn = fmaf(w1, gi - mi, mi);                    // torch.lerp(m, g, 1 - b1) for a weight < 0.5: fma(w, g - m, m)
        const float vn = vi * A.beta2 + w2 * (gi * gi);            // mul_(b2).addcmul_(g, g, value = 1 - b2)
        const float denom = sqrtf(vn) / bc + A.eps;
        m[i] = mn; v[i] = vn;
        p[i] = pi + (-ss) * (mn / denom);                          // addcdiv_(m, denom, value = -step_size)
    }
}

static int adam_launch(int count, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                       const int64_t* sizes, const float* step_size, const float* bc2_sqrt, const float* sched, const int* skip,
                       double beta1, double beta2, float eps, float weight_decay, nf_stream_t stream)
{
    for (int t0 = 0; t0 < count; t0 += NA_MAX) {
        NfAdamArgs A;
        const int c = count - t0 < NA_MAX ? count - t0 : NA_MAX;
        int chunks = 0;
        for (int t = 0; t < c; ++t) {
            NF_CHECK_ARG(sizes[t0 + t] >= 0 && sizes[t0 + t] < (1ll << 31), "tensor too large");
            A.p[t] = params[t0 + t]; A.g[t] = grads[t0 + t]; A.m[t] = exp_avg[t0 + t]; A.v[t] = exp_avg_sq[t0 + t];
            A.n[t] = (int)sizes[t0 + t]; A.chunk0[t] = chunks;
            A.step_size[t] = step_size ? step_size[t0 + t] : 0.f; A.bc2_sqrt[t] = bc2_sqrt ? bc2_sqrt[t0 + t] : 1.f;
            chunks += (A.n[t] + NA_CHUNK - 1) / NA_CHUNK;
        }
        A.chunk0[c] = chunks;
        A.count = c; A.beta2 = (float)beta2; A.w1 = (float)(1.0 - beta1); A.w2 = (float)(1.0 - beta2); A.eps = eps; A.weight_decay = weight_decay;
        A.sched = sched; A.skip = skip;
        if (chunks > 0) hipLaunchKernelGGL(k_adam, dim3(chunks), dim3(256), 0, (hipStream_t)stream, A);
    }
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_adam_step(int count, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                            const int64_t* sizes, const float* step_size, const float* bc2_sqrt, double beta1, double beta2, float eps,
                            float weight_decay, nf_stream_t stream)
{
    NF_CHECK_ARG(count >= 0 && (count == 0 || (params && grads && exp_avg && exp_avg_sq && sizes && step_size && bc2_sqrt)), "null pointer");
    return adam_launch(count, params, grads, exp_avg, exp_avg_sq, sizes, step_size, bc2_sqrt, nullptr, nullptr, beta1, beta2, eps, weight_decay, stream);
}

// The same step for a REPLAYED training step (HIP graph): the two per-step scalars come from device memory (sched[0] = step_size,
// sched[1] = bc2_sqrt, shared by all tensors: they have taken the same number of steps), and a non-zero *skip (or NULL) turns the
// launch into a no-op — the step's forward overflowed its row capacities, the host redoes it (nf_note_overflow).
extern "C" int nf_adam_step_dev(int count, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                                const int64_t* sizes, const float* sched, const int32_t* skip, double beta1, double beta2, float eps,
                                float weight_decay, nf_stream_t stream)
{
    NF_CHECK_ARG(count >= 0 && (count == 0 || (params && grads && exp_avg && exp_avg_sq && sizes && sched)), "null pointer");
    return adam_launch(count, params, grads, exp_avg, exp_avg_sq, sizes, nullptr, nullptr, sched, (const int*)skip, beta1, beta2, eps, weight_decay, stream);
}

// One thread: state = {poisoned, first poisoned step, step counter, count[0], count[1]}.  The counter advances by one per call;
// if a count (device words, e.g. the active rows of a render pass) exceeds its capacity the poison word is set and STAYS set
// (every optimiser step that follows is skipped through nf_adam_step_dev's `skip` until the host clears the word), and the first
// such step's counter value is kept.  host_ring (optional): 8 records of 8 words in host memory mapped into the device address
// space; the record of step s goes to slot s & 7, its word 2 (= s + 1) is written last behind a system-scope fence, so a host that
// finds word 2 == s + 1 reads a complete record of step s.
__global__ void k_note_overflow(const int* __restrict__ c0, int cap0, const int* __restrict__ c1, int cap1, int* __restrict__ state,
                                volatile int* host_ring)
{
    const int n0 = c0 ? *c0 : 0, n1 = c1 ? *c1 : 0;
    const int step = state[2];
    if ((n0 > cap0 || n1 > cap1) && !state[0]) { state[0] = 1; state[1] = step; }
    state[2] = step + 1;
    state[3] = n0; state[4] = n1;
    if (host_ring) {
        volatile int* h = host_ring + (step & 7) * 8;
        h[0] = state[0]; h[1] = state[1]; h[3] = n0; h[4] = n1;
        __threadfence_system();
        h[2] = step + 1;
    }
}

// The same bookkeeping over up to FOUR (count, capacity) pairs — the end-to-end step replayed as a graph watches the two render passes'
// active rows AND the transition step's two neighbour-pair totals.  state = {poisoned, first poisoned step, step counter, count[0..3]};
// the host record carries the same words (word 2 last).
struct NfOverflow4 { const int* c[4]; int cap[4]; };
__global__ void k_note_overflow4(NfOverflow4 A, int* __restrict__ state, volatile int* host_ring)
{
    int n[4];
    bool over = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) { n[k] = A.c[k] ? *A.c[k] : 0; over |= n[k] > A.cap[k]; }
    const int step = state[2];
    if (over && !state[0]) { state[0] = 1; state[1] = step; }
    state[2] = step + 1;
#pragma unroll
    for (int k = 0; k < 4; ++k) state[3 + k] = n[k];
    if (host_ring) {
        volatile int* h = host_ring + (step & 7) * 8;
        h[0] = state[0]; h[1] = state[1];
#pragma unroll
        for (int k = 0; k < 4; ++k) h[3 + k] = n[k];
        __threadfence_system();
        h[2] = step + 1;
    }
}

extern "C" int nf_note_overflow4(const int32_t* const* counts, const int32_t* caps, int32_t* state, int32_t* host_ring, nf_stream_t stream)
{
    NF_CHECK_ARG(counts && caps && state, "null pointer");
    NfOverflow4 A;
    for (int k = 0; k < 4; ++k) { A.c[k] = (const int*)counts[k]; A.cap[k] = caps[k]; }
    hipLaunchKernelGGL(k_note_overflow4, dim3(1), dim3(1), 0, (hipStream_t)stream, A, (int*)state, (volatile int*)host_ring);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_note_overflow(const int32_t* count0, int cap0, const int32_t* count1, int cap1, int32_t* state, int32_t* host_ring,
                                nf_stream_t stream)
{
    NF_CHECK_ARG(state, "null pointer");
    hipLaunchKernelGGL(k_note_overflow, dim3(1), dim3(1), 0, (hipStream_t)stream, (const int*)count0, cap0, (const int*)count1, cap1, (int*)state,
                       (volatile int*)host_ring);
    NF_CHECK_LAUNCH();
    return NF_OK;
}


// ------------------------------------------------------------------------------------------------
// The loss of the end-to-end training step (trainer/trainer_e2e.py:264-280) and its gradients in ONE launch:
//     loss = ( sum (rgb0 - rgb)^2 [+ sum (rgb1 - rgb)^2] ) / denom  +  w_boundary * mean | pos - clamp(pos, lo, hi) |
// (the views of a step are equally sized, so the sum of their MSE means is one sum over one denominator; the boundary term is
// basetrainer.py:108-116's per-axis clamp under an L1 mean).  As torch ops this chain and its autograd backward were ~35 launches of
// 3-9 us in a 3.2 ms step.  One workgroup: 12 k colour values and 15 k coordinates are nothing to reduce.
// Outputs: the loss and the gradients for a unit upstream gradient (the caller scales them).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_e2e_loss(const float* __restrict__ rgb0, const float* __restrict__ rgb1, const float* __restrict__ rgb,
                                                   int n_rgb, float inv_denom, const float* __restrict__ pos, int n_pos3, float lo0, float lo1,
                                                   float lo2, float hi0, float hi1, float hi2, float wb, float* __restrict__ loss,
                                                   float* __restrict__ g_rgb0, float* __restrict__ g_rgb1, float* __restrict__ g_pos)
{
    __shared__ float s_a[16], s_b[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float acc = 0.f, accb = 0.f;
    for (int i = tid; i < n_rgb; i += 1024) {
        const float t = rgb[i];
        const float d0 = rgb0[i] - t;
        acc += d0 * d0;
        g_rgb0[i] = 2.f * d0 * inv_denom;
        if (rgb1) { const float d1 = rgb1[i] - t; acc += d1 * d1; g_rgb1[i] = 2.f * d1 * inv_denom; }
    }
    const float inv_n = n_pos3 > 0 ? 1.f / (float)n_pos3 : 0.f;
    for (int i = tid; i < n_pos3; i += 1024) {
        const int d = i % 3;
        const float l = d == 0 ? lo0 : (d == 1 ? lo1 : lo2), h = d == 0 ? hi0 : (d == 1 ? hi1 : hi2);
        const float p = pos[i], dp = p - fminf(fmaxf(p, l), h);
        accb += fabsf(dp);
        g_pos[i] = (dp > 0.f ? 1.f : (dp < 0.f ? -1.f : 0.f)) * (wb * inv_n);
    }
    acc = nf_wave_sum(acc); accb = nf_wave_sum(accb);
    if (lane == 0) { s_a[wv] = acc; s_b[wv] = accb; }
    __syncthreads();
    if (tid == 0) {
        float a = 0.f, b = 0.f;
        for (int w = 0; w < 16; ++w) { a += s_a[w]; b += s_b[w]; }
        *loss = a * inv_denom + wb * (b * inv_n);
    }
}

extern "C" int nf_e2e_loss(const float* rgb0, const float* rgb1, const float* rgb, int n_rgb, int denom, const float* pos, int n_points,
                           const float lo[3], const float hi[3], float w_boundary, float* loss, float* g_rgb0, float* g_rgb1, float* g_pos,
                           nf_stream_t stream)
{
    NF_CHECK_ARG(rgb0 && rgb && loss && g_rgb0 && (!rgb1 || g_rgb1) && n_rgb >= 0 && denom > 0, "bad colour arguments");
    NF_CHECK_ARG(n_points >= 0 && (n_points == 0 || (pos && lo && hi && g_pos)), "bad position arguments");
    hipLaunchKernelGGL(k_e2e_loss, dim3(1), dim3(1024), 0, (hipStream_t)stream, rgb0, rgb1, rgb, n_rgb, 1.f / (float)denom, pos, 3 * n_points,
                       n_points ? lo[0] : 0.f, n_points ? lo[1] : 0.f, n_points ? lo[2] : 0.f, n_points ? hi[0] : 0.f, n_points ? hi[1] : 0.f,
                       n_points ? hi[2] : 0.f, w_boundary, loss, g_rgb0, g_rgb1, g_pos);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// Up to three contiguous float buffers scaled by one scalar read from device memory (the upstream gradient autograd hands to the fused loss's
// backward), out of place (out == in is allowed): one launch instead of one aten::mul per buffer.
__global__ void __launch_bounds__(256) k_scale3(const float* __restrict__ a, long long na, const float* __restrict__ b, long long nb, const float* __restrict__ c,
                                                long long nc, const float* __restrict__ scale, float* oa, float* ob, float* oc)
{
    const float s = *scale;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < na) { oa[i] = a[i] * s; return; }
    i -= na;
    if (i < nb) { ob[i] = b[i] * s; return; }
    i -= nb;
    if (i < nc) oc[i] = c[i] * s;
}

extern "C" int nf_scale3(const float* a, int64_t na, const float* b, int64_t nb, const float* c, int64_t nc, const float* scale, float* out_a, float* out_b,
                         float* out_c, nf_stream_t stream)
{
    NF_CHECK_ARG(scale && na >= 0 && nb >= 0 && nc >= 0 && ((a && out_a) || !na) && ((b && out_b) || !nb) && ((c && out_c) || !nc), "bad arguments");
    const long long n = (long long)na + nb + nc;
    if (n == 0) return NF_OK;
    hipLaunchKernelGGL(k_scale3, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, (long long)na, b, (long long)nb, c, (long long)nc, scale,
                       out_a, out_b, out_c);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// ------------------------------------------------------------------------------------------------
// Round 5: the glue between the HIP launches of the transition model's backward (autograd_bwd._trans_backward) as three small kernels
// instead of ~25 ATen launches per step (threshold_backward, add, contiguous, sum(0), clone, reshape / permute copies).
// ------------------------------------------------------------------------------------------------
// out = (prev > 0 ? dx : 0) [+ res]: the ReLU in front of a layer (models/transmodel.py:121), back-propagated, plus the residual branch's
// gradient (:127-128).  Row-major contiguous, n elements.
__global__ void __launch_bounds__(256) k_relu_bwd_add(const float* __restrict__ dx, const float* __restrict__ prev, const float* __restrict__ res,
                                                      float* __restrict__ out, long long n)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = prev[i] > 0.f ? dx[i] : 0.f;
    if (res) v += res[i];
    out[i] = v;
}

extern "C" int nf_relu_bwd_add(const float* dx, const float* prev, const float* res, float* out, int64_t n, nf_stream_t stream)
{
    NF_CHECK_ARG(dx && prev && out && n >= 0, "bad arguments");
    if (n == 0) return NF_OK;
    hipLaunchKernelGGL(k_relu_bwd_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dx, prev, res, out, (long long)n);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// ------------------------------------------------------------------------------------------------
// The pixel gathers of one training step (trainer/basetrainer.py:186-193: rays[v][ys, xs], rgbs[v][ys, xs] per view, and the view's camera
// position per ray) in ONE launch: thread i = view v * per_view + k copies row flat[i] of view v's (H*W, 6) rays and (H*W, C) colours and
// writes the view's origin c2w[v][:, 3].  Replaces, per step, 2 index_select per view + 3 cat + a repeat_interleave.
// ------------------------------------------------------------------------------------------------
#define NF_GATHER_MAX_VIEWS 16
struct NfGatherViews { const float* rays[NF_GATHER_MAX_VIEWS]; const float* rgb[NF_GATHER_MAX_VIEWS]; const float* c2w[NF_GATHER_MAX_VIEWS]; };

__global__ void __launch_bounds__(256) k_gather_view_pixels(NfGatherViews V, int n_views, int per_view, int rgb_c, long long n_pixels,
                                                            const long long* __restrict__ flat, float* __restrict__ rays_out,
                                                            float* __restrict__ rgb_out, float* __restrict__ ro_out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_views * per_view) return;
    const int v = i / per_view;
    long long px = flat[i];
    if (px < 0 || px >= n_pixels) px = 0;          // (never read out of bounds; the host wrapper validates the indices before it uploads them)
    const float* r = V.rays[v] + px * 6;
#pragma unroll
    for (int c = 0; c < 6; ++c) rays_out[(size_t)i * 6 + c] = r[c];
    const float* g = V.rgb[v] + px * rgb_c;
    for (int c = 0; c < rgb_c; ++c) rgb_out[(size_t)i * rgb_c + c] = g[c];
    const float* cw = V.c2w[v];
    ro_out[(size_t)i * 3 + 0] = cw[3]; ro_out[(size_t)i * 3 + 1] = cw[7]; ro_out[(size_t)i * 3 + 2] = cw[11];
}

// rays / rgb / c2w: HOST arrays of n_views device pointers ((H*W, 6), (H*W, rgb_c), (3, 4) row-major contiguous); flat: n_views * per_view int64
// pixel indices on the device, each in [0, n_pixels) (the caller validates them on the host, where they are drawn; the kernel clamps strays to 0).
extern "C" int nf_gather_view_pixels(int n_views, const float* const* rays, const float* const* rgb, const float* const* c2w, int per_view,
                                     int rgb_c, int64_t n_pixels, const int64_t* flat, float* rays_out, float* rgb_out, float* ro_out,
                                     nf_stream_t stream)
{
    NF_CHECK_ARG(rays && rgb && c2w && flat && rays_out && rgb_out && ro_out, "null pointer");
    NF_CHECK_ARG(n_views >= 1 && n_views <= NF_GATHER_MAX_VIEWS && per_view >= 0 && rgb_c >= 1 && n_pixels >= 1, "bad sizes (at most 16 views per call)");
    if (per_view == 0) return NF_OK;
    NfGatherViews V;
    for (int v = 0; v < n_views; ++v) {
        NF_CHECK_ARG(rays[v] && rgb[v] && c2w[v], "null view pointer");
        V.rays[v] = rays[v]; V.rgb[v] = rgb[v]; V.c2w[v] = c2w[v];
    }
    for (int v = n_views; v < NF_GATHER_MAX_VIEWS; ++v) { V.rays[v] = nullptr; V.rgb[v] = nullptr; V.c2w[v] = nullptr; }
    const int n = n_views * per_view;
    hipLaunchKernelGGL(k_gather_view_pixels, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, V, n_views, per_view, rgb_c,
                       (long long)n_pixels, (const long long*)flat, rays_out, rgb_out, ro_out);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// The same gather with the views' pointers read from DEVICE memory: table[3 * v + {0, 1, 2}] = rays, rgb, c2w of view v (64-bit device
// addresses).  For a training step that is replayed as a HIP graph while the frame it draws from changes from step to step
// (train_e2e.py walks the frames of a sequence): the launch's arguments are baked into the graph, the table is rewritten per step.
__global__ void __launch_bounds__(256) k_gather_view_pixels_tab(const unsigned long long* __restrict__ table, int n_views, int per_view, int rgb_c,
                                                                long long n_pixels, const long long* __restrict__ flat, float* __restrict__ rays_out,
                                                                float* __restrict__ rgb_out, float* __restrict__ ro_out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_views * per_view) return;
    const int v = i / per_view;
    long long px = flat[i];
    if (px < 0 || px >= n_pixels) px = 0;
    const float* r = (const float*)table[3 * v] + px * 6;
#pragma unroll
    for (int c = 0; c < 6; ++c) rays_out[(size_t)i * 6 + c] = r[c];
    const float* g = (const float*)table[3 * v + 1] + px * rgb_c;
    for (int c = 0; c < rgb_c; ++c) rgb_out[(size_t)i * rgb_c + c] = g[c];
    const float* cw = (const float*)table[3 * v + 2];
    ro_out[(size_t)i * 3 + 0] = cw[3]; ro_out[(size_t)i * 3 + 1] = cw[7]; ro_out[(size_t)i * 3 + 2] = cw[11];
}

extern "C" int nf_gather_view_pixels_tab(int n_views, const uint64_t* table, int per_view, int rgb_c, int64_t n_pixels, const int64_t* flat,
                                         float* rays_out, float* rgb_out, float* ro_out, nf_stream_t stream)
{
    NF_CHECK_ARG(table && flat && rays_out && rgb_out && ro_out, "null pointer");
    NF_CHECK_ARG(n_views >= 1 && per_view >= 0 && rgb_c >= 1 && n_pixels >= 1, "bad sizes");
    if (per_view == 0) return NF_OK;
    const int n = n_views * per_view;
    hipLaunchKernelGGL(k_gather_view_pixels_tab, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const unsigned long long*)table, n_views,
                       per_view, rgb_c, (long long)n_pixels, (const long long*)flat, rays_out, rgb_out, ro_out);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// Column sums of a (rows x cols) row-major matrix with leading dimension lda (a column slice of a wider matrix is fine): the bias
// gradients.  One workgroup per FOUR columns, 256 threads: thread r walks rows r, r + 256, ... with one 16-byte load per row (scalar loads
// when the slice is not 16-byte aligned), then the 256 partial sums of each column are added in a fixed order from LDS — deterministic.
// (Round 5: the first version gave a workgroup 16 columns and a thread every 16th row — 4 workgroups walking 307 rows each at 4 913
// particles, 65-80 us of load latency per call, four calls per end-to-end step.)  out2 (optional) receives a second copy (conv.bias and
// dense.bias of a layer get the same gradient).
__global__ void __launch_bounds__(256) k_colsum(const float* __restrict__ a, int rows, int cols, int lda, float* __restrict__ out, float* __restrict__ out2,
                                                int vec)
{
    __shared__ float part[4][257];
    const int c0 = blockIdx.x * 4, t = threadIdx.x;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (vec) {
        for (int r = t; r < rows; r += 256) {
            const float4 v = *(const float4*)(a + (size_t)r * lda + c0);
            s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
        }
    } else {
        const int nc = min(4, cols - c0);
        for (int r = t; r < rows; r += 256) {
            const float* p = a + (size_t)r * lda + c0;
            s0 += p[0];
            if (nc > 1) s1 += p[1];
            if (nc > 2) s2 += p[2];
            if (nc > 3) s3 += p[3];
        }
    }
    part[0][t] = s0; part[1][t] = s1; part[2][t] = s2; part[3][t] = s3;
    __syncthreads();
    // column c of the four: 64 threads add 4 partials each (fixed order), then lane 0 of the group adds the 64
    const int c = t >> 6, l = t & 63;
    float q = part[c][4 * l] + part[c][4 * l + 1] + part[c][4 * l + 2] + part[c][4 * l + 3];
    __syncthreads();
    part[c][l] = q;
    __syncthreads();
    if (l == 0 && c0 + c < cols) {
        float tot = 0.f;
#pragma unroll 8
        for (int k = 0; k < 64; ++k) tot += part[c][k];
        out[c0 + c] = tot;
        if (out2) out2[c0 + c] = tot;
    }
}

extern "C" int nf_colsum(const float* a, int rows, int cols, int lda, float* out, float* out2, nf_stream_t stream)
{
    NF_CHECK_ARG(a && out && rows >= 0 && cols > 0 && lda >= cols, "bad arguments");
    const int vec = (cols % 4 == 0) && (lda % 4 == 0) && (((uintptr_t)a & 15) == 0);
    hipLaunchKernelGGL(k_colsum, dim3((cols + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, rows, cols, lda, out, out2, vec);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// dB (Cin x 65*Cout: the gradient of [filter as (Cin x 64*Cout) | dense_w^T]) -> the filter's gradient in its own layout (64 cells, Cin, Cout)
// and the Linear weight's (Cout, Cin): what `dB[:, :64*Cout].reshape(Cin, 64, Cout).permute(1, 0, 2)` and `dB[:, 64*Cout:].t()` copied.
__global__ void __launch_bounds__(256) k_cconv_split_db(const float* __restrict__ dB, int cin, int cout, float* __restrict__ dK, float* __restrict__ dW)
{
    const int i = blockIdx.x * 256 + threadIdx.x, w = 65 * cout;
    if (i >= cin * w) return;
    const int ci = i / w, col = i % w;
    const float v = dB[i];
    if (col < 64 * cout) dK[((size_t)(col / cout) * cin + ci) * cout + col % cout] = v;
    else dW[(size_t)(col - 64 * cout) * cin + ci] = v;
}

extern "C" int nf_cconv_split_db(const float* dB, int cin, int cout, float* dK, float* dW, nf_stream_t stream)
{
    NF_CHECK_ARG(dB && dK && dW && cin > 0 && cout > 0, "bad arguments");
    const int n = cin * 65 * cout;
    hipLaunchKernelGGL(k_cconv_split_db, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, dB, cin, cout, dK, dW);
    NF_CHECK_LAUNCH();
    return NF_OK;
}
