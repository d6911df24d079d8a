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
