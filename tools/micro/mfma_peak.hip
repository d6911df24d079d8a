// Micro-benchmark (dev tool): sustained v_mfma_f32_32x32x16_f16 rate of one wave per SIMD with 8 independent
// accumulators, bare and with the per-MFMA filler mix of k_mlp_fwd_h (ds_read_b128 + VALU).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

template <int MODE>
__global__ void __launch_bounds__(256) k_peak(const u32x4* __restrict__ src, float* __restrict__ out, int iters)
{
    extern __shared__ u32x4 lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = src[i];
    __syncthreads();
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    u32x4 a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = lds[i * 64 + lane];
    u32x4 bq = src[lane + 64];
    float v0 = (float)lane, v1 = 1.f, v2 = 2.f;
    for (int it = 0; it < iters; ++it) {
        const u32x4* base = lds + ((it & 7) * 512) + lane;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            h8 av = __builtin_bit_cast(h8, a[i]), bv = __builtin_bit_cast(h8, bq);
            acc[i] = MFMA16(av, bv, acc[i]);
            if (MODE & 1) a[i] = base[i * 64];
            if (MODE & 2) { v0 = v0 * 1.0001f + v1; v1 = v1 * 0.999f + v2; v2 = v2 + v0; }
        }
        if (MODE & 2) bq[0] ^= (__float_as_uint(v2) & 1u);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (MODE & 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (MODE & 2) __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = v0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static void run(const u32x4* src, float* out, int iters, const char* name)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k_peak<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_peak<MODE>, dim3(256), dim3(256), 65536, 0, src, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double flops = 256.0 * 4 * iters * 8 * 32768.0;
        printf("%s: %.3f ms  %.1f TFLOP/s  (%.1f cycles/MFMA at 2.4 GHz)\n", name, ms, flops / ms / 1e9,
               ms * 1e-3 * 2.4e9 / (iters * 8.0));
    }
}

int main()
{
    std::vector<unsigned> h(4096 * 4);
    srand(1);
    for (auto& x : h) {   // random fp16 pairs in [-1,1)
        unsigned lo = 0x3000u + (rand() & 0x0bff) + ((rand() & 1) << 15), hi = 0x3000u + (rand() & 0x0bff) + ((rand() & 1) << 15);
        x = lo | (hi << 16);
    }
    u32x4* src; float* out;
    hipMalloc(&src, h.size() * 4); hipMalloc(&out, 256 * 256 * 4);
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int iters = 20000;
    run<0>(src, out, iters, "bare 8 MFMA            ");
    run<1>(src, out, iters, "+1 ds_read_b128 / MFMA ");
    run<2>(src, out, iters, "+3 VALU / MFMA         ");
    run<3>(src, out, iters, "+ds_read +3 VALU / MFMA");
    return 0;
}
