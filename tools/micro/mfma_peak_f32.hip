// Micro-benchmark (dev tool): sustained v_mfma_f32_32x32x2_f32 rate, one wave per SIMD, 8 independent accumulators,
// bare and with the filler mix of k_mlp_fwd (one global_load_dwordx4 per 4 MFMAs, one VALU per MFMA).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int MODE>
__global__ void __launch_bounds__(256) k_peak(const f32x4* __restrict__ src, float* __restrict__ out, int iters)
{
    const int lane = threadIdx.x & 63;
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x4 w0 = src[lane], w1 = src[lane + 64];
    float b = src[lane + 128][0], v0 = (float)lane;
    for (int it = 0; it < iters; ++it) {
        f32x4 n0 = w0, n1 = w1;
        if (MODE & 1) { n0 = src[((it & 31) * 128) + lane]; n1 = src[((it & 31) * 128) + 64 + lane]; }
        acc[0] = MFMA32(w0[0], b, acc[0]); acc[1] = MFMA32(w0[1], b, acc[1]);
        acc[2] = MFMA32(w0[2], b, acc[2]); acc[3] = MFMA32(w0[3], b, acc[3]);
        acc[4] = MFMA32(w1[0], b, acc[4]); acc[5] = MFMA32(w1[1], b, acc[5]);
        acc[6] = MFMA32(w1[2], b, acc[6]); acc[7] = MFMA32(w1[3], b, acc[7]);
        if (MODE & 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v0 = fmaxf(v0 * 1.0001f, 0.5f);
            b = b + (v0 > 1e30f ? 1.f : 0.f);
        }
        if (MODE & 1) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        w0 = n0; w1 = n1;
    }
    float s = v0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static void run(const f32x4* src, float* out, int iters, const char* name)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_peak<MODE>, dim3(256), dim3(256), 0, 0, src, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double flops = 256.0 * 4 * iters * 8 * 4096.0;
        printf("%s: %.3f ms  %.1f TFLOP/s  (%.1f cycles/MFMA at 2.4 GHz)\n", name, ms, flops / ms / 1e9,
               ms * 1e-3 * 2.4e9 / (iters * 8.0));
    }
}

int main()
{
    std::vector<float> h(32 * 128 * 4 + 1024);
    srand(1);
    for (auto& x : h) x = (float)rand() / RAND_MAX * 2.f - 1.f;
    f32x4* src; float* out;
    hipMalloc(&src, h.size() * 4); hipMalloc(&out, 256 * 256 * 4);
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int iters = 40000;
    run<0>(src, out, iters, "bare 8 MFMA f32              ");
    run<1>(src, out, iters, "+2 global_load_dwordx4 / 8   ");
    run<2>(src, out, iters, "+1 VALU / MFMA               ");
    run<3>(src, out, iters, "+loads +VALU                 ");
    return 0;
}
