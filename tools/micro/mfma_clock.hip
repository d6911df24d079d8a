// Micro-benchmark (dev tool, round 5): is the fp16 matrix peak of this chip operand-dependent?  A bare v_mfma_f32_32x32x16_f16
// loop (one wave per SIMD, 8 independent accumulators, operands in registers) is timed with four operand fills — all zero, a
// constant, uniform random in [-1, 1), and "post-ReLU" (random, half of the values zero) — and every run reports, next to the
// wall-clock TFLOP/s, what the waves themselves saw: shader cycles per MFMA (s_memtime delta / MFMAs issued) and the effective
// shader clock (s_memtime delta / s_memrealtime delta x 100 MHz).  If cycles / MFMA stays at the pipe's 32 while the clock drops
// with the operands' switching activity, the gap to the nominal 2.5 PFLOP/s is power management (DVFS), not a schedule.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_clock(const u32x4* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ ticks,
                                               int iters, int f32)
{
    const int lane = threadIdx.x & 63;
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    u32x4 a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = src[i * 64 + lane];
    const u32x4 bq = src[512 + lane];
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    if (f32) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a[i][0]), __builtin_bit_cast(float, bq[0]), acc[i], 0, 0, 0);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a[i]), __builtin_bit_cast(h8, bq), acc[i], 0, 0, 0);
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        ticks[2 * w] = c1 - c0;
        ticks[2 * w + 1] = r1 - r0;
    }
}

static unsigned short f2h(float f)
{
    _Float16 h = (_Float16)f;
    unsigned short u;
    memcpy(&u, &h, 2);
    return u;
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 400000;
    u32x4* src; float* out; unsigned long long* ticks;
    hipMalloc(&src, 1024 * 16); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&ticks, 1024 * 16);
    const char* names[4] = {"all zero", "constant 0.5", "random [-1,1)", "post-ReLU (random, half zero)"};
    for (int f32 = 0; f32 < 2; ++f32)
        for (int mode = 0; mode < 4; ++mode) {
            std::vector<unsigned> h(1024 * 4);
            srand(7);
            for (auto& x : h) {
                float v[2];
                for (int k = 0; k < 2; ++k) {
                    float r = (float)rand() / RAND_MAX * 2.f - 1.f;
                    v[k] = mode == 0 ? 0.f : mode == 1 ? 0.5f : mode == 2 ? r : (r > 0.f ? r : 0.f);
                }
                if (f32) memcpy(&x, &v[0], 4);
                else x = f2h(v[0]) | ((unsigned)f2h(v[1]) << 16);
            }
            hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
            const int it = f32 ? iters / 4 : iters;
            for (int rep = 0; rep < 3; ++rep) {
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0);
                hipLaunchKernelGGL(k_clock, dim3(256), dim3(256), 0, 0, src, out, ticks, it, f32);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                std::vector<unsigned long long> t(2048);
                hipMemcpy(t.data(), ticks, 2048 * 8, hipMemcpyDeviceToHost);
                double cyc = 0, real = 0;
                for (int w = 0; w < 1024; ++w) { cyc += (double)t[2 * w]; real += (double)t[2 * w + 1]; }
                cyc /= 1024; real /= 1024;
                const double flop_per = f32 ? 4096.0 : 32768.0, nm = (double)it * 8;
                if (rep == 2)
                    printf("%-5s %-30s: %8.3f ms  %7.1f TFLOP/s   %.2f shader cycles / MFMA   effective clock %.3f GHz (nominal 2.4)\n", f32 ? "f32" : "f16",
                           names[mode], ms, 256.0 * 4 * nm * flop_per / ms / 1e9, cyc / nm, cyc / real * 0.1);
            }
        }
    return 0;
}
