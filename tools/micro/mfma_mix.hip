// Micro-benchmark (dev tool, round 5): what does the fp16 matrix pipe deliver under the INSTRUCTION MIX of the fp16 NeRF MLP kernel when
// the matrix pipe never waits?  A v_mfma_f32_32x32x16_f16 stream with the kernel's operand statistics (A = random "weights", B = post-ReLU
// "activations" of two tiles, half of them zero), in four steps: (0) operands in registers, (1) + one ds_read_b128 A operand per two MFMAs
// out of a 48 KB LDS image (the kernel's weight ring), (2) + the conversion work of a finished block (16 accumulator reads, 8 v_cvt_pk,
// 8 v_pk_max per 34 MFMAs), (3) + the ring refill (global -> LDS, a quarter of a 16 KB chunk per wave per 32 MFMAs).  Reports wall-clock
// PFLOP/s, shader cycles per MFMA and the effective clock, like mfma_clock.hip.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0)

#include "mfma_mix_asm.inc"
#define R8(n) "v" #n "0", "v" #n "1", "v" #n "2", "v" #n "3", "v" #n "4", "v" #n "5", "v" #n "6", "v" #n "7", "v" #n "8", "v" #n "9"
#define A8(n) "a" #n "0", "a" #n "1", "a" #n "2", "a" #n "3", "a" #n "4", "a" #n "5", "a" #n "6", "a" #n "7", "a" #n "8", "a" #n "9"
#define MIX_CLOB "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", R8(1), R8(2), R8(3), R8(4), R8(5), R8(6), "v70", "v71", "v72", "v73", \
                 "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", A8(1), A8(2), A8(3), A8(4), A8(5), "a60", "a61", "a62", "a63", \
                 "s40", "s41", "s42", "s43", "s44", "scc", "memory"

// the same four steps with the loop written out and scheduled by hand (A operands requested three steps ahead)
template <int MODE>
__global__ void __launch_bounds__(256) k_mix_asm(const u32x4* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ ticks, int iters)
{
    extern __shared__ u32x4 ring[];           // 48 KB ring + 16 KB refill target
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 48 * 64; i += 256) ring[i] = src[i];
    __syncthreads();
    const unsigned lds = (unsigned)(size_t)ring + lane * 16, loff = lane * 16, roff = wave * 4096 + lane * 16;
    const u32x4* bsrc = src + 48 * 64;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    if (MODE == 1) asm volatile(MIX_BODY_1 : : [iters] "s"(iters), [lds] "v"(lds), [loff] "v"(loff), [roff] "v"(roff), [src] "s"(src), [bsrc] "s"(bsrc), [bsrc2] "s"(bsrc + 256) : MIX_CLOB, "m0");
    if (MODE == 2) asm volatile(MIX_BODY_2 : : [iters] "s"(iters), [lds] "v"(lds), [loff] "v"(loff), [roff] "v"(roff), [src] "s"(src), [bsrc] "s"(bsrc), [bsrc2] "s"(bsrc + 256) : MIX_CLOB, "m0");
    if (MODE == 3) asm volatile(MIX_BODY_3 : : [iters] "s"(iters), [lds] "v"(lds), [loff] "v"(loff), [roff] "v"(roff), [src] "s"(src), [bsrc] "s"(bsrc), [bsrc2] "s"(bsrc + 256) : MIX_CLOB, "m0");
    if (MODE == 4) asm volatile(MIX_BODY_4 : : [iters] "s"(iters), [lds] "v"(lds), [loff] "v"(loff), [roff] "v"(roff), [src] "s"(src), [bsrc] "s"(bsrc), [bsrc2] "s"(bsrc + 256) : MIX_CLOB, "m0");
    if (MODE == 5) asm volatile(MIX_BODY_5 : : [iters] "s"(iters), [lds] "v"(lds), [loff] "v"(loff), [roff] "v"(roff), [src] "s"(src), [bsrc] "s"(bsrc), [bsrc2] "s"(bsrc + 256) : MIX_CLOB, "m0");
    if (MODE == 6) asm volatile(MIX_BODY_6 : : [iters] "s"(iters), [lds] "v"(lds), [loff] "v"(loff), [roff] "v"(roff), [src] "s"(src), [bsrc] "s"(bsrc), [bsrc2] "s"(bsrc + 256) : MIX_CLOB, "m0");
    if (MODE == 7) asm volatile(MIX_BODY_7 : : [iters] "s"(iters), [lds] "v"(lds), [loff] "v"(loff), [roff] "v"(roff), [src] "s"(src), [bsrc] "s"(bsrc), [bsrc2] "s"(bsrc + 256) : MIX_CLOB, "m0");
    if (MODE == 8) asm volatile(MIX_BODY_8 : : [iters] "s"(iters), [lds] "v"(lds), [loff] "v"(loff), [roff] "v"(roff), [src] "s"(src), [bsrc] "s"(bsrc), [bsrc2] "s"(bsrc + 256) : MIX_CLOB, "m0");
    if (MODE == 9) asm volatile(MIX_BODY_9 : : [iters] "s"(iters), [lds] "v"(lds), [loff] "v"(loff), [roff] "v"(roff), [src] "s"(src), [bsrc] "s"(bsrc), [bsrc2] "s"(bsrc + 256) : MIX_CLOB, "m0");
    if (MODE == 10) asm volatile(MIX_BODY_10 : : [iters] "s"(iters), [lds] "v"(lds), [loff] "v"(loff), [roff] "v"(roff), [src] "s"(src), [bsrc] "s"(bsrc), [bsrc2] "s"(bsrc + 256) : MIX_CLOB, "m0");
    if (MODE == 11) asm volatile(MIX_BODY_11 : : [iters] "s"(iters), [lds] "v"(lds), [loff] "v"(loff), [roff] "v"(roff), [src] "s"(src), [bsrc] "s"(bsrc), [bsrc2] "s"(bsrc + 256) : MIX_CLOB, "m0");
    if (MODE == 12) asm volatile(MIX_BODY_12 : : [iters] "s"(iters), [lds] "v"(lds), [loff] "v"(loff), [roff] "v"(roff), [src] "s"(src), [bsrc] "s"(bsrc), [bsrc2] "s"(bsrc + 256) : MIX_CLOB, "m0");
    if (MODE == 13) asm volatile(MIX_BODY_13 : : [iters] "s"(iters), [lds] "v"(lds), [loff] "v"(loff), [roff] "v"(roff), [src] "s"(src), [bsrc] "s"(bsrc), [bsrc2] "s"(bsrc + 256) : MIX_CLOB, "m0");
    if (MODE == 14) asm volatile(MIX_BODY_14 : : [iters] "s"(iters), [lds] "v"(lds), [loff] "v"(loff), [roff] "v"(roff), [src] "s"(src), [bsrc] "s"(bsrc), [bsrc2] "s"(bsrc + 256) : MIX_CLOB, "m0");
    if (MODE == 15) asm volatile(MIX_BODY_15 : : [iters] "s"(iters), [lds] "v"(lds), [loff] "v"(loff), [roff] "v"(roff), [src] "s"(src), [bsrc] "s"(bsrc), [bsrc2] "s"(bsrc + 256) : MIX_CLOB, "m0");
    if (MODE == 16) asm volatile(MIX_BODY_16 : : [iters] "s"(iters), [lds] "v"(lds), [loff] "v"(loff), [roff] "v"(roff), [src] "s"(src), [bsrc] "s"(bsrc), [bsrc2] "s"(bsrc + 256) : MIX_CLOB, "m0", "s45", "s46", "s47", "s48", "s49");
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = 0.f;
    if (lane == 0) {
        const int w = blockIdx.x * 4 + wave;
        ticks[2 * w] = c1 - c0;
        ticks[2 * w + 1] = r1 - r0;
    }
}

template <int MODE>
__global__ void __launch_bounds__(256) k_mix(const u32x4* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ ticks, int iters)
{
    __shared__ u32x4 ring[48 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 48 * 64; i += 256) ring[i] = src[i];
    __syncthreads();
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    u32x4 b0[4], b1[4];          // "activations": 4 K-steps per tile kept in registers, cycled
    for (int i = 0; i < 4; ++i) { b0[i] = src[48 * 64 + i * 64 + lane]; b1[i] = src[48 * 64 + (4 + i) * 64 + lane]; }
    u32x4 areg[4];
    for (int i = 0; i < 4; ++i) areg[i] = src[i * 64 + lane];
    unsigned keep = 0;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        // one "output block": 16 K-steps x 2 tiles = 32 MFMAs; the A operand of step s + 3 is requested at step s
        if (MODE >= 1) {
            u32x4 pa[16];
#pragma unroll
            for (int s = 0; s < 3; ++s) pa[s] = ring[((it * 16 + s) % 48) * 64 + lane];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (s + 3 < 16) pa[s + 3] = ring[((it * 16 + s + 3) % 48) * 64 + lane];
                acc0 = MF(pa[s], b0[s & 3], acc0);
                acc1 = MF(pa[s], b1[s & 3], acc1);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                acc0 = MF(areg[s & 3], b0[s & 3], acc0);
                acc1 = MF(areg[s & 3], b1[s & 3], acc1);
            }
        }
        if (MODE >= 2) {
            // the finished block: ReLU + round to packed fp16 (what the kernel does once per produced register)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                h2v p = __builtin_bit_cast(h2v, __builtin_amdgcn_cvt_pkrtz(acc0[r], acc0[r + 1]));
                h2v q = __builtin_bit_cast(h2v, __builtin_amdgcn_cvt_pkrtz(acc1[r], acc1[r + 1]));
                h2v z = {(_Float16)0.f, (_Float16)0.f};
                p = __builtin_elementwise_max(p, z);
                q = __builtin_elementwise_max(q, z);
                keep ^= __builtin_bit_cast(unsigned, p) + __builtin_bit_cast(unsigned, q);
            }
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        }
        if (MODE >= 3) {
            // ring refill: this wave's quarter of a 16 KB chunk
            __syncthreads();
            const int chunk = (it + 2) % 3;
#pragma unroll
            for (int k = 0; k < 4; ++k) ring[chunk * 1024 + (wave * 4 + k) * 64 + lane] = src[((it * 16 + wave * 4 + k) % 48) * 64 + lane];
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = (float)keep;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) {
        const int w = blockIdx.x * 4 + wave;
        ticks[2 * w] = c1 - c0;
        ticks[2 * w + 1] = r1 - r0;
    }
}

static unsigned short f2h(float f) { _Float16 h = (_Float16)f; unsigned short u; memcpy(&u, &h, 2); return u; }

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    u32x4* src; float* out; unsigned long long* ticks;
    const int nq = 48 * 64 + 8 * 64;
    hipMalloc(&src, 2 << 20); hipMemset(src, 0, 2 << 20); /* (mode 16 walks 1.44 MB of it) */ hipMalloc(&out, 256 * 256 * 4); hipMalloc(&ticks, 1024 * 16);
    const char* names[3] = {"A random, B random", "A random, B post-ReLU (half zero)", "A small weights (|w|<0.06), B post-ReLU"};
    for (int fill = 0; fill < 3; ++fill) {
        std::vector<unsigned> h(nq * 4);
        srand(7);
        for (size_t i = 0; i < h.size(); ++i) {
            float v[2];
            const bool isb = i >= 48 * 64 * 4;
            for (int k = 0; k < 2; ++k) {
                float r = (float)rand() / RAND_MAX * 2.f - 1.f;
                if (isb) v[k] = fill == 0 ? r : (r > 0.f ? r : 0.f);
                else v[k] = fill == 2 ? r * 0.0625f : r;
            }
            h[i] = f2h(v[0]) | ((unsigned)f2h(v[1]) << 16);
        }
        hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 20; ++mode)
            for (int rep = 0; rep < 3; ++rep) {
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k_mix<0>, dim3(256), dim3(256), 0, 0, src, out, ticks, iters);
                if (mode == 1) hipLaunchKernelGGL(k_mix<1>, dim3(256), dim3(256), 0, 0, src, out, ticks, iters);
                if (mode == 2) hipLaunchKernelGGL(k_mix<2>, dim3(256), dim3(256), 0, 0, src, out, ticks, iters);
                if (mode == 3) hipLaunchKernelGGL(k_mix<3>, dim3(256), dim3(256), 0, 0, src, out, ticks, iters);
                if (mode == 4) hipLaunchKernelGGL(k_mix_asm<1>, dim3(256), dim3(256), 65536, 0, src, out, ticks, iters);
                if (mode == 5) hipLaunchKernelGGL(k_mix_asm<2>, dim3(256), dim3(256), 65536, 0, src, out, ticks, iters);
                if (mode == 6) hipLaunchKernelGGL(k_mix_asm<3>, dim3(256), dim3(256), 65536, 0, src, out, ticks, iters);
                if (mode == 7) hipLaunchKernelGGL(k_mix_asm<4>, dim3(256), dim3(256), 65536, 0, src, out, ticks, iters);
                if (mode == 8) hipLaunchKernelGGL(k_mix_asm<5>, dim3(256), dim3(256), 65536, 0, src, out, ticks, iters);
                if (mode == 9) hipLaunchKernelGGL(k_mix_asm<6>, dim3(256), dim3(256), 65536, 0, src, out, ticks, iters);
                if (mode == 10) hipLaunchKernelGGL(k_mix_asm<7>, dim3(256), dim3(256), 65536, 0, src, out, ticks, iters);
                if (mode == 11) hipLaunchKernelGGL(k_mix_asm<8>, dim3(256), dim3(256), 65536, 0, src, out, ticks, iters);
                if (mode == 12) hipLaunchKernelGGL(k_mix_asm<9>, dim3(256), dim3(256), 65536, 0, src, out, ticks, iters);
                if (mode == 13) hipLaunchKernelGGL(k_mix_asm<10>, dim3(256), dim3(256), 65536, 0, src, out, ticks, iters);
                if (mode == 14) hipLaunchKernelGGL(k_mix_asm<11>, dim3(256), dim3(256), 65536, 0, src, out, ticks, iters);
                if (mode == 15) hipLaunchKernelGGL(k_mix_asm<12>, dim3(256), dim3(256), 65536, 0, src, out, ticks, iters);
                if (mode == 16) hipLaunchKernelGGL(k_mix_asm<13>, dim3(256), dim3(256), 65536, 0, src, out, ticks, iters);
                if (mode == 17) hipLaunchKernelGGL(k_mix_asm<14>, dim3(256), dim3(256), 65536, 0, src, out, ticks, iters);
                if (mode == 18) hipLaunchKernelGGL(k_mix_asm<15>, dim3(256), dim3(256), 65536, 0, src, out, ticks, iters);
                if (mode == 19) hipLaunchKernelGGL(k_mix_asm<16>, dim3(256), dim3(256), 65536, 0, src, out, ticks, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                std::vector<unsigned long long> t(2048);
                hipMemcpy(t.data(), ticks, 2048 * 8, hipMemcpyDeviceToHost);
                double cyc = 0, real = 0;
                for (int w = 0; w < 1024; ++w) { cyc += (double)t[2 * w]; real += (double)t[2 * w + 1]; }
                cyc /= 1024; real /= 1024;
                const double nm = (double)iters * 32;
                if (rep == 2)
                    printf("%-42s mode %d%s: %8.3f ms  %7.1f TFLOP/s   %.2f shader cycles / MFMA   effective clock %.3f GHz\n", names[fill], mode > 3 ? mode - 3 : mode, mode > 3 ? " hand-scheduled" : "               ", ms,
                           256.0 * 4 * nm * 32768.0 / ms / 1e9, cyc / nm, cyc / real * 0.1);
            }
    }
    return 0;
}
