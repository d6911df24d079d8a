// does gfx950 execute scalar stores (s_store_dwordx2 + s_dcache_wb)?  prints the 64-bit ballots a wave stored without a vector instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* x, unsigned long long* out)
{
    const float v = x[threadIdx.x];
    const unsigned long long m = __builtin_amdgcn_ballot_w64(v > 0.f);
    unsigned long long* p = out + blockIdx.x;
    asm volatile("s_store_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::"s"(m), "s"(p) : "memory");
}
int main()
{
    float h[64];
    for (int i = 0; i < 64; ++i) h[i] = (i % 3 == 0) ? 1.f : -1.f;
    float* d; unsigned long long* o;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 8 * 4); hipMemset(o, 0, 32);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(4), dim3(64), 0, 0, d, o);
    hipError_t e = hipDeviceSynchronize();
    unsigned long long r[4] = {};
    hipMemcpy(r, o, 32, hipMemcpyDeviceToHost);
    unsigned long long want = 0;
    for (int i = 0; i < 64; ++i) if (i % 3 == 0) want |= 1ull << i;
    printf("sync: %s; stored %016llx %016llx %016llx %016llx; want %016llx -> %s\n", hipGetErrorString(e), r[0], r[1], r[2], r[3], want,
           (r[0] == want && r[3] == want) ? "SCALAR STORES WORK" : "MISMATCH");
    return 0;
}
