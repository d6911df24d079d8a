// nf_mlp_a_kernel.inl — one instantiation of the hand-scheduled MLP kernel (included by nf_mlp_a.hip once per feature-row shape with
// NF_A_QX / NF_A_QD / NF_A_BODY_FILE defined): the C++ only hands the arguments over in SGPRs; the body is the generated asm statement.
#define NF_A_CAT4_(a, b, c, d) a##b##c##d
#define NF_A_CAT4(a, b, c, d) NF_A_CAT4_(a, b, c, d)
#define NF_A_KNAME NF_A_CAT4(k_mlp_fwd_a_, NF_A_QX, _, NF_A_QD)

__global__ void __launch_bounds__(256) NF_A_KNAME(const float* __restrict__ wstream, const float* __restrict__ wsig,
                                                  const float* __restrict__ wrgb, const float* __restrict__ bias,
                                                  const float* __restrict__ X, const int* __restrict__ n_rows,
                                                  const int* __restrict__ row_sample, float4* __restrict__ rgbsigma, int max_rows)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int block = (int)blockIdx.x, nblocks = (int)gridDim.x;
    asm volatile(
#include NF_A_BODY_FILE
        :
        : "s"(wstream), "s"(wsig), "s"(wrgb), "s"(bias), "s"(X), "s"(n_rows), "s"(row_sample), "s"(rgbsigma), "s"(max_rows), "s"(wave),
          "s"(block), "s"(nblocks)
        : "memory", "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54",
          "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73",
          "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91",
          A_R0_255(v), A_R0_255(a));
}
#undef NF_A_KNAME
