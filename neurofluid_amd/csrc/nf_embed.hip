// nf_embed.hip — standalone positional encoding, models/nerf.py:4-38 (Embedding.forward), for callers that keep the
// reference's own models/renderer.py and drive the modules one by one (INTEGRATION.md level 2).  The fused renderer never
// launches these: its feature kernel (nf_render.hip:emit_pe) emits the same values straight into the MLP operand layout.
//
//   out[b][c]                = x[b][c]
//   out[b][C (1 + 2f) + c]   = sin(2^f x[b][c])
//   out[b][C (2 + 2f) + c]   = cos(2^f x[b][c])          f < n_freqs   (freq_bands = 2^linspace(0, N-1, N), logscale)
//
// Same arithmetic as emit_pe: ONE double-precision sincos per input value, the octaves by angle doubling in double — every
// value is the correctly rounded sin / cos of the reference's exact fp32 argument 2^f x.
#include "nf_common.h"
#include <math.h>

__global__ void __launch_bounds__(256) k_embed_fwd(const float* __restrict__ x, int64_t n_elem, int C, int NF,
                                                   float* __restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_elem) return;
    const int64_t b = e / C;
    const int c = (int)(e - b * C);
    const float v = x[e];
    float* o = out + b * (int64_t)(C * (2 * NF + 1));
    o[c] = v;
    double sn, cs;
    sincos((double)v, &sn, &cs);
    for (int f = 0; f < NF; ++f) {
        o[C * (1 + 2 * f) + c] = (float)sn;
        o[C * (2 + 2 * f) + c] = (float)cs;
        const double s2 = 2.0 * sn * cs, c2 = 1.0 - 2.0 * sn * sn;
        sn = s2; cs = c2;
    }
}

// dL/dx[b][c] = g[b][c] + sum_f 2^f (g_sin cos(2^f x) - g_cos sin(2^f x))      (what autograd derives for :33-36)
__global__ void __launch_bounds__(256) k_embed_bwd(const float* __restrict__ x, const float* __restrict__ g, int64_t n_elem,
                                                   int C, int NF, float* __restrict__ dx)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_elem) return;
    const int64_t b = e / C;
    const int c = (int)(e - b * C);
    const float* gr = g + b * (int64_t)(C * (2 * NF + 1));
    float acc = gr[c];
    double sn, cs;
    sincos((double)x[e], &sn, &cs);
    for (int f = 0; f < NF; ++f) {
        const float fr = (float)(1 << f);
        acc += fr * (gr[C * (1 + 2 * f) + c] * (float)cs - gr[C * (2 + 2 * f) + c] * (float)sn);
        const double s2 = 2.0 * sn * cs, c2 = 1.0 - 2.0 * sn * sn;
        sn = s2; cs = c2;
    }
    dx[e] = acc;
}

extern "C" int nf_embed_fwd(const float* x, int64_t n_rows, int channels, int n_freqs, float* out, nf_stream_t stream)
{
    NF_CHECK_ARG(x && out, "null pointer");
    NF_CHECK_ARG(channels >= 1 && n_freqs >= 0 && n_freqs <= 30, "bad channels / n_freqs");
    const int64_t n = n_rows * channels;
    if (n <= 0) return NF_OK;
    hipLaunchKernelGGL(k_embed_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n, channels, n_freqs, out);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_embed_bwd(const float* x, const float* d_out, int64_t n_rows, int channels, int n_freqs, float* d_x,
                            nf_stream_t stream)
{
    NF_CHECK_ARG(x && d_out && d_x, "null pointer");
    NF_CHECK_ARG(channels >= 1 && n_freqs >= 0 && n_freqs <= 30, "bad channels / n_freqs");
    const int64_t n = n_rows * channels;
    if (n <= 0) return NF_OK;
    hipLaunchKernelGGL(k_embed_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, d_out, n, channels, n_freqs, d_x);
    NF_CHECK_LAUNCH();
    return NF_OK;
}
