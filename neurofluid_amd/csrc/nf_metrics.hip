// nf_metrics.hip — image metric of the evaluation callers: SSIM with an 11x11 gaussian window (sigma 1.5), no padding,
// as /root/reference/utils/evaluate_images.ipynb cell 5 (class SSIM) computes it with five grouped conv2d calls:
//   mu1 = w * p, mu2 = w * g, s11 = w * p^2 - mu1^2, s22 = w * g^2 - mu2^2, s12 = w * (p g) - mu1 mu2
//   ssim = (2 mu1 mu2 + C1)(2 s12 + C2) / ((mu1^2 + mu2^2 + C1)(s11 + s22 + C2)),  C1 = (0.01 L)^2, C2 = (0.03 L)^2
// One pass over the two images: a 16x16 tile of outputs per workgroup, the 26x26 input patches in LDS, the window applied
// separably (rows, then columns: 22 taps per map instead of 121), the five maps never leave the chip.  Per-tile sums go to
// `partial`, a second tiny kernel adds them per image in a fixed order (deterministic).
#include "nf_common.h"

#define SS_W 11
#define SS_T 16
#define SS_P (SS_T + SS_W - 1)      // 26

struct SsimWin { float g[SS_W]; };

__global__ void __launch_bounds__(SS_T * SS_T) k_ssim_tiles(const float* __restrict__ pred, const float* __restrict__ gt, int H, int W,
                                                            SsimWin win, float C1, float C2, float* __restrict__ partial)
{
    __shared__ float sp[SS_P][SS_P + 1], sg[SS_P][SS_P + 1];
    __shared__ float hm[5][SS_P][SS_T + 1];
    __shared__ float red[SS_T * SS_T / 64];
    const int plane = blockIdx.z;
    const float* p = pred + (size_t)plane * H * W;
    const float* g = gt + (size_t)plane * H * W;
    const int x0 = blockIdx.x * SS_T, y0 = blockIdx.y * SS_T;
    const int tid = threadIdx.x;
    for (int e = tid; e < SS_P * SS_P; e += SS_T * SS_T) {
        const int r = e / SS_P, c = e - r * SS_P;
        const int y = y0 + r, x = x0 + c;
        const bool in = y < H && x < W;
        sp[r][c] = in ? p[(size_t)y * W + x] : 0.f;
        sg[r][c] = in ? g[(size_t)y * W + x] : 0.f;
    }
    __syncthreads();
    for (int e = tid; e < SS_P * SS_T; e += SS_T * SS_T) {       // rows: 26 x 16 outputs of the horizontal pass, 5 maps
        const int r = e / SS_T, c = e - r * SS_T;
        float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
        for (int k = 0; k < SS_W; ++k) {
            const float pv = sp[r][c + k], gv = sg[r][c + k], w = win.g[k];
            a += w * pv; b += w * gv; aa += w * (pv * pv); bb += w * (gv * gv); ab += w * (pv * gv);
        }
        hm[0][r][c] = a; hm[1][r][c] = b; hm[2][r][c] = aa; hm[3][r][c] = bb; hm[4][r][c] = ab;
    }
    __syncthreads();
    const int ox = tid & (SS_T - 1), oy = tid / SS_T;
    float v = 0.f;
    if (x0 + ox < W - SS_W + 1 && y0 + oy < H - SS_W + 1) {
        float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < SS_W; ++k) {
            const float w = win.g[k];
            m1 += w * hm[0][oy + k][ox]; m2 += w * hm[1][oy + k][ox];
            e11 += w * hm[2][oy + k][ox]; e22 += w * hm[3][oy + k][ox]; e12 += w * hm[4][oy + k][ox];
        }
        const float m11 = m1 * m1, m22 = m2 * m2, m12 = m1 * m2;
        const float s11 = e11 - m11, s22 = e22 - m22, s12 = e12 - m12;
        const float v1 = 2.0f * s12 + C2, v2 = s11 + s22 + C2;
        v = ((2.f * m12 + C1) * v1) / ((m11 + m22 + C1) * v2);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int k = 0; k < SS_T * SS_T / 64; ++k) s += red[k];
        partial[((size_t)plane * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = s;
    }
}

// one workgroup per image: fixed-order sum of its C * tiles partials in double, divided by the number of outputs
__global__ void __launch_bounds__(256) k_ssim_reduce(const float* __restrict__ partial, int per_image, double inv_count,
                                                     float* __restrict__ out)
{
    __shared__ double red[256];
    const float* p = partial + (size_t)blockIdx.x * per_image;
    double s = 0.0;
    for (int e = threadIdx.x; e < per_image; e += 256) s += (double)p[e];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = (float)(red[0] * inv_count);
}

extern "C" size_t nf_image_ssim_workspace_floats(int B, int C, int H, int W)
{
    if (B < 1 || C < 1 || H < SS_W || W < SS_W) return 0;
    const size_t tx = (size_t)(W - SS_W + 1 + SS_T - 1) / SS_T, ty = (size_t)(H - SS_W + 1 + SS_T - 1) / SS_T;
    return (size_t)B * C * tx * ty;
}

extern "C" int nf_image_ssim(const float* pred, const float* gt, int B, int C, int H, int W, const float window[11], float L,
                             float* workspace, float* ssim_per_image, nf_stream_t stream)
{
    NF_CHECK_ARG(pred && gt && window && workspace && ssim_per_image, "null pointer");
    NF_CHECK_ARG(B >= 1 && C >= 1 && H >= SS_W && W >= SS_W && (long)B * C <= 65535, "need B, C >= 1, H, W >= 11, B*C <= 65535");
    SsimWin win;
    for (int k = 0; k < SS_W; ++k) win.g[k] = window[k];
    const int tx = (W - SS_W + 1 + SS_T - 1) / SS_T, ty = (H - SS_W + 1 + SS_T - 1) / SS_T;
    const float C1 = (0.01f * L) * (0.01f * L), C2 = (0.03f * L) * (0.03f * L);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_ssim_tiles, dim3(tx, ty, B * C), dim3(SS_T * SS_T), 0, st, pred, gt, H, W, win, C1, C2, workspace);
    const double inv = 1.0 / ((double)C * (H - SS_W + 1) * (W - SS_W + 1));
    hipLaunchKernelGGL(k_ssim_reduce, dim3(B), dim3(256), 0, st, (const float*)workspace, C * tx * ty, inv, ssim_per_image);
    NF_CHECK_LAUNCH();
    return NF_OK;
}
