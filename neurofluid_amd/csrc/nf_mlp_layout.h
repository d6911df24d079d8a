// nf_mlp_layout.h — packed-weight layout of the fp32 NeRF MLP kernels (shared by nf_mlp.hip / nf_mlp_h.hip)
#pragma once
#include "nf_common.h"

// ------------------------------------------------------------------------------------------------
// packed-weight layout
// ------------------------------------------------------------------------------------------------
struct NfMlpLayout {
    int cx, cd, qx, qd;
    int off_x[9];   // layer l (0 = xyz_encoding_1 .. 7 = xyz_encoding_8, 8 = xyz_encoding_final): X-part [qx*4][2][64][4] or -1
    int off_h[9];   // hidden part [128][2][64][4] or -1
    int off_dir_h;  // [128][64][4]
    int off_dir_x;  // [qd*4][64][4]
    int off_wsig;   // [128][2]
    int off_wrgb;   // [3][64][2]
    int off_b[9];   // natural bias vectors (256 each)
    int off_bdir;   // 128
    int off_bsig;   // 1
    int off_brgb;   // 3
    int off_bstep[9];  // bias as one extra K-step per layer: [2][64][4], A = bias (lanes < 32) / 0, B operand = 1.0
    int off_bstep_dir; // [64][4]
    int total;
};

static inline NfMlpLayout mlp_layout(int cx, int cd)
{
    NfMlpLayout L;
    L.cx = cx; L.cd = cd; L.qx = (cx + 7) / 8; L.qd = (cd + 7) / 8;
    int o = 0;
    for (int l = 0; l < 9; ++l) {
        L.off_x[l] = -1; L.off_h[l] = -1;
        if (l == 0 || l == 4) { L.off_x[l] = o; o += L.qx * 4 * 512; }
        if (l != 0) { L.off_h[l] = o; o += 128 * 512; }
    }
    L.off_dir_h = o; o += 128 * 256;
    L.off_dir_x = o; o += L.qd * 4 * 256;
    L.off_wsig = o; o += 256;
    L.off_wrgb = o; o += 384;
    for (int l = 0; l < 9; ++l) { L.off_b[l] = o; o += 256; }
    L.off_bdir = o; o += 128;
    L.off_bsig = o; o += 4;
    L.off_brgb = o; o += 4;
    for (int l = 0; l < 9; ++l) { L.off_bstep[l] = o; o += 512; }
    L.off_bstep_dir = o; o += 256;
    L.total = o;
    return L;
}

// transposed weights of the backward (data-gradient) kernels
// dpre buffer per row: [dpre1..dpre8 (8*256) | dpre_final (256) | dpre_dir (128) | dz_rgb (3) | dsigma (1)]
struct NfMlpLayoutT {
    int off_h[9];   // l = 1..8: W_l^T hidden part [128 steps][2][64][4]   (index 0 unused)
    int off_dir;    // W_dir[:, :256]^T  [64 steps][2][64][4]
    // round 6, the feature gradient dL/dX inside the tile-per-workgroup backward (nf_nerf_mlp_bwd_n3): the three weight blocks that
    // read the feature row, transposed like the hidden parts, output features (rows of X) padded with zeros to 8 blocks of 32
    int off_dx0;    // W_1[:, :cx]^T          [128 steps][2][64][4]
    int off_dx4;    // W_5[:, :cx]^T          [128 steps][2][64][4]
    int off_dxd;    // W_dir[:, 256:256+cd]^T [64 steps][2][64][4]
    int total;
};

static inline NfMlpLayoutT mlp_layout_t()
{
    NfMlpLayoutT T;
    int o = 0;
    T.off_h[0] = -1;
    for (int l = 1; l < 9; ++l) { T.off_h[l] = o; o += 128 * 512; }
    T.off_dir = o; o += 64 * 512;
    T.off_dx0 = o; o += 128 * 512;
    T.off_dx4 = o; o += 128 * 512;
    T.off_dxd = o; o += 64 * 512;
    T.total = o;
    return T;
}


// feature held by register r of block b in half-wave h (MFMA 32x32 C/D layout)
__device__ __host__ __forceinline__ int frag_feature(int b, int r, int h) { return 32 * b + (r & 3) + 8 * (r >> 2) + 4 * h; }


struct NfNerfPtrs {
    const float* w[12];
    const float* b[12];
};

// Stops LICM from hoisting the (loop-invariant) head-weight loads out of the persistent tile loop, where they
// would cost hundreds of live registers.
// (An opaque zero OFFSET rather than an opaque pointer: laundering the pointer itself drops its
// global address space and turns every load into flat_load + vmcnt(0)/lgkmcnt(0) waits.)
__device__ __forceinline__ int opaque_zero()
{
    int z = 0;
    asm volatile("" : "+s"(z));
    return z;
}
