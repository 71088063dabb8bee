// Internal (not part of the C ABI): kernel parameter block and tile
// configuration of the fused conv3d kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "ddpm3d.h"

#define DDPM3D_CONV_CK 16  // input channels staged per LDS tile

struct ConvK {
    const float* src0;
    const float* src1;
    const float* affA;
    const float* affB;
    const float* w;
    const float* bias;
    const float* res;
    float* out;
    float* stats;
    int N, D, H, W, Cin, Cout, C0, C1, CinPad, CoutPad;
    int in_mode, act, bias_stride_n, res_mode, out_layout, stats_rows;
    int tilesX, tilesY, tilesZ;
};

struct ConvCfg {
    int KS;        // 3 or 1
    int WN;        // waves along Cout (4, 2, 1); waves along voxels = 4 / WN
    int TXL, TYL;  // log2 of the tile extent in W and H; tile depth = 128 >> (TXL+TYL)
};

static inline int ddpm3d_round_up(int v, int m) { return (v + m - 1) / m * m; }

// One rule, used by the launcher, by ddpm3d_conv_stats_rows and by the weight
// packer (CoutPad / CinPad): how a conv of this shape is tiled.
static inline ConvCfg ddpm3d_conv_cfg(int H, int W, int Cout, int ksize) {
    ConvCfg c;
    c.KS = ksize;
    c.WN = Cout > 64 ? 4 : (Cout > 32 ? 2 : 1);
    if (H >= 8 && W >= 8) { c.TXL = 3; c.TYL = 3; } else { c.TXL = 2; c.TYL = 2; }
    return c;
}
static inline int ddpm3d_cout_pad(int Cout) { return ddpm3d_round_up(Cout, 32); }
static inline int ddpm3d_cin_pad(int Cin) { return ddpm3d_round_up(Cin, DDPM3D_CONV_CK); }

hipError_t ddpm3d_launch_conv_f32(const ConvK& k, const ConvCfg& c, hipStream_t st);
