// Internal (not part of the C ABI): kernel parameter block and tile
// configuration of the fused conv3d kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdlib.h>
#include "ddpm3d.h"

#define DDPM3D_CONV_CK 16      // input channels staged per LDS tile
#define DDPM3D_REDUCE_VOX 16   // voxels per workgroup (= per statistics row) of the split-K reduce
// split-f16 modes: each output channel's weights are multiplied by 2^floor(log2(W_TARGET / max|w|))
// at pack time; the activations of a launch by the power of two that puts the caller's bound of
// their magnitude just below 2^15 (ddpm3d_conv_desc.in_bound, act_scale() in conv3d_load.h); the
// epilogue multiplies by the exact inverses.  Keeps lo = x - f16(x) a NORMAL f16 for all but
// negligibly small operands and hi finite for EVERY input (no clamp anywhere).
#define DDPM3D_X3_W_TARGET 8.0f

struct ConvK {
    const float* src0;
    const float* src1;
    const float* affA;
    const float* affB;
    const float* w;
    const float* bias;
    const float* res;
    float* out;
    double* stats;   // [N][Cout][stats_rows][2] fp64 (sum, sum of squares)
    float* partial;  // split-K slabs [ksplit][N*D*H*W][Cout], raw accumulators
    const float* wscale;  // split-f16 modes: per-cout 1 / weight scale
    const float* in_bound;    // ddpm3d_conv_desc.in_bound (split-f16 modes)
    int in_bound_count, in_bound_stride;
    unsigned w_bytes, src0_bytes, src1_bytes;  // extents for the buffer resource descriptors
    int N, D, H, W, Cin, Cout, C0, C1, CinPad, CoutPad;
    int in_mode, act, bias_stride_n, res_mode, out_layout, stats_rows;
    int tilesX, tilesY, tilesZ;
    int ksplit, chunks_per_split;
    int wstat;  // weight-stationary workgroup -> XCD order (conv3d_load.h wg_id)
    int reduce_vox;  // voxels per statistics row of the split-K reduce
    int hint;        // ddpm3d_conv_desc.kernel_hint
    int io;          // ddpm3d_conv_desc.io_dtype (DDPM3D_IO_* bits)
};

struct ConvCfg {
    int PREC;      // 0 exact fp32 MFMA, 1 split-f16 (3 MFMA per product)
    int KS;        // 3 or 1
    int WN;        // waves along Cout (4, 2, 1); waves along voxels = 4 / WN
    int MT;        // 32-voxel accumulators per wave; tile = (4/WN) * MT * 32 voxels (128 or 256)
    int TXL, TYL;  // log2 of the tile extent in W and H; tile depth = 128 >> (TXL+TYL)
    int S;         // split-K factor over the Cin chunks (1 = none)
    int tilesX, tilesY, tilesZ;
    int stats_rows;           // rows per sample the executing path writes
    int reduce_vox;           // voxels per statistics row of the split-K reduce (16, 8 or 4)
    size_t workspace_bytes;   // slabs needed when S > 1
};

static inline int ddpm3d_round_up(int v, int m) { return (v + m - 1) / m * m; }
static inline int ddpm3d_cout_pad(int Cout) { return ddpm3d_round_up(Cout, 32); }
static inline int ddpm3d_cin_pad(int Cin) { return ddpm3d_round_up(Cin, DDPM3D_CONV_CK); }

// Fill the split-dependent fields for a split factor S (the rule's own, or a forced one).
static inline void ddpm3d_conv_cfg_split(ConvCfg& c, int S, int N, int D, int H, int W, int Cout) {
    const long long vox = (long long)D * H * W;
    c.S = S;
    if (S > 1) {
        c.stats_rows = (int)((vox + c.reduce_vox - 1) / c.reduce_vox);
        c.workspace_bytes = (size_t)S * N * vox * Cout * sizeof(float);
    } else {
        c.stats_rows = c.tilesZ * c.tilesY * c.tilesX * (4 / c.WN);
        c.workspace_bytes = 0;
    }
}

// One rule, used by the launcher, by ddpm3d_conv_stats_rows /
// ddpm3d_conv_workspace_bytes and by the weight packer: how a conv of this
// shape is tiled and whether its reduction dimension is split over workgroups.
//
// Split-K: a low-resolution level has few 128-voxel tiles (64x4x4 -> 8), so
// without it a 512->512 conv occupies 32 of 256 CUs.  The Cin chunks are dealt
// to S workgroups per tile; the cost model below is (workgroups on the busiest
// CU) x (chunks per workgroup + fixed cost) / (MFMA efficiency at that
// co-residency), efficiencies measured on MI355X with the unsplit kernel.
// prec = DDPM3D_PREC_* of the call (ABI 12): the one-MFMA modes (f16, bf16) spend a third of the split-f16
// modes' matrix time per chunk against the same fixed costs, so their split rule has constants of its own.
static inline ConvCfg ddpm3d_conv_cfg(int N, int D, int H, int W, int Cin, int Cout, int ksize, int prec) {
    ConvCfg c;
    c.PREC = prec;
    const bool one_mfma = prec == DDPM3D_PREC_F16 || prec == DDPM3D_PREC_F16_WZ || prec == DDPM3D_PREC_BF16 ||
                          prec == DDPM3D_PREC_BF16_WZ;
    c.KS = ksize;
    c.WN = Cout > 64 ? 4 : (Cout > 32 ? 2 : 1);
#ifdef DDPM3D_FORCE_T4_3X3   // measurement only: 4x4x8 tiles for every 3x3x3 layer (r03: slower)
    if (H >= 8 && W >= 8 && ksize != 3) { c.TXL = 3; c.TYL = 3; } else { c.TXL = 2; c.TYL = 2; }
#else
    if (H >= 8 && W >= 8) { c.TXL = 3; c.TYL = 3; } else { c.TXL = 2; c.TYL = 2; }
#endif
    c.MT = c.WN;  // 128-voxel tile (a 256-voxel tile, 8 accumulators per wave, measured 2.2x slower: r01)
    const int TX = 1 << c.TXL, TY = 1 << c.TYL, TZ = (4 / c.WN) * c.MT * 32 / (TX * TY);
    c.tilesX = (W + TX - 1) / TX;
    c.tilesY = (H + TY - 1) / TY;
    c.tilesZ = (D + TZ - 1) / TZ;
    const long long mtiles = (long long)N * c.tilesZ * c.tilesY * c.tilesX;
    const long long blocks = mtiles * ((ddpm3d_cout_pad(Cout) + 32 * c.WN - 1) / (32 * c.WN));
    const int nch = ddpm3d_cin_pad(Cin) / DDPM3D_CONV_CK;
    int best = 1;
    if (ksize == 1 && c.WN == 4) {
        // 1x1: few workgroups each walking many chunks are latency-bound, and the split pays once that walk
        // outweighs a second launch.  Units ~us, fitted to the forced-split sweeps of the network's fourteen
        // skip-connection shapes in the f16x3 and bf16 modes (conv1x1.hip; profiles/r03_splitk_sweep_1x1_*.txt:
        // the rule's choices cost 0.576 ms over both tables, the best choice per cell 0.567): chunk 0.25,
        // prologue + epilogue 6, the reduce launch 6.  (r01-r02, general kernel: 1 / 3 / 10.)
        double best_cost = 1e300;
        for (int s = 1; s <= 32 && s <= nch; ++s) {
            const int cps = (nch + s - 1) / s;
            if (s > 1 && cps < 4) break;
            const long long per_cu = (blocks * s + 255) / 256;
            const double cost = (double)per_cu * (0.25 * cps + 6.0) + (s > 1 ? 6.0 : 0.0);
            if (cost < best_cost * 0.9) { best_cost = cost; best = s; }
        }
    }
    if (ksize == 3 && c.WN == 4 && !one_mfma) {
        static const double eff[4] = {1.0, 0.62, 0.78, 0.82};
        double best_cost = 1e300;
        // (at most 16 ways, and a 1 % instead of a 3 % hysteresis: the model does not price the reduce kernel's
        // S slab reads; the one shape it sent to 32 -- 1024 -> 384 @ 64x4x4 -- is 11 % / 23 % faster at 16 in the
        // f16x3 / bf16 forms.  Over the twenty shapes of profiles/r03_splitk_sweep_with_4x4x8.txt the rule's
        // choices cost 1.693 ms, the best factor per cell 1.681.)
        for (int s = 1; s <= 16 && s <= nch; ++s) {
            const int cps = (nch + s - 1) / s;
            if (s > 1 && cps < 2) break;
            const long long per_cu = (blocks * s + 255) / 256;
            const double e = eff[per_cu > 3 ? 3 : (int)per_cu];
            const double cost = (double)per_cu * (cps + (s > 1 ? 0.75 : 0.5)) / e;
            if (cost < best_cost * 0.99) { best_cost = cost; best = s; }   // (0.97 until r03: see the cap's note)
        }
    }
    if (ksize == 3 && c.WN == 4 && one_mfma) {
        // f16 / bf16 (r04, ABI 12): a chunk is ~3x shorter, so the fixed cost of a split workgroup (prologue,
        // slab epilogue) weighs 3 chunks and the reduce launch's S slab reads are priced (0.2 chunk-times per
        // S x MB of output).  Fitted to the forced-split sweep of the twenty shapes in the bf16 form
        // (profiles/r03_splitk_sweep_bf16.txt; scratch fit: the shape-only rule's choices cost 0.911 ms there,
        // these 0.886, the best factor per cell 0.879).  Candidates are the sweep's own columns.
        static const double eff[4] = {1.0, 0.5, 1.0, 0.9};
        static const int cand[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16};
        const double out_mb = (double)N * D * H * W * Cout * 4.0 / 1e6;
        double best_cost = 1e300;
        for (int i = 0; i < 10 && cand[i] <= nch; ++i) {
            const int s = cand[i];
            const int cps = (nch + s - 1) / s;
            if (s > 1 && cps < 2) break;
            const long long per_cu = (blocks * s + 255) / 256;
            const double e = eff[per_cu > 3 ? 3 : (int)per_cu];
            const double cost = (double)per_cu * (cps + (s > 1 ? 3.0 : 0.25)) / e + (s > 1 ? 0.2 * s * out_mb : 0.0);
            if (cost < best_cost * 0.99) { best_cost = cost; best = s; }
        }
    }
    const long long vox = (long long)D * H * W;
    // voxels per workgroup (= per statistics row) of the split-K reduce: 16, or fewer on the
    // small levels so that the reduce still launches >= 1024 workgroups (r01: the 64x4x4
    // level ran it on 64 workgroups, 23 us a launch, 6.6 % of the forward)
    c.reduce_vox = DDPM3D_REDUCE_VOX;
    while (c.reduce_vox > 4 &&
           (long long)N * ((vox + c.reduce_vox - 1) / c.reduce_vox) * ((Cout / 4 + 63) / 64) < 1024)
        c.reduce_vox /= 2;
    ddpm3d_conv_cfg_split(c, best, N, D, H, W, Cout);
    return c;
}

// bytes of the packed weight image: fp32 [tap][ci/8][CoutPad][8] for PREC 0;
// f16 [tap][ci/16][hi|lo][CoutPad][16] followed by CoutPad fp32 output scales for PREC 1
static inline size_t ddpm3d_packed_bytes(int Cout, int Cin, int ksize, int prec) {
    // Winograd-D form: 4 transformed depth taps x 3 x 3 = 36 instead of 27
    const size_t taps = (prec == DDPM3D_PREC_F16X3_WZ || prec == DDPM3D_PREC_F16_WZ || prec == DDPM3D_PREC_BF16_WZ)
                            ? 36 : (size_t)ksize * ksize * ksize;
    const size_t body = taps * ddpm3d_cin_pad(Cin) * ddpm3d_cout_pad(Cout) * 4;
    return prec != 0 ? body + (size_t)ddpm3d_cout_pad(Cout) * 4 : body;  // modes 1 and 2 share the image
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: raise it once
// per (kernel instantiation, device), not once per process (a process that drives two GPUs would
// otherwise fail its first > 64 KB launch on the second one).  `done` = one flag word per kernel.
struct DynLdsOnce { unsigned long long mask[4]; };   // 256 devices
static inline hipError_t ddpm3d_allow_dynamic_lds(DynLdsOnce& done, const void* kernel, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    dev &= 255;
    unsigned long long& w = done.mask[dev >> 6];
    const unsigned long long bit = 1ull << (dev & 63);
    if (__atomic_load_n(&w, __ATOMIC_ACQUIRE) & bit) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) __atomic_fetch_or(&w, bit, __ATOMIC_RELEASE);
    return e;
}

hipError_t ddpm3d_launch_conv(const ConvK& k, const ConvCfg& c, hipStream_t st);
hipError_t ddpm3d_launch_splitk_reduce(const ConvK& k, hipStream_t st);
hipError_t ddpm3d_launch_conv_skinny(const ConvK& k, int prec, hipStream_t st);   // conv3d_skinny.hip
hipError_t ddpm3d_launch_conv_pw(const ConvK& k, const ConvCfg& c, hipStream_t st);   // conv1x1.hip
// the 1x1x1 convs on raw inputs (ResBlock skip connections) that conv1x1.hip's register-fed GEMM takes;
// k is the filled launch record (chunks_per_split included)
// (r03 kept the split-f16 form above 1024 workgroups -- the three 256 -> 128 skip convs at 64^3 -- on conv3d.hip's
// kernel, 0.110 against 0.120 ms; with the lean epilogue (conv3d_epilogue.h, r04) a 1x1 workgroup is 7 k cycles
// shorter and the order is the other way round, 0.108 against 0.115: profiles/r04_lib_ab_conv1x1_cut_lean_epilogue.txt)
static inline bool ddpm3d_pw_ok(const ConvK& k, const ConvCfg& c, int ksize) {
    const bool s0 = (k.io & DDPM3D_IO_SRC0_BF16) != 0, s1 = (k.io & DDPM3D_IO_SRC1_BF16) != 0;
    return ksize == 1 && (c.PREC == 1 || c.PREC == 2 || c.PREC == 5) && c.WN == 4 && c.MT == 4 &&
           k.in_mode == DDPM3D_IN_SAME && k.affA == nullptr && k.act == 0 && k.stats == nullptr &&
           k.Cout % 128 == 0 && k.Cin % 32 == 0 && k.C0 % 32 == 0 && (k.C1 == 0 || s0 == s1) &&
           (k.ksplit == 1 || k.chunks_per_split % 2 == 0);
}
bool ddpm3d_skinny_ok(int CinPad, int prec, int src16);   // src16: 0 fp32, 1 bf16, 2 f16

// GroupNorm partial sums are STORED in fp64 (r03).  With fp32 sums the variance E[x^2] - mean^2 loses
// |mean|^2 / var x 1e-7 of its value to cancellation -- invisible on normalised data, 1e-2 on a tensor
// whose mean is 300 standard deviations (un-normalised PET counts behind the first conv,
// scripts/test.py:201-203).  A lane ACCUMULATES in fp32 around a pivot -- the first value it sees --
//     s1 = sum (x - p),  s2 = sum (x - p)^2          (no cancellation: x - p is of the size of the spread)
// and converts once, in fp64:  sum x = n p + s1,  sum x^2 = s2 + 2 p s1 + n p^2.  Three fp32 instructions
// per element (fp64 accumulation per element cost the HBM-bound 1x1 and pooled-input kernels 2-10 %:
// profiles/r03_tree_ab_r02_vs_r03_*.txt).  -DDDPM3D_STATS_F64: per-element fp64 accumulation (measurement).
struct GnAcc {
#ifdef DDPM3D_STATS_F64
    double a1, a2;
    __device__ __forceinline__ void init(float) { a1 = 0.0; a2 = 0.0; }
    __device__ __forceinline__ void add(float v) { const double d = (double)v; a1 += d; a2 = __builtin_fma(d, d, a2); }
    __device__ __forceinline__ double sum1(float) const { return a1; }
    __device__ __forceinline__ double sum2(float) const { return a2; }
#else
    float p, s1, s2;
    __device__ __forceinline__ void init(float pivot) { p = pivot; s1 = 0.0f; s2 = 0.0f; }
    __device__ __forceinline__ void add(float v) { const float d = v - p; s1 += d; s2 = __builtin_fmaf(d, d, s2); }
    // n = number of values added
    __device__ __forceinline__ double sum1(float n) const { return (double)n * (double)p + (double)s1; }
    __device__ __forceinline__ double sum2(float n) const {
        const double dp = (double)p;
        return (double)s2 + 2.0 * dp * (double)s1 + (double)n * dp * dp;
    }
#endif
};

// Residual term of the conv epilogue for output element (n, z, y, x, cout);
// shared by the conv kernel and the split-K reduce kernel.
// element e of an activation tensor that holds fp32 or (b16) 16-bit values: bf16, or (f16) IEEE f16
__device__ __forceinline__ float ddpm3d_half_to_float(unsigned short h, bool f16) {
    if (f16) return (float)__builtin_bit_cast(_Float16, h);
    return __builtin_bit_cast(float, (unsigned)h << 16);
}
__device__ __forceinline__ float ddpm3d_act_load(const float* base, size_t e, bool b16, bool f16) {
    if (b16) return ddpm3d_half_to_float(reinterpret_cast<const unsigned short*>(base)[e], f16);
    return base[e];
}
// fp32 -> bf16, round to nearest even, NaN stays NaN (hipcc lowers the cast to v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned short ddpm3d_to_bf16(float v) {
    return __builtin_bit_cast(unsigned short, (__bf16)v);
}
// fp32 -> the tensor's 16-bit type, round to nearest even
__device__ __forceinline__ unsigned short ddpm3d_to_half(float v, bool f16) {
    if (f16) return __builtin_bit_cast(unsigned short, (_Float16)v);
    return ddpm3d_to_bf16(v);
}
__device__ __forceinline__ void ddpm3d_act_store(float* base, size_t e, float v, bool b16, bool f16) {
    if (b16) reinterpret_cast<unsigned short*>(base)[e] = ddpm3d_to_half(v, f16);
    else base[e] = v;
}

__device__ __forceinline__ float ddpm3d_residual(const ConvK& p, int n, int z, int y, int x, int cout) {
    const bool b16 = (p.io & DDPM3D_IO_RES_BF16) != 0, f16 = (p.io & DDPM3D_IO_HALF_IS_F16) != 0;
    if (p.res_mode == DDPM3D_RES_SAME) {
        const size_t vox = (((size_t)n * p.D + z) * p.H + y) * p.W + x;
        return ddpm3d_act_load(p.res, vox * p.Cout + cout, b16, f16);
    }
    if (p.res_mode == DDPM3D_RES_UP) {
        const int Hr = p.H / 2, Wr = p.W / 2;
        const size_t rv = (((size_t)n * p.D + z) * Hr + (y >> 1)) * Wr + (x >> 1);
        return ddpm3d_act_load(p.res, rv * p.Cout + cout, b16, f16);
    }
    if (p.res_mode == DDPM3D_RES_POOL) {
        // AvgPool3d window order (h, w): ((r00 + r01) + r10) + r11, then * 1/4
        const int Hr = p.H * 2, Wr = p.W * 2;
        const size_t rv = (((size_t)n * p.D + z) * Hr + 2 * y) * Wr + 2 * x;
        const size_t e = rv * p.Cout + cout;
        const float r = ((ddpm3d_act_load(p.res, e, b16, f16) + ddpm3d_act_load(p.res, e + p.Cout, b16, f16)) +
                         ddpm3d_act_load(p.res, e + (size_t)Wr * p.Cout, b16, f16)) +
                        ddpm3d_act_load(p.res, e + (size_t)Wr * p.Cout + p.Cout, b16, f16);
        return r * 0.25f;
    }
    return 0.0f;
}
