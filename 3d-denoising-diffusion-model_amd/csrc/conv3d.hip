// Fused 3-D convolution for gfx950 (MI355X): implicit GEMM on the matrix cores
// with the GroupNorm-apply / SiLU / FiLM / pool / upsample / concat prologue
// and the bias / residual / GroupNorm-statistics epilogue fused in.
//
//   GEMM view      M = output voxels, N = Cout, K = taps * Cin
//   workgroup      256 threads = 4 waves, tile = 128 voxels x (32*WN) couts
//   wave (wm, wn)  MT 32x32 accumulators: 32*MT voxels x 32 couts
//   A operand      input halo tile [voxel][16 ci] staged ONCE per ci-chunk in
//                  LDS (already normalised+activated), read per tap at a
//                  constant LDS offset (ds_read_b128)
//   B operand      packed weights streamed L2 -> VGPR (global_load_dwordx4),
//                  private to the wave's 32 couts, prefetched one tap ahead
//   split-K        blockIdx.z owns a range of the ci-chunks (low-res levels)
//
// Two arithmetic modes (template PREC), same fp32 inputs, outputs and
// accumulators:
//   PREC 0  v_mfma_f32_32x32x2_f32: exact fp32 products (bitwise an fmaf chain),
//           64 FLOP/clk/SIMD.
//   PREC 1  each fp32 operand is split x = hi + lo into two f16 (after a
//           power-of-two scaling that keeps lo out of the f16 subnormals) and
//           a*b is evaluated as hi*hi + hi*lo + lo*hi on
//           v_mfma_f32_32x32x16_f16 (every f16 x f16 product is exact in fp32).
//           Operand representation error ~2^-23 relative -- below the fp32
//           accumulation error both modes share -- at 16/3 the MFMA rate.
//
// K order inside a channel block is permuted in PREC 0 (lane half h supplies
// channel 4h+s at step s) so both operands are 16-byte loads; the weight packer
// (ops.hip) stores the order each mode reads.
#include <hip/hip_runtime.h>
#include "conv3d_load.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

// smallest x >= v with x % 16 == res
constexpr int pad_to_residue(int v, int res) { return v + ((res - v % 16) + 16) % 16; }

// LDS strides (in 16-byte slots) of the halo image; see the bank-conflict note in the kernel.
template <int TX, int HX, int HY>
struct LdsGeom {
    static constexpr int RY = pad_to_residue(HX * 5, TX == 8 ? 8 : 4);
    static constexpr int RZ = TX == 8 ? HY * RY : pad_to_residue(HY * RY, 0);
};

template <int PREC, int KS, int WN, int TXL, int TYL>
__global__ __launch_bounds__(256, 2) void conv3d_kernel(const ConvK p) {
    constexpr int CK = DDPM3D_CONV_CK;
    constexpr int WM = 4 / WN;
    constexpr int MT = 4 / WM;  // 32-row accumulators per wave; tile is always 128 voxels
    constexpr int TX = 1 << TXL, TY = 1 << TYL;
    constexpr int TZ = 128 / (TX * TY);
    constexpr int PAD = KS / 2;
    constexpr int HX = TX + 2 * PAD, HY = TY + 2 * PAD, HZ = TZ + 2 * PAD;
    constexpr int HV = HX * HY * HZ;
    // LDS image of the halo tile.  One voxel = 80 bytes = 5 slots of 16 B in both modes:
    //   PREC 0: 16 floats + 4 pad;  PREC 1: 16 f16 hi | 16 f16 lo | 8 f16 pad.
    // A ds_read_b128 is served in four fixed 16-lane groups, one LDS cycle each when the
    // 16 lanes hit 16 different slots of the 256-B bank row.  A group's lanes are four
    // half-rows of 4 consecutive x, so its slots are {0,5,10,15} + offset per half-row and
    // the four offsets must be {0,4,8,12} in some order: that holds when the x-row stride
    // is == 8 (8x8 tiles) or == 4 with a z-plane stride == 0 (4x4 tiles), mod 16 slots.
    // The halo rows / planes are padded to those residues (measured before the padding:
    // 2/3 of all LDS cycles were bank-conflict cycles, 3-way on every fragment read).
    constexpr int VS = 5;                                       // slots per voxel
    constexpr int RY = LdsGeom<TX, HX, HY>::RY;                  // slots per x-row
    constexpr int RZ = LdsGeom<TX, HX, HY>::RZ;                  // slots per z-plane
    constexpr int QPV = CK / 4;
    constexpr int NT = KS * KS * KS;

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5;

    int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tx_i = tile % p.tilesX; tile /= p.tilesX;
    const int ty_i = tile % p.tilesY; tile /= p.tilesY;
    const int tz_i = tile % p.tilesZ; tile /= p.tilesZ;
    const int n = tile;
    const int x0 = tx_i * TX, y0 = ty_i * TY, z0 = tz_i * TZ;

    // per-lane A row byte offsets at tap (0,0,0), including this half's 16-byte k-slot
    int arow[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = (wm * MT + t) * 32 + (lane & 31);
        const int tx = m & (TX - 1), ty = (m >> TXL) & (TY - 1), tz = m >> (TXL + TYL);
        arow[t] = (tz * RZ + ty * RY + tx * VS + half) * 16;
    }

    const int cout = blockIdx.y * (32 * WN) + wn * 32 + (lane & 31);
    const bool wave_active = (blockIdx.y * (32 * WN) + wn * 32) < p.CoutPad;
    const int cout_ld = wave_active ? cout : 0;
    // Weight stream in 16-byte units.
    //   PREC 0: [tap][ci/8][CoutPad][8 f32]            -> 2 units per (8-ci block, cout)
    //   PREC 1: [tap][ci/16][hi|lo][CoutPad][16 f16]   -> 2 units per (16-ci block, part, cout)
    const uint4* wbase = reinterpret_cast<const uint4*>(p.w) + ((size_t)cout_ld * 2 + half);
    const size_t wpart = (size_t)p.CoutPad * 2;                       // PREC0: next 8-ci block; PREC1: hi -> lo
    const size_t wchunk_stride = 2 * wpart;                           // one 16-ci chunk
    const size_t wtap_stride = (size_t)(p.CinPad / CK) * wchunk_stride;

    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;

    const int nchunks = p.CinPad / CK;
    // split-K: blockIdx.z owns a contiguous range of the Cin chunks
    const int chunk_begin = blockIdx.z * p.chunks_per_split;
    const int chunk_end = min(nchunks, chunk_begin + p.chunks_per_split);
    for (int chunk = chunk_begin; chunk < chunk_end; ++chunk) {
        __syncthreads();  // everyone done reading the previous chunk's tile
        // ------------------------------------------------ stage the halo tile
#ifdef DDPM3D_ABL_NO_RESTAGE  // (timing experiments only: stage the first chunk, then reuse it)
        if (chunk == chunk_begin)
#endif
        {
            const int q = tid % QPV;  // fixed per thread: 256 % QPV == 0
            const HaloSrc hs = halo_src<CK>(p, n, chunk, q);
            for (int idx = tid; idx < HV * QPV; idx += 256) {
                const int hv = idx / QPV;
                const int hz = hv / (HY * HX);
                const int rem = hv - hz * (HY * HX);
                const int hy = rem / HX;
                const int hx = rem - hy * HX;
                const f32x4 v = halo_fetch<PREC == 1>(p, hs, n, z0 - PAD + hz, y0 - PAD + hy, x0 - PAD + hx, q,
                                                      chunk == 0);
                unsigned char* vrow = lds + (hz * RZ + hy * RY + hx * VS) * 16;
                if constexpr (PREC == 0) {
                    *reinterpret_cast<f32x4*>(vrow + q * 16) = v;
                } else {
                    h4 hi, lo;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        // x8: keeps lo = x - hi a normal f16 down to |x| ~ 2^-6; clamp keeps hi finite
                        const float s = fminf(fmaxf(v[i] * DDPM3D_X3_ACT_SCALE, -60000.0f), 60000.0f);
                        hi[i] = (_Float16)s;
                        lo[i] = (_Float16)(s - (float)hi[i]);
                    }
                    *reinterpret_cast<h4*>(vrow + q * 8) = hi;
                    *reinterpret_cast<h4*>(vrow + 32 + q * 8) = lo;
                }
            }
        }
        __syncthreads();
        if (!wave_active) continue;

        // ------------------------------------------------ taps x k-steps
        const uint4* wchunk = wbase + (size_t)chunk * wchunk_stride;
        uint4 bcur[2], bnxt[2];
        bcur[0] = wchunk[0];
        bcur[1] = wchunk[wpart];
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
#ifndef DDPM3D_ABL_NO_BSTREAM  // (timing experiments only: reuse tap 0's weights)
            if (tap + 1 < NT) {
                bnxt[0] = wchunk[(size_t)(tap + 1) * wtap_stride];
                bnxt[1] = wchunk[(size_t)(tap + 1) * wtap_stride + wpart];
            }
#else
            bnxt[0] = bcur[0];
            bnxt[1] = bcur[1];
#endif
            const int dz = tap / (KS * KS), dy = (tap / KS) % KS, dx = tap % KS;
            const int tapoff = (dz * RZ + dy * RY + dx * VS) * 16;
            if constexpr (PREC == 0) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    f32x4 a[MT];
#pragma unroll
                    for (int t = 0; t < MT; ++t)
                        a[t] = *reinterpret_cast<const f32x4*>(lds + arow[t] + tapoff + kk * 32);
                    const f32x4 b = __builtin_bit_cast(f32x4, bcur[kk]);
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int t = 0; t < MT; ++t)
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][s], b[s], acc[t], 0, 0, 0);
                }
            } else {
                const h8 bhi = __builtin_bit_cast(h8, bcur[0]);
                const h8 blo = __builtin_bit_cast(h8, bcur[1]);
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const h8 ahi = *reinterpret_cast<const h8*>(lds + arow[t] + tapoff);
                    const h8 alo = *reinterpret_cast<const h8*>(lds + arow[t] + tapoff + 32);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi, acc[t], 0, 0, 0);
                }
            }
            if (tap + 1 < NT) {
                bcur[0] = bnxt[0];
                bcur[1] = bnxt[1];
            }
        }
    }

    if (!wave_active) return;

    // ---------------------------------------------------------- epilogue
    // C/D map of 32x32 MFMA: col = lane&31 (cout), row = (reg&3) + 8*(reg>>2) + 4*half
    const bool cvalid = cout < p.Cout;
    const size_t DHW = (size_t)p.D * p.H * p.W;
    // PREC 1: undo the operand scaling (exact: a power of two per cout)
    const float oscale = (PREC == 1 && cvalid) ? p.wscale[cout] : 1.0f;
    if (p.ksplit > 1) {
        // split-K: raw partial sums to this split's slab; bias / residual / statistics
        // are applied by the reduce kernel once all splits are in
        float* slab = p.partial + ((size_t)blockIdx.z * p.N + n) * DHW * p.Cout;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = (reg & 3) + 8 * (reg >> 2) + 4 * half;
                const int m = (wm * MT + t) * 32 + row;
                const int tx = m & (TX - 1), ty = (m >> TXL) & (TY - 1), tz = m >> (TXL + TYL);
                const int z = z0 + tz, y = y0 + ty, x = x0 + tx;
                if (cvalid && z < p.D && y < p.H && x < p.W)
                    slab[(((size_t)z * p.H + y) * p.W + x) * p.Cout + cout] =
                        PREC == 1 ? acc[t][reg] * oscale : acc[t][reg];
            }
        }
        return;
    }
    const float bias = cvalid ? p.bias[(size_t)n * p.bias_stride_n + cout] : 0.0f;
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * half;
            const int m = (wm * MT + t) * 32 + row;
            const int tx = m & (TX - 1), ty = (m >> TXL) & (TY - 1), tz = m >> (TXL + TYL);
            const int z = z0 + tz, y = y0 + ty, x = x0 + tx;
            const bool ok = cvalid && z < p.D && y < p.H && x < p.W;
            if (ok) {
                float val = (PREC == 1 ? acc[t][reg] * oscale : acc[t][reg]) + bias;
                const size_t vox = ((size_t)z * p.H + y) * p.W + x;
                if (p.res_mode != DDPM3D_RES_NONE) val += ddpm3d_residual(p, n, z, y, x, cout);
                if (p.out_layout == DDPM3D_OUT_NDHWC)
                    p.out[((size_t)n * DHW + vox) * p.Cout + cout] = val;
                else
                    p.out[((size_t)n * p.Cout + cout) * DHW + vox] = val;
                s1 += val;
                s2 = fmaf(val, val, s2);
            }
        }
    }
    if (p.stats != nullptr) {
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (half == 0 && cvalid) {
            const int tile_in_n = (tz_i * p.tilesY + ty_i) * p.tilesX + tx_i;
            const size_t row = (size_t)n * p.stats_rows + (size_t)tile_in_n * WM + wm;
            float2 v2 = make_float2(s1, s2);
            *reinterpret_cast<float2*>(p.stats + (row * p.Cout + cout) * 2) = v2;
        }
    }
}

// ------------------------------------------------------- split-K reduction
// out = sum_s slab[s] + bias + residual, plus the GroupNorm partial sums the
// conv epilogue would have written.  One workgroup per DDPM3D_REDUCE_VOX
// voxels of one sample (= one statistics row); threads run along Cout, so every
// slab / residual / output access is a contiguous row segment.
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const ConvK p) {
    const size_t DHW = (size_t)p.D * p.H * p.W;
    const int rows = p.stats_rows;
    const int n = blockIdx.x / rows, r = blockIdx.x % rows;
    const size_t v0 = (size_t)r * DDPM3D_REDUCE_VOX;
    const size_t v1 = min(v0 + (size_t)DDPM3D_REDUCE_VOX, DHW);
    const size_t slab_stride = (size_t)p.N * DHW * p.Cout;
    for (int cout = threadIdx.x; cout < p.Cout; cout += 256) {
        const float bias = p.bias[(size_t)n * p.bias_stride_n + cout];
        float s1 = 0.0f, s2 = 0.0f;
        for (size_t v = v0; v < v1; ++v) {
            const size_t e = ((size_t)n * DHW + v) * p.Cout + cout;
            float val = p.partial[e];
            for (int s = 1; s < p.ksplit; ++s) val += p.partial[e + s * slab_stride];
            val += bias;
            if (p.res_mode != DDPM3D_RES_NONE) {
                const int x = (int)(v % p.W), y = (int)((v / p.W) % p.H), z = (int)(v / ((size_t)p.W * p.H));
                val += ddpm3d_residual(p, n, z, y, x, cout);
            }
            if (p.out_layout == DDPM3D_OUT_NDHWC)
                p.out[e] = val;
            else
                p.out[((size_t)n * p.Cout + cout) * DHW + v] = val;
            s1 += val;
            s2 = fmaf(val, val, s2);
        }
        if (p.stats != nullptr) {
            float2 v2 = make_float2(s1, s2);
            *reinterpret_cast<float2*>(p.stats + (((size_t)n * rows + r) * p.Cout + cout) * 2) = v2;
        }
    }
}

hipError_t ddpm3d_launch_splitk_reduce(const ConvK& k, hipStream_t st) {
    hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3(k.N * k.stats_rows), dim3(256), 0, st, k);
    return hipGetLastError();
}

// ---------------------------------------------------------------- dispatch
template <int PREC, int KS, int WN, int TXL, int TYL>
static hipError_t launch_cfg(const ConvK& k, int grid_x, int grid_y, hipStream_t st) {
    constexpr int TX = 1 << TXL, TY = 1 << TYL, TZ = 128 / (TX * TY);
    constexpr int PAD = KS / 2;
    constexpr int HX = TX + 2 * PAD, HY = TY + 2 * PAD, HZ = TZ + 2 * PAD;
    constexpr size_t lds_bytes = (size_t)HZ * LdsGeom<TX, HX, HY>::RZ * 16;
    hipLaunchKernelGGL((conv3d_kernel<PREC, KS, WN, TXL, TYL>), dim3(grid_x, grid_y, k.ksplit), dim3(256),
                       lds_bytes, st, k);
    return hipGetLastError();
}

hipError_t ddpm3d_launch_conv(const ConvK& k, const ConvCfg& c, hipStream_t st) {
    const int gx = k.N * k.tilesZ * k.tilesY * k.tilesX;
    const int gy = (k.CoutPad + 32 * c.WN - 1) / (32 * c.WN);
#define CASE(P_, KS_, WN_, TXL_, TYL_)                                                   \
    if (c.PREC == P_ && c.KS == KS_ && c.WN == WN_ && c.TXL == TXL_ && c.TYL == TYL_)   \
        return launch_cfg<P_, KS_, WN_, TXL_, TYL_>(k, gx, gy, st);
    CASE(0, 3, 4, 3, 3) CASE(0, 3, 2, 3, 3) CASE(0, 3, 1, 3, 3)
    CASE(0, 3, 4, 2, 2) CASE(0, 3, 2, 2, 2) CASE(0, 3, 1, 2, 2)
    CASE(0, 1, 4, 3, 3) CASE(0, 1, 2, 3, 3) CASE(0, 1, 1, 3, 3)
    CASE(0, 1, 4, 2, 2) CASE(0, 1, 2, 2, 2) CASE(0, 1, 1, 2, 2)
    CASE(1, 3, 4, 3, 3) CASE(1, 3, 2, 3, 3) CASE(1, 3, 1, 3, 3)
    CASE(1, 3, 4, 2, 2) CASE(1, 3, 2, 2, 2) CASE(1, 3, 1, 2, 2)
    CASE(1, 1, 4, 3, 3) CASE(1, 1, 2, 3, 3) CASE(1, 1, 1, 3, 3)
    CASE(1, 1, 4, 2, 2) CASE(1, 1, 2, 2, 2) CASE(1, 1, 1, 2, 2)
#undef CASE
    return hipErrorInvalidValue;
}
