// Pieces shared by every conv kernel: the operand vector types, the LDS geometry of the halo
// image and the epilogue (bias / residual / store / GroupNorm partial sums, or the raw split-K
// slab).
#pragma once
#include "conv3d_load.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

// one 32x32x16 MFMA on 16-bit operands: f16, or (BF = true) bf16 bit patterns in the same registers
template <bool BF>
__device__ __forceinline__ f32x16 mfma16(h8 a, h8 b, f32x16 c) {
    if constexpr (BF)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// smallest x >= v with x % 16 == res
constexpr int pad_to_residue(int v, int res) { return v + ((res - v % 16) + 16) % 16; }

// LDS strides (in 16-byte slots) of the halo image; see the bank-conflict note in conv3d.hip.
template <int TX, int HX, int HY>
struct LdsGeom {
    static constexpr int RY = pad_to_residue(HX * 5, TX == 8 ? 8 : 4);
    static constexpr int RZ = TX == 8 ? HY * RY : pad_to_residue(HY * RY, 0);
};


// ---- 4x4 transpose of four registers across the four lanes of a quad (lanes 4q .. 4q+3):
// new a_r(lane l) = old a_l(lane r).  Two butterfly stages of DPP quad permutes; b0 / b1 = bit 0 / 1
// of the lane id.  Used by the epilogue: a lane's four consecutive accumulator registers are four
// consecutive x voxels of ONE cout; after the transpose the lane holds four consecutive COUTS of one
// voxel, i.e. one 16-byte store instead of four 4-byte ones.
__device__ __forceinline__ float quad_xchg1(float v) {   // lane ^ 1
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_xchg2(float v) {   // lane ^ 2
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ void quad_transpose(float (&a)[4], bool b0, bool b1) {
#pragma unroll
    for (int k = 0; k < 4; k += 2) {          // pairs (0,1), (2,3) across lane bit 0
        const float y = quad_xchg1(b0 ? a[k] : a[k + 1]);
        a[k] = b0 ? y : a[k];
        a[k + 1] = b0 ? a[k + 1] : y;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {             // pairs (0,2), (1,3) across lane bit 1
        const float y = quad_xchg2(b1 ? a[k] : a[k + 2]);
        a[k] = b1 ? y : a[k];
        a[k + 2] = b1 ? a[k + 2] : y;
    }
}

// Epilogue shared by the conv kernels: bias / residual / store / GroupNorm partial sums, or
// the raw split-K slab.  C/D map of a 32x32 MFMA: col = lane&31 (cout),
// row = (reg&3) + 8*(reg>>2) + 4*half.
// inv_act = 1 / activation scale of this sample (split-f16 modes; 1 in the exact mode)
// WIDE: the 16-byte store form below (one-MFMA modes), or the four-byte form
// ZPAIRS (conv3d_wz.h on tiles of several z-pairs, RP = TX * TY positions each): accumulator u = zbit*2 + t holds, in
// row r, output plane zbit of z-pair (t*32 + r) / RP -- depth 2 * pair + zbit instead of the plain m / RP
template <int TXL, int TYL, bool ZPAIRS>
__device__ __forceinline__ constexpr int epi_tz(int m) {
    const int u = m >> 5, r = m & 31;
    return ZPAIRS ? 2 * ((((u & 1) << 5) + r) >> (TXL + TYL)) + (u >> 1) : m >> (TXL + TYL);
}

// KSPLIT: the caller knows this is a split launch (p.ksplit > 1): only the slab code is instantiated.
// Returns true when every slab store of the lane's tile was a WRITE-THROUGH (sc1) store -- the 16-byte form on a
// full tile -- so that splitk_finish (below) may signal without an agent-scope release.
template <int PREC, int WM, int MT, int TXL, int TYL, bool WIDE = false, bool ZPAIRS = false, bool KSPLIT = false>
// pre_ws / pre_bias (pre = true): p.wscale[cout] and the lane's bias, loaded by the caller at kernel
// START (conv3d_wz.h): at the epilogue's start they are a dependent global load -- ~2k cycles at the
// head of a phase in which the wave issues no MFMA
__device__ __forceinline__ bool conv_epilogue(const ConvK& p, const f32x16 (&acc)[MT], int n, int z0, int y0,
                                              int x0, int tile_in_n, int wm, int cout, int half, int ksplit_idx,
                                              float inv_act, bool pre = false, float pre_ws = 1.0f,
                                              float pre_bias = 0.0f) {
    constexpr int TX = 1 << TXL, TY = 1 << TYL;
    const bool cvalid = cout < p.Cout;
    const size_t DHW = (size_t)p.D * p.H * p.W;
    // PREC 1/2: undo the operand scaling (exact: a power of two per cout)
    const float oscale = (PREC != 0 && cvalid) ? (pre ? pre_ws : p.wscale[cout]) * inv_act : 1.0f;

    // Fast path -- every launch of the network except ragged edge tiles, the NCDHW output conv
    // and the up/down-sampling skip sums: the tile lies inside the volume, so element (t, reg)
    // of a lane sits at  lane base (cout, half) + a WAVE-UNIFORM offset.  Stores (and the
    // same-shaped residual loads) are buffer instructions with that offset in an SGPR: no
    // per-element index arithmetic, bounds test or 64-bit address.  The general path below
    // costs ~200 instructions per element, which for a 128-cin layer was a third of the
    // kernel.  Same values in the same order, so both paths agree bit for bit.
    {
        constexpr int TZ = WM * MT * 32 / (TX * TY);
        const bool split = KSPLIT || p.ksplit > 1;
        const bool full = z0 + TZ <= p.D && y0 + TY <= p.H && x0 + TX <= p.W;
        // residual of a split conv is the reduce kernel's business
        const int rm = split ? DDPM3D_RES_NONE : p.res_mode;
        // the residual tensor of the up / down ResBlocks is the block input at the other
        // resolution (H/2 x W/2 or 2H x 2W); its sample must be 32-bit addressable too
        const size_t Hr = rm == DDPM3D_RES_UP ? p.H / 2 : (rm == DDPM3D_RES_POOL ? (size_t)p.H * 2 : p.H);
        const size_t Wr = rm == DDPM3D_RES_UP ? p.W / 2 : (rm == DDPM3D_RES_POOL ? (size_t)p.W * 2 : p.W);
        const size_t samp_r = (size_t)p.D * Hr * Wr * p.Cout;
        // bf16 tensors (ddpm3d_conv_desc.io_dtype): 2-byte elements, same offsets in elements
        const bool o16 = !split && (p.io & DDPM3D_IO_OUT_BF16), r16 = (p.io & DDPM3D_IO_RES_BF16) != 0;
        const bool f16 = (p.io & DDPM3D_IO_HALF_IS_F16) != 0;      // the 16-bit tensors hold IEEE f16, not bf16
        const unsigned eso = o16 ? 2u : 4u, esr = r16 ? 2u : 4u;
        if (full && p.out_layout == DDPM3D_OUT_NDHWC && samp_r * 4 < 0xFFFFFFF0ull) {
            const size_t samp = DHW * p.Cout;                       // elements per sample (< 2^30, C ABI guard)
            const unsigned cstride = (unsigned)p.Cout * eso;        // bytes per voxel of the output
            const unsigned rstride = (unsigned)p.Cout * esr;        // ... of the residual
            float* dst = split ? p.partial + ((size_t)ksplit_idx * p.N + n) * samp
                               : reinterpret_cast<float*>(reinterpret_cast<char*>(p.out) + (size_t)n * samp * eso);
            const __amdgpu_buffer_rsrc_t drsrc = make_rsrc(dst, (unsigned)(samp * eso));
            const bool resid = rm != DDPM3D_RES_NONE;
            const __amdgpu_buffer_rsrc_t rrsrc =
                resid ? make_rsrc(reinterpret_cast<const char*>(p.res) + (size_t)n * samp_r * esr, (unsigned)(samp_r * esr))
                      : drsrc;
            // the lane's half adds 4 to the MFMA row: 4 voxels in x (8-wide tile) or one row in y (4-wide)
            const unsigned hx = (4 * half) & (TX - 1), hy = ((4 * half) >> TXL) & (TY - 1);
            const unsigned vbase = (((unsigned)z0 * p.H + y0 + hy) * p.W + x0 + hx) * cstride + (unsigned)cout * eso;
            const unsigned voff = cvalid ? vbase : DDPM3D_OOB_OFFSET;   // out-of-range lanes: loads 0, stores dropped
            // Residual addressing, also "lane base + wave-uniform offset" (tiles start at even y0, x0):
            //   SAME  x[z][y][x]                      : the output's own offsets
            //   UP    x[z][y>>1][x>>1] (unet.py:241)  : (t0 + h) >> 1 = (t0 >> 1) + (h >> 1) -- the half's
            //         4 voxels in x are 2 source voxels; in a 4-wide tile its one row in y shares the source row
            //   POOL  mean of x[z][2y+{0,1}][2x+{0,1}]: doubled offsets, four loads
            const unsigned rW = (unsigned)Wr, rH = (unsigned)Hr;
            unsigned rbase = (((unsigned)z0 * p.H + y0 + hy) * p.W + x0 + hx) * rstride + (unsigned)cout * esr;
            if (rm == DDPM3D_RES_UP)
                rbase = (((unsigned)z0 * rH + (y0 >> 1)) * rW + (x0 >> 1) + (hx >> 1)) * rstride + (unsigned)cout * esr;
            else if (rm == DDPM3D_RES_POOL)
                rbase = (((unsigned)z0 * rH + 2 * (y0 + hy)) * rW + 2 * (x0 + hx)) * rstride + (unsigned)cout * esr;
            unsigned roff_ = 0;
            // one element of the residual at lane base + scalar offset
            auto rload = [&](const unsigned so) {
                if (r16) return ddpm3d_half_to_float((unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rrsrc, roff_, so, 0), f16);
                return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrsrc, roff_, so, 0));
            };
            // ---- WIDE form (r03): residual none / same and Cout % 4 == 0, in the ONE-MFMA modes (f16, bf16).
            // Same box, whole forward (profiles/r03_lib_ab_epilogue_forms_{f16x3,bf16,f16,f32}.txt): -3...4 % there; +0.4 % in the
            // f16x3 mode and +1.4 % in the exact mode, whose epilogues hide behind three and sixteen times the
            // MFMA work and only pay for the transposes -- those keep the four-byte form.
            // Phase stamps of the dominant kernel (profiles/r03_wz_stamps_*.txt) put its epilogue at 8 % of a
            // wave's life without a residual and 11.5 % with one: 64 four-byte stores (+ 64 four-byte loads) per
            // lane, bound by the CU's store ISSUE rate (~7 B/clk/CU), not by bandwidth.  A lane's registers
            // 4g .. 4g+3 are four consecutive x voxels of its cout; transposed across the lane quad it holds
            // four consecutive couts of ONE voxel: 16 stores of 16 bytes (one wave instruction = 8 voxels x
            // 128 contiguous bytes), the residual read the same way.  The statistics are taken in the
            // transposed layout (after the residual add) and folded over the quad and the two halves.
            if (WIDE && (rm == DDPM3D_RES_NONE || rm == DDPM3D_RES_SAME) && (p.Cout & 3) == 0) {
                const int li = threadIdx.x & 3;
                const bool b0 = (li & 1) != 0, b1 = (li & 2) != 0;
                const unsigned lane_vox = (((unsigned)z0 * p.H + y0 + hy) * p.W + x0 + hx + li);
                const unsigned cq = (unsigned)(cout & ~3);
                const unsigned wv = cvalid ? lane_vox * cstride + cq * eso : DDPM3D_OOB_OFFSET;
                const unsigned wr = cvalid ? lane_vox * rstride + cq * esr : DDPM3D_OOB_OFFSET;
                const float bias_w = (!split && cvalid) ? (pre ? pre_bias : p.bias[(size_t)n * p.bias_stride_n + cout]) : 0.0f;
                GnAcc ga[4];
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    f32x4 rq[4];
                    if (resid) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int m0 = (wm * MT + t) * 32 + 8 * g;
                            const int ty = (m0 >> TXL) & (TY - 1), tz = epi_tz<TXL, TYL, ZPAIRS>(m0), tx = m0 & (TX - 1);
                            const unsigned so = (unsigned)((tz * p.H + ty) * p.W + tx) * rstride;
                            if (r16) rq[g] = half4_expand(__builtin_amdgcn_raw_buffer_load_b64(rrsrc, wr, so, 0), f16);
                            else rq[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, wr, so, 0));
                        }
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float a[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float v = PREC != 0 ? acc[t][4 * g + i] * oscale : acc[t][4 * g + i];
                            a[i] = split ? v : v + bias_w;
                        }
                        quad_transpose(a, b0, b1);
                        if (!split) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                if (resid) a[i] += rq[g][i];
                                if (p.stats != nullptr) {
                                    if (t == 0 && g == 0) ga[i].init(a[i]);
                                    ga[i].add(a[i]);
                                }
                            }
                        }
                        const int m0 = (wm * MT + t) * 32 + 8 * g;
                        const int ty = (m0 >> TXL) & (TY - 1), tz = epi_tz<TXL, TYL, ZPAIRS>(m0), tx = m0 & (TX - 1);
                        const unsigned so = (unsigned)((tz * p.H + ty) * p.W + tx) * cstride;
                        // raw slab: WRITE-THROUGH (aux 16 = sc1), the hand-off form of splitk_finish -- with the tile offset
                        // in the VGPR and soffset 0.  Measured (r04, scratch/dbg_fused*.py -> profiles/r04_sc1_store_hazard.txt):
                        // `buffer_store_dwordx4 ... sc1` with an SGPR soffset stores, now and then, the NEXT group's value in its
                        // first data register (one wrong float per 16-cout run, in late workgroups of a launch, ~1e-4 of the
                        // elements; never with plain stores, never with four-byte stores): hipcc inserts the wait state between a
                        // > 8-byte store and the VALU that overwrites its data registers only when soffset is NOT a register
                        // (the documented gfx9 exemption), and the exemption does not hold for the slower sc1 store.
                        if (split)
                            __builtin_amdgcn_raw_buffer_store_b128(u32x4{__builtin_bit_cast(unsigned, a[0]), __builtin_bit_cast(unsigned, a[1]),
                                                                         __builtin_bit_cast(unsigned, a[2]), __builtin_bit_cast(unsigned, a[3])},
                                                                   drsrc, wv + so, 0, 16);
                        else if (KSPLIT) { }
                        else if (o16)
                            __builtin_amdgcn_raw_buffer_store_b64(u32x2{half_pack(a[0], a[1], f16), half_pack(a[2], a[3], f16)}, drsrc, wv, so, 0);
                        else
                            __builtin_amdgcn_raw_buffer_store_b128(u32x4{__builtin_bit_cast(unsigned, a[0]), __builtin_bit_cast(unsigned, a[1]),
                                                                         __builtin_bit_cast(unsigned, a[2]), __builtin_bit_cast(unsigned, a[3])},
                                                                   drsrc, wv, so, 0);
                    }
                }
                if (!split && p.stats != nullptr) {
                    // fold the quad's four voxel lanes and the two halves; then lane li keeps cout cq + li = its own
                    double d1[4], d2[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        d1[i] = ga[i].sum1((float)(MT * 4));
                        d2[i] = ga[i].sum2((float)(MT * 4));
                        d1[i] += __shfl_xor(d1[i], 1); d2[i] += __shfl_xor(d2[i], 1);
                        d1[i] += __shfl_xor(d1[i], 2); d2[i] += __shfl_xor(d2[i], 2);
                        d1[i] += __shfl_xor(d1[i], 32); d2[i] += __shfl_xor(d2[i], 32);
                    }
                    const double s1 = b1 ? (b0 ? d1[3] : d1[2]) : (b0 ? d1[1] : d1[0]);
                    const double s2 = b1 ? (b0 ? d2[3] : d2[2]) : (b0 ? d2[1] : d2[0]);
                    if (half == 0 && cvalid) {
                        const size_t row = (size_t)tile_in_n * WM + wm;
                        *reinterpret_cast<double2*>(p.stats + (((size_t)n * p.Cout + cout) * p.stats_rows + row) * 2) =
                            make_double2(s1, s2);
                    }
                }
                return split;
            }
            roff_ = cvalid ? rbase : DDPM3D_OOB_OFFSET;
            const float bias = (!split && cvalid) ? (pre ? pre_bias : p.bias[(size_t)n * p.bias_stride_n + cout]) : 0.0f;
            GnAcc gs;   // GroupNorm partial sums (conv3d_params.h)
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                unsigned soff[16];
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int m0 = (wm * MT + t) * 32 + (reg & 3) + 8 * (reg >> 2);
                    const int tx = m0 & (TX - 1), ty = (m0 >> TXL) & (TY - 1), tz = epi_tz<TXL, TYL, ZPAIRS>(m0);
                    soff[reg] = (unsigned)((tz * p.H + ty) * p.W + tx) * cstride;
                }
                float r[16];
                if (rm == DDPM3D_RES_SAME) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int m0 = (wm * MT + t) * 32 + (reg & 3) + 8 * (reg >> 2);
                        const int tx = m0 & (TX - 1), ty = (m0 >> TXL) & (TY - 1), tz = epi_tz<TXL, TYL, ZPAIRS>(m0);
                        r[reg] = rload((unsigned)((tz * p.H + ty) * p.W + tx) * rstride);
                    }
                } else if (rm == DDPM3D_RES_UP) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int m0 = (wm * MT + t) * 32 + (reg & 3) + 8 * (reg >> 2);
                        const int tx = m0 & (TX - 1), ty = (m0 >> TXL) & (TY - 1), tz = epi_tz<TXL, TYL, ZPAIRS>(m0);
                        r[reg] = rload((unsigned)((tz * rH + (ty >> 1)) * rW + (tx >> 1)) * rstride);
                    }
                } else if (rm == DDPM3D_RES_POOL) {
                    // AvgPool3d window order (h, w): ((r00 + r01) + r10) + r11, then * 1/4 (ddpm3d_residual)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int m0 = (wm * MT + t) * 32 + (reg & 3) + 8 * (reg >> 2);
                        const int tx = m0 & (TX - 1), ty = (m0 >> TXL) & (TY - 1), tz = epi_tz<TXL, TYL, ZPAIRS>(m0);
                        const unsigned so = (unsigned)((tz * rH + 2 * ty) * rW + 2 * tx) * rstride;
                        const float r00 = rload(so), r01 = rload(so + rstride);
                        const float r10 = rload(so + rW * rstride), r11 = rload(so + (rW + 1) * rstride);
                        r[reg] = (((r00 + r01) + r10) + r11) * 0.25f;
                    }
                }
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    float val = PREC != 0 ? acc[t][reg] * oscale : acc[t][reg];
                    if (!split) {
                        val += bias;
                        if (resid) val += r[reg];
                        if (t == 0 && reg == 0) gs.init(val);
                        gs.add(val);
                    }
                    if (o16)
                        __builtin_amdgcn_raw_buffer_store_b16(ddpm3d_to_half(val, f16), drsrc, voff, soff[reg], 0);
                    else
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), drsrc, voff, soff[reg], 0);
                }
            }
            if (!split && p.stats != nullptr) {
                double s1 = gs.sum1((float)(MT * 16)), s2 = gs.sum2((float)(MT * 16));
                s1 += __shfl_xor(s1, 32);
                s2 += __shfl_xor(s2, 32);
                if (half == 0 && cvalid) {
                    const size_t row = (size_t)tile_in_n * WM + wm;
                    *reinterpret_cast<double2*>(p.stats + (((size_t)n * p.Cout + cout) * p.stats_rows + row) * 2) =
                        make_double2(s1, s2);
                }
            }
            return false;
        }
    }
    if (KSPLIT || p.ksplit > 1) {
        // split-K: raw partial sums to this split's slab; bias / residual / statistics
        // are applied by the reduce kernel once all splits are in
        float* slab = p.partial + ((size_t)ksplit_idx * p.N + n) * DHW * p.Cout;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = (reg & 3) + 8 * (reg >> 2) + 4 * half;
                const int m = (wm * MT + t) * 32 + row;
                const int tx = m & (TX - 1), ty = (m >> TXL) & (TY - 1), tz = epi_tz<TXL, TYL, ZPAIRS>(m);
                const int z = z0 + tz, y = y0 + ty, x = x0 + tx;
                if (cvalid && z < p.D && y < p.H && x < p.W)
                    slab[(((size_t)z * p.H + y) * p.W + x) * p.Cout + cout] =
                        PREC != 0 ? acc[t][reg] * oscale : acc[t][reg];
            }
        }
        return false;
    }
    if constexpr (KSPLIT) return false;
    const float bias = cvalid ? (pre ? pre_bias : p.bias[(size_t)n * p.bias_stride_n + cout]) : 0.0f;
    GnAcc gs;
    gs.init(0.0f);
    float cnt = 0.0f;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * half;
            const int m = (wm * MT + t) * 32 + row;
            const int tx = m & (TX - 1), ty = (m >> TXL) & (TY - 1), tz = epi_tz<TXL, TYL, ZPAIRS>(m);
            const int z = z0 + tz, y = y0 + ty, x = x0 + tx;
            const bool ok = cvalid && z < p.D && y < p.H && x < p.W;
            if (ok) {
                float val = (PREC != 0 ? acc[t][reg] * oscale : acc[t][reg]) + bias;
                const size_t vox = ((size_t)z * p.H + y) * p.W + x;
                if (p.res_mode != DDPM3D_RES_NONE) val += ddpm3d_residual(p, n, z, y, x, cout);
                if (p.out_layout == DDPM3D_OUT_NDHWC)
                    ddpm3d_act_store(p.out, ((size_t)n * DHW + vox) * p.Cout + cout, val, (p.io & DDPM3D_IO_OUT_BF16) != 0,
                                     (p.io & DDPM3D_IO_HALF_IS_F16) != 0);
                else
                    p.out[((size_t)n * p.Cout + cout) * DHW + vox] = val;
                if (cnt == 0.0f) gs.init(val);      // pivot = the lane's first valid value
                gs.add(val);
                cnt += 1.0f;
            }
        }
    }
    if (p.stats != nullptr) {
        double s1 = gs.sum1(cnt), s2 = gs.sum2(cnt);
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (half == 0 && cvalid) {
            // channel-major [N][Cout][rows][2]: a GroupNorm group's partial sums are contiguous
            const size_t row = (size_t)tile_in_n * WM + wm;
            *reinterpret_cast<double2*>(p.stats + (((size_t)n * p.Cout + cout) * p.stats_rows + row) * 2) =
                make_double2(s1, s2);
        }
    }
    return false;
}


// ---- split-K: combine INSIDE the launch (r04) --------------------------------------------------------------
// The S workgroups of one (sample, tile, 128-cout block) each leave a raw slab (conv_epilogue above), then draw a
// ticket; the one that draws S - 1 -- the last to arrive, whichever split it computed -- sums the S slabs of its
// tile IN SLAB ORDER, adds bias and residual, stores the output and the tile's GroupNorm partial sums: what
// conv_splitk_reduce_v4_kernel did in a launch of its own (same additions in the same order: the output is
// bit-identical to the two-launch form's; the statistics are grouped per TILE instead of per reduce row, as an
// unsplit launch groups them).  No workgroup waits for another: nothing can hang.
//
// Visibility (MI355X_MICROARCH.md, inter-workgroup visibility / cdna_hip_programming.md, in-launch split-K):
//   producer  slab stores write-through (sc1, `wt`) or plain + agent-scope release; every wave drains its stores
//             (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane: [release,] relaxed agent-scope fetch_add;
//   consumer  the wave whose add came last: agent-scope acquire (invalidates this CU's L1), s_waitcnt vmcnt(0),
//             workgroup barrier, then plain loads by every wave.
// The tickets (ConvK::tickets, one word per (n, tile, cout block)) are zero before the launch and the last
// arriver puts its word back to zero: a later launch on the same workspace (stream order) finds them clean.
// lds: the kernel's dynamic LDS (>= 8208 bytes, dead by now: the first barrier below is also its last reader's).
template <int TXL, int TYL>
__device__ __forceinline__ void splitk_finish(const ConvK& p, unsigned char* lds, bool wt, int n, int z0, int y0,
                                              int x0, int tile_in_n, int cy) {
    constexpr int TX = 1 << TXL, TY = 1 << TYL;
    const int tid = threadIdx.x;
    unsigned* flag = reinterpret_cast<unsigned*>(lds);
    const unsigned group = ((unsigned)n * (unsigned)(p.tilesZ * p.tilesY * p.tilesX) + (unsigned)tile_in_n) *
                               (unsigned)(p.CoutPad / 128) + (unsigned)cy;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        if (!wt) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *flag = __hip_atomic_fetch_add(p.tickets + group, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (*flag != (unsigned)p.ksplit - 1u) return;
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(p.tickets + group, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();

    // thread = (cout quad q of the block's 32, voxel lane vl of 8): a wave reads two voxels x 512 contiguous bytes
    const int q = tid & 31, vl = tid >> 5;
    const int c4 = cy * 128 + 4 * q;
    const size_t DHW = (size_t)p.D * p.H * p.W;
    const size_t slab_stride = (size_t)p.N * DHW * p.Cout;
    const bool b16 = (p.io & DDPM3D_IO_OUT_BF16) != 0, f16 = (p.io & DDPM3D_IO_HALF_IS_F16) != 0;
    const f32x4 bias = *reinterpret_cast<const f32x4*>(p.bias + (size_t)n * p.bias_stride_n + c4);
    GnAcc gs[4];
    float cnt = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) gs[c].init(0.0f);
#pragma unroll 2
    for (int k = 0; k < 16; ++k) {
        const int i = vl + 8 * k;
        const int z = z0 + (i >> (TXL + TYL)), y = y0 + ((i >> TXL) & (TY - 1)), x = x0 + (i & (TX - 1));
        if (z < p.D && y < p.H && x < p.W) {
            const size_t e = ((size_t)n * DHW + ((size_t)z * p.H + y) * p.W + x) * p.Cout + c4;
            f32x4 val = *reinterpret_cast<const f32x4*>(p.partial + e);
            int s = 1;
            for (; s + 3 < p.ksplit; s += 4) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(p.partial + e + (size_t)s * slab_stride);
                const f32x4 b = *reinterpret_cast<const f32x4*>(p.partial + e + (size_t)(s + 1) * slab_stride);
                const f32x4 c = *reinterpret_cast<const f32x4*>(p.partial + e + (size_t)(s + 2) * slab_stride);
                const f32x4 d = *reinterpret_cast<const f32x4*>(p.partial + e + (size_t)(s + 3) * slab_stride);
                val += a; val += b; val += c; val += d;
            }
            for (; s < p.ksplit; ++s) val += *reinterpret_cast<const f32x4*>(p.partial + e + (size_t)s * slab_stride);
            val += bias;
            if (p.res_mode != DDPM3D_RES_NONE) {
#pragma unroll
                for (int c = 0; c < 4; ++c) val[c] += ddpm3d_residual(p, n, z, y, x, c4 + c);
            }
            if (b16)
                *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(p.out) + e) =
                    u32x2{half_pack(val[0], val[1], f16), half_pack(val[2], val[3], f16)};
            else
                *reinterpret_cast<f32x4*>(p.out + e) = val;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (cnt == 0.0f) gs[c].init(val[c]);
                gs[c].add(val[c]);
            }
            cnt += 1.0f;
        }
    }
    if (p.stats != nullptr) {
        // one row per tile (row = tile_in_n, as an unsplit launch of this tile grid writes them).  Fold in a fixed
        // order: the two voxel lanes of a wave (lane, lane ^ 32), then the four waves through LDS.
        double* red = reinterpret_cast<double*>(lds + 16);      // [4 waves][128 couts][2]
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            double d1 = gs[c].sum1(cnt), d2 = gs[c].sum2(cnt);
            const double o1 = __shfl_xor(d1, 32), o2 = __shfl_xor(d2, 32);
            if (lane < 32) {
                red[((wave * 128) + 4 * q + c) * 2] = d1 + o1;
                red[((wave * 128) + 4 * q + c) * 2 + 1] = d2 + o2;
            }
        }
        __syncthreads();
        if (tid < 128) {
            const double a = (red[tid * 2] + red[(128 + tid) * 2]) + (red[(256 + tid) * 2] + red[(384 + tid) * 2]);
            const double b = (red[tid * 2 + 1] + red[(128 + tid) * 2 + 1]) + (red[(256 + tid) * 2 + 1] + red[(384 + tid) * 2 + 1]);
            *reinterpret_cast<double2*>(p.stats + (((size_t)n * p.Cout + cy * 128 + tid) * p.stats_rows + tile_in_n) * 2) =
                make_double2(a, b);
        }
    }
}
