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

// The 8- and 16-byte buffer stores of the epilogues.  gfx950 reads the FIRST data register of such a store late for the
// last four lanes of a 16-lane row: a VALU write to it shortly behind the store can overtake the read, and the lane
// stores the NEXT group's value (one element per 16-cout run, different ones from run to run, only under load).  The
// compiler places no wait state here -- for 16 bytes because the store's soffset is an SGPR (the documented gfx9
// exemption), for 8 bytes never -- so how close the overwrite comes is the scheduler's choice: r03's 16-byte epilogue
// never showed it, its write-through (sc1) form did (profiles/r04_sc1_store_hazard.txt), and the lean epilogue below
// showed it with plain stores in three layers of the bf16 network (profiles/r04_store_data_hazard_plain.txt).
// The asm pins the data registers for eight wait states behind the store ("+v": no value can be allocated to them
// before it; "memory": it stays behind the store).
#ifndef DDPM3D_EPI_PIN
#define DDPM3D_EPI_PIN "s_nop 7"     // measurement builds: "s_nop 0", "" (pin without wait states), ...
#endif
__device__ __forceinline__ void epi_store_b128(u32x4 d, __amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, voff, soff, 0);
    asm volatile(DDPM3D_EPI_PIN : "+v"(d) : : "memory");
}
__device__ __forceinline__ void epi_store_b64(u32x2 d, __amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b64(d, rsrc, voff, soff, 0);
    asm volatile(DDPM3D_EPI_PIN : "+v"(d) : : "memory");
}

// ---- The lean 16-byte epilogue (r04).  conv_epilogue below decides everything at run time and computes, before its
// first branch, what every one of its paths needs (sample sizes and residual bases of the up / down forms, both store
// forms' lane offsets, 64-bit products): ~500 scalar instructions that a wave issues one by one, with half of its
// ~100 wave-uniform values parked in VGPR lanes (v_readlane / v_writelane).  Phase stamps put that epilogue at
// 15.7 k cycles for the 16 stores of a 1x1 workgroup -- half its life on the short layers -- and at 9.5-14 k in the
// Winograd kernels (profiles/r04_pw_stamps_*.txt, r04_wz_stamps_*.txt).  This function is the same arithmetic in the
// same order (bit-identical results) for the case nearly every launch of the network is: tile inside the volume,
// NDHWC output (or a split launch's slab), no residual or a same-shape one, Cout % 4 == 0.  It returns false without
// side effects when the launch is anything else; the caller then runs conv_epilogue.
//   acc   NJ cout blocks (couts cout0 + 32 j) x MT row tiles; ws / bs: the blocks' output scale and bias
//   every residual load of the wave is issued before its first store (in-place residuals: out == res is allowed)
template <int PREC, int WM, int MT, int NJ, int TXL, int TYL, bool ZPAIRS, bool KSPLIT>
__device__ __forceinline__ bool conv_epilogue_lean(const ConvK& p, const f32x16 (&acc)[NJ * MT], int n, int z0, int y0, int x0,
                                                   int tile_in_n, int wm, int cout0, int half, int ksplit_idx, float inv_act,
                                                   const float (&ws)[NJ], const float (&bs)[NJ]) {
    constexpr int TX = 1 << TXL, TY = 1 << TYL, TZ = WM * MT * 32 / (TX * TY);
    const bool split = KSPLIT || p.ksplit > 1;
    const int rm = split ? DDPM3D_RES_NONE : p.res_mode;
    const size_t samp = (size_t)p.D * p.H * p.W * p.Cout;              // elements per sample
    const bool ok = z0 + TZ <= p.D && y0 + TY <= p.H && x0 + TX <= p.W && p.out_layout == DDPM3D_OUT_NDHWC &&
                    (rm == DDPM3D_RES_NONE || rm == DDPM3D_RES_SAME) && (p.Cout & 3) == 0 && samp * 4 < 0xFFFFFFF0ull;
    if (!ok) return false;
    const bool resid = rm == DDPM3D_RES_SAME;
    const bool o16 = !split && (p.io & DDPM3D_IO_OUT_BF16), r16 = (p.io & DDPM3D_IO_RES_BF16) != 0;
    const bool f16 = (p.io & DDPM3D_IO_HALF_IS_F16) != 0;
    const unsigned eso = o16 ? 2u : 4u, esr = r16 ? 2u : 4u;
    const unsigned cstride = (unsigned)p.Cout * eso, rstride = (unsigned)p.Cout * esr;
    float* dst = split ? p.partial + ((size_t)ksplit_idx * p.N + n) * samp
                       : reinterpret_cast<float*>(reinterpret_cast<char*>(p.out) + (size_t)n * samp * eso);
    const __amdgpu_buffer_rsrc_t drsrc = make_rsrc(dst, (unsigned)(samp * eso));
    const __amdgpu_buffer_rsrc_t rrsrc =
        resid ? make_rsrc(reinterpret_cast<const char*>(p.res) + (size_t)n * samp * esr, (unsigned)(samp * esr)) : drsrc;
    const int li = threadIdx.x & 3;
    const bool b0 = (li & 1) != 0, b1 = (li & 2) != 0;
    const unsigned hx = (4 * half) & (TX - 1), hy = ((4 * half) >> TXL) & (TY - 1);
    const unsigned lane_vox = (((unsigned)z0 * p.H + y0 + hy) * p.W + x0 + hx + li);
    unsigned wv[NJ], wr[NJ];
    bool cv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int cout = cout0 + 32 * j;
        cv[j] = cout < p.Cout;
        const unsigned cq = (unsigned)(cout & ~3);
        wv[j] = cv[j] ? lane_vox * cstride + cq * eso : DDPM3D_OOB_OFFSET;
        wr[j] = cv[j] ? lane_vox * rstride + cq * esr : DDPM3D_OOB_OFFSET;
    }
    // wave-uniform voxel offset of (row tile t, register quad g), in voxels
    unsigned so[MT][4];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int m0 = (wm * MT + t) * 32 + 8 * g;
            const int ty = (m0 >> TXL) & (TY - 1), tz = epi_tz<TXL, TYL, ZPAIRS>(m0), tx = m0 & (TX - 1);
            so[t][g] = (unsigned)((tz * p.H + ty) * p.W + tx);
        }
    f32x4 rq[NJ][MT][4];
    if (resid) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (r16) rq[j][t][g] = half4_expand(__builtin_amdgcn_raw_buffer_load_b64(rrsrc, wr[j], so[t][g] * rstride, 0), f16);
                    else rq[j][t][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, wr[j], so[t][g] * rstride, 0));
                }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const float oscale = (PREC != 0 && cv[j]) ? ws[j] * inv_act : 1.0f;
        const float bias_w = (!split && cv[j]) ? bs[j] : 0.0f;
        GnAcc ga[4];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float a[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = PREC != 0 ? acc[j * MT + t][4 * g + i] * oscale : acc[j * MT + t][4 * g + i];
                    a[i] = split ? v : v + bias_w;
                }
                quad_transpose(a, b0, b1);
                if (!split) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (resid) a[i] += rq[j][t][g][i];
                        if (p.stats != nullptr) {
                            if (t == 0 && g == 0) ga[i].init(a[i]);
                            ga[i].add(a[i]);
                        }
                    }
                }
                const unsigned sb = so[t][g] * cstride;
                if (!split && o16)
                    epi_store_b64(u32x2{half_pack(a[0], a[1], f16), half_pack(a[2], a[3], f16)}, drsrc, wv[j], sb);
                else
                    epi_store_b128(u32x4{__builtin_bit_cast(unsigned, a[0]), __builtin_bit_cast(unsigned, a[1]),
                                         __builtin_bit_cast(unsigned, a[2]), __builtin_bit_cast(unsigned, a[3])}, drsrc, wv[j], sb);
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
            if (half == 0 && cv[j]) {
                const size_t row = (size_t)tile_in_n * WM + wm;
                *reinterpret_cast<double2*>(p.stats + (((size_t)n * p.Cout + cout0 + 32 * j) * p.stats_rows + row) * 2) =
                    make_double2(s1, s2);
            }
        }
    }
    return true;
}

// KSPLIT: the caller knows this is a split launch (p.ksplit > 1): only the slab code is instantiated.
template <int PREC, int WM, int MT, int TXL, int TYL, bool WIDE = false, bool ZPAIRS = false, bool KSPLIT = false>
// pre_ws / pre_bias (pre = true): p.wscale[cout] and the lane's bias, loaded by the caller at kernel
// START (conv3d_wz.h): at the epilogue's start they are a dependent global load -- ~2k cycles at the
// head of a phase in which the wave issues no MFMA
__device__ __forceinline__ void conv_epilogue(const ConvK& p, const f32x16 (&acc)[MT], int n, int z0, int y0,
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
                        // raw slab of a split launch (the reduce launch reads it): plain 16-byte stores (epi_store_b128)
                        if (split)
                            epi_store_b128(u32x4{__builtin_bit_cast(unsigned, a[0]), __builtin_bit_cast(unsigned, a[1]),
                                                 __builtin_bit_cast(unsigned, a[2]), __builtin_bit_cast(unsigned, a[3])}, drsrc, wv, so);
                        else if (KSPLIT) { }
                        else if (o16)
                            epi_store_b64(u32x2{half_pack(a[0], a[1], f16), half_pack(a[2], a[3], f16)}, drsrc, wv, so);
                        else
                            epi_store_b128(u32x4{__builtin_bit_cast(unsigned, a[0]), __builtin_bit_cast(unsigned, a[1]),
                                                 __builtin_bit_cast(unsigned, a[2]), __builtin_bit_cast(unsigned, a[3])}, drsrc, wv, so);
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
                return;
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
            return;
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
        return;
    }
    if constexpr (KSPLIT) return;
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
}


