// Double-buffered variant of the 3x3x3 conv kernel (see conv3d.hip for the common design)
// for the shapes that carry the network's FLOPs: pipelined inputs (IN_SAME / IN_UP), a
// 128-voxel x 128-cout tile, every wave active (CoutPad % 128 == 0).
//
// The single-buffer kernel has, per 16-channel chunk, barrier -> staging (affine + SiLU +
// f16 split + LDS stores, ~500 VALU per thread) -> barrier -> 27 taps of MFMA; during the
// staging phase this workgroup's matrix pipes idle and only the other resident workgroups
// cover it.  Here the LDS holds two halo images: while the taps of chunk c read image c&1,
// the staging of chunk c+1 is spread over the tap loop (one item after every third tap, VALU
// and LDS stores co-issuing with the MFMAs of the same wave) into image (c+1)&1, and the raw
// loads of chunk c+2 are issued at tap 24, after the chunk's last weight loads (vmcnt retires
// in order).  One barrier per chunk.
#pragma once
#include "conv3d_load.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

// smallest x >= v with x % 16 == res
constexpr int pad_to_residue(int v, int res) { return v + ((res - v % 16) + 16) % 16; }

// LDS strides (in 16-byte slots) of the halo image; see the bank-conflict note in conv3d.hip.
template <int TX, int HX, int HY>
struct LdsGeom {
    static constexpr int RY = pad_to_residue(HX * 5, TX == 8 ? 8 : 4);
    static constexpr int RZ = TX == 8 ? HY * RY : pad_to_residue(HY * RY, 0);
};

// Epilogue shared by the conv kernels: bias / residual / store / GroupNorm partial sums, or
// the raw split-K slab.  C/D map of a 32x32 MFMA: col = lane&31 (cout),
// row = (reg&3) + 8*(reg>>2) + 4*half.
template <int PREC, int WM, int MT, int TXL, int TYL>
__device__ __forceinline__ void conv_epilogue(const ConvK& p, const f32x16 (&acc)[MT], int n, int z0, int y0,
                                              int x0, int tile_in_n, int wm, int cout, int half, int ksplit_idx) {
    constexpr int TX = 1 << TXL, TY = 1 << TYL;
    const bool cvalid = cout < p.Cout;
    const size_t DHW = (size_t)p.D * p.H * p.W;
    // PREC 1/2: undo the operand scaling (exact: a power of two per cout)
    const float oscale = (PREC != 0 && cvalid) ? p.wscale[cout] : 1.0f;

    // Fast path -- every launch of the network except ragged edge tiles, the NCDHW output conv
    // and the up/down-sampling skip sums: the tile lies inside the volume, so element (t, reg)
    // of a lane sits at  lane base (cout, half) + a WAVE-UNIFORM offset.  Stores (and the
    // same-shaped residual loads) are buffer instructions with that offset in an SGPR: no
    // per-element index arithmetic, bounds test or 64-bit address.  The general path below
    // costs ~200 instructions per element, which for a 128-cin layer was a third of the
    // kernel.  Same values in the same order, so both paths agree bit for bit.
    {
        constexpr int TZ = WM * MT * 32 / (TX * TY);
        const bool split = p.ksplit > 1;
        const bool full = z0 + TZ <= p.D && y0 + TY <= p.H && x0 + TX <= p.W;
        // residual of a split conv is the reduce kernel's business
        const int rm = split ? DDPM3D_RES_NONE : p.res_mode;
        // the residual tensor of the up / down ResBlocks is the block input at the other
        // resolution (H/2 x W/2 or 2H x 2W); its sample must be 32-bit addressable too
        const size_t Hr = rm == DDPM3D_RES_UP ? p.H / 2 : (rm == DDPM3D_RES_POOL ? (size_t)p.H * 2 : p.H);
        const size_t Wr = rm == DDPM3D_RES_UP ? p.W / 2 : (rm == DDPM3D_RES_POOL ? (size_t)p.W * 2 : p.W);
        const size_t samp_r = (size_t)p.D * Hr * Wr * p.Cout;
        if (full && p.out_layout == DDPM3D_OUT_NDHWC && samp_r * 4 < 0xFFFFFFF0ull) {
            const size_t samp = DHW * p.Cout;                       // elements per sample (< 2^30, C ABI guard)
            const unsigned cstride = (unsigned)p.Cout * 4;          // bytes per voxel
            float* dst = split ? p.partial + ((size_t)ksplit_idx * p.N + n) * samp : p.out + (size_t)n * samp;
            const __amdgpu_buffer_rsrc_t drsrc = make_rsrc(dst, (unsigned)(samp * 4));
            const bool resid = rm != DDPM3D_RES_NONE;
            const __amdgpu_buffer_rsrc_t rrsrc =
                make_rsrc(resid ? p.res + (size_t)n * samp_r : dst, (unsigned)((resid ? samp_r : samp) * 4));
            // the lane's half adds 4 to the MFMA row: 4 voxels in x (8-wide tile) or one row in y (4-wide)
            const unsigned hx = (4 * half) & (TX - 1), hy = ((4 * half) >> TXL) & (TY - 1);
            const unsigned vbase = (((unsigned)z0 * p.H + y0 + hy) * p.W + x0 + hx) * cstride + (unsigned)cout * 4;
            const unsigned voff = cvalid ? vbase : DDPM3D_OOB_OFFSET;   // out-of-range lanes: loads 0, stores dropped
            // Residual addressing, also "lane base + wave-uniform offset" (tiles start at even y0, x0):
            //   SAME  x[z][y][x]                      : the output's own offsets
            //   UP    x[z][y>>1][x>>1] (unet.py:241)  : (t0 + h) >> 1 = (t0 >> 1) + (h >> 1) -- the half's
            //         4 voxels in x are 2 source voxels; in a 4-wide tile its one row in y shares the source row
            //   POOL  mean of x[z][2y+{0,1}][2x+{0,1}]: doubled offsets, four loads
            const unsigned rW = (unsigned)Wr, rH = (unsigned)Hr;
            unsigned rbase = vbase;
            if (rm == DDPM3D_RES_UP)
                rbase = (((unsigned)z0 * rH + (y0 >> 1)) * rW + (x0 >> 1) + (hx >> 1)) * cstride + (unsigned)cout * 4;
            else if (rm == DDPM3D_RES_POOL)
                rbase = (((unsigned)z0 * rH + 2 * (y0 + hy)) * rW + 2 * (x0 + hx)) * cstride + (unsigned)cout * 4;
            const unsigned roff = cvalid ? rbase : DDPM3D_OOB_OFFSET;
            const float bias = (!split && cvalid) ? p.bias[(size_t)n * p.bias_stride_n + cout] : 0.0f;
            float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                unsigned soff[16];
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int m0 = (wm * MT + t) * 32 + (reg & 3) + 8 * (reg >> 2);
                    const int tx = m0 & (TX - 1), ty = (m0 >> TXL) & (TY - 1), tz = m0 >> (TXL + TYL);
                    soff[reg] = (unsigned)((tz * p.H + ty) * p.W + tx) * cstride;
                }
                float r[16];
                if (rm == DDPM3D_RES_SAME) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg)
                        r[reg] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrsrc, roff, soff[reg], 0));
                } else if (rm == DDPM3D_RES_UP) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int m0 = (wm * MT + t) * 32 + (reg & 3) + 8 * (reg >> 2);
                        const int tx = m0 & (TX - 1), ty = (m0 >> TXL) & (TY - 1), tz = m0 >> (TXL + TYL);
                        const unsigned so = (unsigned)((tz * rH + (ty >> 1)) * rW + (tx >> 1)) * cstride;
                        r[reg] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrsrc, roff, so, 0));
                    }
                } else if (rm == DDPM3D_RES_POOL) {
                    // AvgPool3d window order (h, w): ((r00 + r01) + r10) + r11, then * 1/4 (ddpm3d_residual)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int m0 = (wm * MT + t) * 32 + (reg & 3) + 8 * (reg >> 2);
                        const int tx = m0 & (TX - 1), ty = (m0 >> TXL) & (TY - 1), tz = m0 >> (TXL + TYL);
                        const unsigned so = (unsigned)((tz * rH + 2 * ty) * rW + 2 * tx) * cstride;
                        const float r00 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrsrc, roff, so, 0));
                        const float r01 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrsrc, roff, so + cstride, 0));
                        const float r10 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrsrc, roff, so + rW * cstride, 0));
                        const float r11 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrsrc, roff, so + (rW + 1) * cstride, 0));
                        r[reg] = (((r00 + r01) + r10) + r11) * 0.25f;
                    }
                }
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    float val = PREC != 0 ? acc[t][reg] * oscale : acc[t][reg];
                    if (!split) {
                        val += bias;
                        if (resid) val += r[reg];
                        s1 += val;
                        s2 = fmaf(val, val, s2);
                    }
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), drsrc, voff, soff[reg], 0);
                }
            }
            if (!split && p.stats != nullptr) {
                s1 += __shfl_xor(s1, 32);
                s2 += __shfl_xor(s2, 32);
                if (half == 0 && cvalid) {
                    const size_t row = (size_t)tile_in_n * WM + wm;
                    *reinterpret_cast<float2*>(p.stats + (((size_t)n * p.Cout + cout) * p.stats_rows + row) * 2) =
                        make_float2(s1, s2);
                }
            }
            return;
        }
    }
    if (p.ksplit > 1) {
        // split-K: raw partial sums to this split's slab; bias / residual / statistics
        // are applied by the reduce kernel once all splits are in
        float* slab = p.partial + ((size_t)ksplit_idx * p.N + n) * DHW * p.Cout;
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
                        PREC != 0 ? acc[t][reg] * oscale : acc[t][reg];
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
                float val = (PREC != 0 ? acc[t][reg] * oscale : acc[t][reg]) + bias;
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
            // channel-major [N][Cout][rows][2]: a GroupNorm group's partial sums are contiguous
            const size_t row = (size_t)tile_in_n * WM + wm;
            float2 v2 = make_float2(s1, s2);
            *reinterpret_cast<float2*>(p.stats + (((size_t)n * p.Cout + cout) * p.stats_rows + row) * 2) = v2;
        }
    }
}

template <int PREC, int TXL, int TYL>
__global__ __launch_bounds__(256, 2) void conv3d_db_kernel(const ConvK p) {
    constexpr int CK = DDPM3D_CONV_CK, MT = 4, NT = 27, PAD = 1;
    constexpr int TX = 1 << TXL, TY = 1 << TYL, TZ = 128 / (TX * TY);
    constexpr int HX = TX + 2, HY = TY + 2, HZ = TZ + 2;
    constexpr int HV = HX * HY * HZ;
    constexpr int VS = 5;
    constexpr int RY = LdsGeom<TX, HX, HY>::RY;
    constexpr int RZ = LdsGeom<TX, HX, HY>::RZ;
    constexpr int BUF = HZ * RZ * 16;                 // bytes of one halo image
    constexpr int QPV = CK / 4;
    constexpr int NL = (HV * QPV + 255) / 256;        // staging items per thread
    static_assert(1 + 3 * (NL - 1) < NT - 3, "staging must finish before the next raw loads are issued");

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);   // 4 waves along Cout
    const int half = lane >> 5;

    int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tx_i = tile % p.tilesX; tile /= p.tilesX;
    const int ty_i = tile % p.tilesY; tile /= p.tilesY;
    const int tz_i = tile % p.tilesZ; tile /= p.tilesZ;
    const int n = tile;
    const int x0 = tx_i * TX, y0 = ty_i * TY, z0 = tz_i * TZ;

    int arow[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = t * 32 + (lane & 31);
        const int tx = m & (TX - 1), ty = (m >> TXL) & (TY - 1), tz = m >> (TXL + TYL);
        arow[t] = (tz * RZ + ty * RY + tx * VS + half) * 16;
    }

    const int cout = blockIdx.y * 128 + wn * 32 + (lane & 31);
    const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(p.w, p.w_bytes);
    const unsigned wlane = ((unsigned)cout * 2 + half) * 16;
    const unsigned wpart = (unsigned)p.CoutPad * 32;
    const unsigned wchunk_stride = 2 * wpart;
    const unsigned wtap_stride = (unsigned)(p.CinPad / CK) * wchunk_stride;

    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;

    const int nchunks = p.CinPad / CK;
    const int chunk_begin = blockIdx.z * p.chunks_per_split;
    const int chunk_end = min(nchunks, chunk_begin + p.chunks_per_split);

    const int q = tid % QPV;
    const int up_shift = p.in_mode == DDPM3D_IN_UP ? 1 : 0;
    const unsigned act_mask = p.act ? 0xFFFFFFFFu : 0u;
    f32x4 raw[NL];
    int vox[NL];
    HaloSrc hs = halo_src<CK>(p, n, chunk_begin < chunk_end ? chunk_begin : 0, q);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int idx = tid + i * 256;
        const int hv = idx / QPV;
        const int hz = hv / (HY * HX);
        const int rem = hv - hz * (HY * HX);
        const int hy = rem / HX;
        const int hx = rem - hy * HX;
        vox[i] = idx < HV * QPV ? halo_vox(p, n, z0 - PAD + hz, y0 - PAD + hy, x0 - PAD + hx, hs.Hs, hs.Ws, up_shift)
                                : -1;
    }
    auto issue_raw = [&](const HaloSrc& h) {
        const __amdgpu_buffer_rsrc_t srsrc = make_rsrc(h.src, h.src_bytes);
        const unsigned row_bytes = (unsigned)h.Cs * 4, soff = (unsigned)h.cb * 4;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const unsigned voff = vox[i] < 0 ? DDPM3D_OOB_OFFSET : (unsigned)vox[i] * row_bytes + q * 16;
            raw[i] = __builtin_bit_cast(f32x4, buffer_load16(srsrc, voff, soff));
        }
    };
    // affine + activation + (split) + store of staging item i into the image at `buf`
    auto finish_item = [&](const int i, unsigned char* buf) {
        const int idx = tid + i * 256;
        if (idx < HV * QPV) {
            const int hv = idx / QPV;
            const int hz = hv / (HY * HX);
            const int rem = hv - hz * (HY * HX);
            const int hy = rem / HX;
            const int hx = rem - hy * HX;
            const f32x4 v = halo_finish<PREC != 0>(hs, raw[i], vox[i] >= 0, act_mask);
            unsigned char* vrow = buf + (hz * RZ + hy * RY + hx * VS) * 16;
            if constexpr (PREC == 0) {
                *reinterpret_cast<f32x4*>(vrow + q * 16) = v;
            } else {
                h4 hi, lo;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float s = fminf(fmaxf(v[c] * DDPM3D_X3_ACT_SCALE, -60000.0f), 60000.0f);
                    hi[c] = (_Float16)s;
                    lo[c] = (_Float16)(s - (float)hi[c]);
                }
                *reinterpret_cast<h4*>(vrow + q * 8) = hi;
                if constexpr (PREC == 1) *reinterpret_cast<h4*>(vrow + 32 + q * 8) = lo;
            }
        }
    };

    // prologue: image 0 <- first chunk; raw <- second chunk
    if (chunk_begin < chunk_end) {
        issue_raw(hs);
#pragma unroll
        for (int i = 0; i < NL; ++i) finish_item(i, lds);
        if (chunk_begin + 1 < chunk_end) {
            hs = halo_src<CK>(p, n, chunk_begin + 1, q);
            issue_raw(hs);
        }
    }
    __syncthreads();

    constexpr bool LO = PREC != 2;
    int par = 0;
    for (int chunk = chunk_begin; chunk < chunk_end; ++chunk) {
        const bool more = chunk + 1 < chunk_end, more2 = chunk + 2 < chunk_end;
        const unsigned char* bufc = lds + par * BUF;
        unsigned char* bufn = lds + (par ^ 1) * BUF;
        const unsigned wchunk = (unsigned)chunk * wchunk_stride;
        u32x4 bq[3][2];
        bq[0][0] = buffer_load16(wrsrc, wlane, wchunk);
        if (LO) bq[0][1] = buffer_load16(wrsrc, wlane, wchunk + wpart);
        bq[1][0] = buffer_load16(wrsrc, wlane, wchunk + wtap_stride);
        if (LO) bq[1][1] = buffer_load16(wrsrc, wlane, wchunk + wtap_stride + wpart);
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
            if (tap + 2 < NT) {
                bq[(tap + 2) % 3][0] = buffer_load16(wrsrc, wlane, wchunk + (tap + 2) * wtap_stride);
                if (LO) bq[(tap + 2) % 3][1] = buffer_load16(wrsrc, wlane, wchunk + (tap + 2) * wtap_stride + wpart);
            }
            // staging of chunk+1, one item after every third tap (raw[] was loaded during the
            // previous chunk's last taps, hs holds chunk+1's affine quad)
            if ((tap % 3) == 1 && (tap / 3) < NL && more) finish_item(tap / 3, bufn);
            // all of this chunk's weight loads are in flight and raw[] is consumed: chunk+2's loads
            if (tap == NT - 3 && more2) {
                hs = halo_src<CK>(p, n, chunk + 2, q);
                issue_raw(hs);
            }
            const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
            const int tapoff = (dz * RZ + dy * RY + dx * VS) * 16;
            const u32x4 b0 = bq[tap % 3][0], b1 = LO ? bq[tap % 3][1] : b0;
            if constexpr (PREC == 0) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    f32x4 a[MT];
#pragma unroll
                    for (int t = 0; t < MT; ++t)
                        a[t] = *reinterpret_cast<const f32x4*>(bufc + arow[t] + tapoff + kk * 32);
                    const f32x4 b = __builtin_bit_cast(f32x4, kk == 0 ? b0 : b1);
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int t = 0; t < MT; ++t)
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][s], b[s], acc[t], 0, 0, 0);
                }
            } else if constexpr (PREC == 2) {
                const h8 bhi = __builtin_bit_cast(h8, b0);
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const h8 ahi = *reinterpret_cast<const h8*>(bufc + arow[t] + tapoff);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi, acc[t], 0, 0, 0);
                }
            } else {
                const h8 bhi = __builtin_bit_cast(h8, b0);
                const h8 blo = __builtin_bit_cast(h8, b1);
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const h8 ahi = *reinterpret_cast<const h8*>(bufc + arow[t] + tapoff);
                    const h8 alo = *reinterpret_cast<const h8*>(bufc + arow[t] + tapoff + 32);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi, acc[t], 0, 0, 0);
                }
            }
        }
        __syncthreads();  // image (par^1) complete, image par free for chunk+2's staging
        par ^= 1;
    }

    const int tile_in_n = (tz_i * p.tilesY + ty_i) * p.tilesX + tx_i;
    conv_epilogue<PREC, 1, MT, TXL, TYL>(p, acc, n, z0, y0, x0, tile_in_n, 0, cout, half, blockIdx.z);
}
