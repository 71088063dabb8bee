// 3x3x3 convolution with Winograd F(2,3) along the DEPTH axis, split-f16 arithmetic
// (DDPM3D_PREC_F16X3_WZ), for the layers that carry the network's FLOPs: 8x8x2 tiles,
// 128-cout workgroups with every wave active, pipelined inputs (IN_SAME / IN_UP).
//
// For one output pair (z, z+1) at a fixed (y, x) and one (dy, dx):
//     V0 = d0 - d2   V1 = d1 + d2   V2 = d2 - d1   V3 = d1 - d3        (d_k = input plane z-1+k)
//     U0 = g0        U1 = (g0+g1+g2)/2   U2 = (g0-g1+g2)/2   U3 = g2    (g_k = weight at dz = k)
//     M_j = sum over (ci, dy, dx) of U_j * V_j
//     out(z) = M0 + M1 + M2        out(z+1) = M1 - M2 - M3
// i.e. 4 products per two outputs instead of 6: 36 "taps" (j, dy, dx) per 16-channel chunk
// feeding four accumulator sets, 216 MFMAs per wave and chunk instead of 324.
//
// Why depth: D is never strided in this network (unet.py:129), the workgroup tile is 8x8x2
// = exactly ONE z-pair per (y, x), and that pair's four halo planes are exactly the
// transform's four inputs -- so the transformed image V has the SAME LDS footprint as the
// plain halo image (4 planes of 10x10 voxels) and a tap is, as before, a compile-time LDS
// offset (plane j instead of plane dz).  The input transform is done in fp32 on the
// normalised+activated values while staging (before the f16 hi/lo split); the weight
// transform at pack time (ops.hip); the output transform is register-local in the epilogue
// (the four M_j of an output live in the same lane and register index).
//
// The four accumulator sets cost 128 VGPRs, so this kernel runs two workgroups per CU
// whatever else it does.  DB = 1 therefore also double-buffers the LDS image (2 x 35 KB still
// fits twice) and spreads the staging of chunk c+1 over the tap loop of chunk c: one
// barrier per chunk, and the staging VALU / LDS stores co-issue with the MFMAs of the same
// wave instead of idling the matrix pipe.
#pragma once
#include "conv3d_db.h"

template <int DB>
__global__ __launch_bounds__(256, 2) void conv3d_wz_kernel(const ConvK p) {
    constexpr int CK = DDPM3D_CONV_CK, NT = 36;
    constexpr int TX = 8, TXL = 3, TYL = 3;
    constexpr int HX = 10, HY = 10, NP = 4;          // NP: input planes = transformed planes
    constexpr int VS = 5;
    constexpr int RY = LdsGeom<TX, HX, HY>::RY;
    constexpr int RZ = LdsGeom<TX, HX, HY>::RZ;
    constexpr int BUF = NP * RZ * 16;                // bytes of one transformed image
    constexpr int QPV = CK / 4;
    constexpr int HC = HX * HY * QPV;                // staging items: (y, x, channel quad) columns
    constexpr int NL = (HC + 255) / 256;
    static_assert(2 + 10 * (NL - 1) + 8 < NT - 3, "staging must finish before the next raw loads are issued");

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;

    const WgId wg = wg_id(p);
    int tile = wg.tile;
    const int tx_i = tile % p.tilesX; tile /= p.tilesX;
    const int ty_i = tile % p.tilesY; tile /= p.tilesY;
    const int tz_i = tile % p.tilesZ; tile /= p.tilesZ;
    const int n = tile;
    const int x0 = tx_i * TX, y0 = ty_i * 8, z0 = tz_i * 2;

    // GEMM rows of this wave: the 64 (y, x) positions of the tile, two 32-row MFMA tiles
    int arow[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int m = t * 32 + (lane & 31);
        arow[t] = ((m >> TXL) * RY + (m & (TX - 1)) * VS + half) * 16;
    }

    const int cout = wg.cy * 128 + wn * 32 + (lane & 31);
    const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(p.w, p.w_bytes);
    const unsigned wlane = ((unsigned)cout * 2 + half) * 16;
    const unsigned wpart = (unsigned)p.CoutPad * 32;
    const unsigned wchunk_stride = 2 * wpart;
    const unsigned wtap_stride = (unsigned)(p.CinPad / CK) * wchunk_stride;

    f32x16 acc[4][2];   // [transformed plane j][row tile]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[j][t][i] = 0.0f;

    const int nchunks = p.CinPad / CK;
    const int chunk_begin = wg.split * p.chunks_per_split;
    const int chunk_end = min(nchunks, chunk_begin + p.chunks_per_split);

    const int q = tid % QPV;
    const int up_shift = p.in_mode == DDPM3D_IN_UP ? 1 : 0;
    const unsigned act_mask = p.act ? 0xFFFFFFFFu : 0u;
    HaloSrc hs = halo_src<CK>(p, n, chunk_begin < chunk_end ? chunk_begin : 0, q);
    const int plane = hs.Hs * hs.Ws;                 // source voxels per z-plane
    // per item: source voxel of input plane 1 (z = z0, always inside the volume) at its (y, x),
    // or -1 outside H x W; plane k is (k - 1) source planes away
    int vox0[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int idx = tid + i * 256;
        const int hyx = idx / QPV;
        const int hy = hyx / HX, hx = hyx - hy * HX;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        const bool ok = idx < HC && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        vox0[i] = ok ? ((n * p.D + z0) * hs.Hs + (y >> up_shift)) * hs.Ws + (x >> up_shift) : -1;
    }
    f32x4 raw[NL][NP];   // raw source values, then (in place) the normalised+activated d_k
    auto issue_raw = [&](const HaloSrc& h) {
        const __amdgpu_buffer_rsrc_t srsrc = make_rsrc(h.src, h.src_bytes);
        const unsigned row_bytes = (unsigned)h.Cs * 4, soff = (unsigned)h.cb * 4;
#pragma unroll
        for (int i = 0; i < NL; ++i)
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const bool zok = (unsigned)(z0 - 1 + k) < (unsigned)p.D;     // uniform per workgroup
                const unsigned voff = (vox0[i] < 0 || !zok)
                                          ? DDPM3D_OOB_OFFSET
                                          : (unsigned)(vox0[i] + (k - 1) * plane) * row_bytes + q * 16;
                raw[i][k] = __builtin_bit_cast(f32x4, buffer_load16(srsrc, voff, soff));
            }
    };
    // raw -> d (affine + SiLU, exact zero outside the volume), in place
    auto finish_d = [&](const int i) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const bool inb = vox0[i] >= 0 && (unsigned)(z0 - 1 + k) < (unsigned)p.D;
            raw[i][k] = halo_finish<true>(hs, raw[i][k], inb, act_mask);
        }
    };
    // transformed plane j of item i: V_j, x8, f16 hi/lo split, store into the image at `buf`
    auto store_j = [&](const int i, const int j, unsigned char* buf) {
        const int idx = tid + i * 256;
        if (idx < HC) {
            const int hyx = idx / QPV;
            const int hy = hyx / HX, hx = hyx - hy * HX;
            const f32x4 v = j == 0 ? raw[i][0] - raw[i][2]
                          : j == 1 ? raw[i][1] + raw[i][2]
                          : j == 2 ? raw[i][2] - raw[i][1]
                                   : raw[i][1] - raw[i][3];
            h4 hi, lo;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float s = fminf(fmaxf(v[c] * DDPM3D_X3_ACT_SCALE, -60000.0f), 60000.0f);
                hi[c] = (_Float16)s;
                lo[c] = (_Float16)(s - (float)hi[c]);
            }
            unsigned char* vrow = buf + (j * RZ + hy * RY + hx * VS) * 16;
            *reinterpret_cast<h4*>(vrow + q * 8) = hi;
            *reinterpret_cast<h4*>(vrow + 32 + q * 8) = lo;
        }
    };

    if (chunk_begin < chunk_end) issue_raw(hs);
    if (DB) {
        // prologue: image 0 <- first chunk, raw <- second chunk
        if (chunk_begin < chunk_end) {
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                finish_d(i);
#pragma unroll
                for (int j = 0; j < 4; ++j) store_j(i, j, lds);
            }
            if (chunk_begin + 1 < chunk_end) {
                hs = halo_src<CK>(p, n, chunk_begin + 1, q);
                issue_raw(hs);
            }
        }
        __syncthreads();
    }

    int par = 0;
    for (int chunk = chunk_begin; chunk < chunk_end; ++chunk) {
        const bool more = chunk + 1 < chunk_end, more2 = chunk + 2 < chunk_end;
        const unsigned char* bufc = lds + (DB ? par * BUF : 0);
        unsigned char* bufn = lds + (DB ? (par ^ 1) * BUF : 0);
        if (!DB) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                finish_d(i);
#pragma unroll
                for (int j = 0; j < 4; ++j) store_j(i, j, lds);
            }
            __syncthreads();
            if (more) hs = halo_src<CK>(p, n, chunk + 1, q);
        }

        // ---- 36 taps (j, dy, dx); weight ring of 3 taps, prefetch distance 2
        const unsigned wchunk = (unsigned)chunk * wchunk_stride;
        u32x4 bq[3][2];
        bq[0][0] = buffer_load16(wrsrc, wlane, wchunk);
        bq[0][1] = buffer_load16(wrsrc, wlane, wchunk + wpart);
        bq[1][0] = buffer_load16(wrsrc, wlane, wchunk + wtap_stride);
        bq[1][1] = buffer_load16(wrsrc, wlane, wchunk + wtap_stride + wpart);
        h8 af[2][2][2];   // [slot][row tile][hi|lo]: A operands, read one tap ahead
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            af[0][t][0] = *reinterpret_cast<const h8*>(bufc + arow[t]);
            af[0][t][1] = *reinterpret_cast<const h8*>(bufc + arow[t] + 32);
        }
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
            // pin the tap boundary: the scheduler otherwise moves the LDS reads (and, in long
            // straight-line runs, the weight loads) down to their first use
            __builtin_amdgcn_sched_barrier(0);
            if (tap + 1 < NT) {
                const int t1 = tap + 1;
                const int off1 = ((t1 / 9) * RZ + ((t1 / 3) % 3) * RY + (t1 % 3) * VS) * 16;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    af[t1 & 1][t][0] = *reinterpret_cast<const h8*>(bufc + arow[t] + off1);
                    af[t1 & 1][t][1] = *reinterpret_cast<const h8*>(bufc + arow[t] + off1 + 32);
                }
            }
            if (tap + 2 < NT) {
                bq[(tap + 2) % 3][0] = buffer_load16(wrsrc, wlane, wchunk + (tap + 2) * wtap_stride);
                bq[(tap + 2) % 3][1] = buffer_load16(wrsrc, wlane, wchunk + (tap + 2) * wtap_stride + wpart);
            }
            __builtin_amdgcn_sched_barrier(0);   // prefetches issue BEFORE this tap's MFMAs
            if (DB) {
                // staging of chunk+1 spread over the taps: item i at taps 2+10i (finish) and
                // 4,6,8,10 + 10i (one transformed plane each), into the other image
                if (more && tap >= 2 && (tap - 2) / 10 < NL) {
                    const int i = (tap - 2) / 10, ph = (tap - 2) % 10;
                    if (ph == 0) finish_d(i);
                    else if ((ph & 1) == 0) store_j(i, ph / 2 - 1, bufn);
                }
                if (tap == NT - 3 && more2) {       // after the chunk's last weight loads (vmcnt order)
                    hs = halo_src<CK>(p, n, chunk + 2, q);
                    issue_raw(hs);
                }
            } else {
                if (tap == NT - 3 && more) issue_raw(hs);
            }
            const int j = tap / 9;
            const h8 bhi = __builtin_bit_cast(h8, bq[tap % 3][0]);
            const h8 blo = __builtin_bit_cast(h8, bq[tap % 3][1]);
            // per accumulator the order stays lo*hi, hi*lo, hi*hi; the two row tiles alternate
#pragma unroll
            for (int t = 0; t < 2; ++t)
                acc[j][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[tap & 1][t][1], bhi, acc[j][t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; ++t)
                acc[j][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[tap & 1][t][0], blo, acc[j][t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; ++t)
                acc[j][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[tap & 1][t][0], bhi, acc[j][t], 0, 0, 0);
        }
        if (DB) {
            __syncthreads();   // image (par^1) complete, image par free for chunk+2's staging
            par ^= 1;
        }
    }

    // ---- output transform (register-local), then the common epilogue on the 8x8x2 tile:
    // virtual accumulator u = zbit*2 + t covers rows m = u*32 + row -> (tz = zbit, ty, tx)
    f32x16 outv[4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        outv[t] = acc[0][t] + acc[1][t] + acc[2][t];
        outv[2 + t] = acc[1][t] - acc[2][t] - acc[3][t];
    }
    const int tile_in_n = (tz_i * p.tilesY + ty_i) * p.tilesX + tx_i;
    conv_epilogue<1, 1, 4, TXL, TYL>(p, outv, n, z0, y0, x0, tile_in_n, 0, cout, half, wg.split);
}
