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
#pragma once
#include "conv3d_db.h"

template <int UNUSED = 0>
__global__ __launch_bounds__(256, 2) void conv3d_wz_kernel(const ConvK p) {
    constexpr int CK = DDPM3D_CONV_CK, NT = 36;
    constexpr int TX = 8, TY = 8, TXL = 3, TYL = 3;
    constexpr int HX = 10, HY = 10, NP = 4;          // NP: input planes = transformed planes
    constexpr int VS = 5;
    constexpr int RY = LdsGeom<TX, HX, HY>::RY;
    constexpr int RZ = LdsGeom<TX, HX, HY>::RZ;
    constexpr int QPV = CK / 4;
    constexpr int HC = HX * HY * QPV;                // staging items: (y, x, channel quad) columns
    constexpr int NL = (HC + 255) / 256;

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;

    int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tx_i = tile % p.tilesX; tile /= p.tilesX;
    const int ty_i = tile % p.tilesY; tile /= p.tilesY;
    const int tz_i = tile % p.tilesZ; tile /= p.tilesZ;
    const int n = tile;
    const int x0 = tx_i * TX, y0 = ty_i * TY, z0 = tz_i * 2;

    // GEMM rows of this wave: the 64 (y, x) positions of the tile, two 32-row MFMA tiles
    int arow[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int m = t * 32 + (lane & 31);
        arow[t] = ((m >> TXL) * RY + (m & (TX - 1)) * VS + half) * 16;
    }

    const int cout = blockIdx.y * 128 + wn * 32 + (lane & 31);
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
    const int chunk_begin = blockIdx.z * p.chunks_per_split;
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
    f32x4 raw[NL][NP];
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
    if (chunk_begin < chunk_end) issue_raw(hs);

    for (int chunk = chunk_begin; chunk < chunk_end; ++chunk) {
        __syncthreads();
        // ---- finish the prefetched columns: affine + SiLU, input transform, f16 split, store
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int idx = tid + i * 256;
            if (idx < HC) {
                const int hyx = idx / QPV;
                const int hy = hyx / HX, hx = hyx - hy * HX;
                f32x4 d[NP];
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const bool inb = vox0[i] >= 0 && (unsigned)(z0 - 1 + k) < (unsigned)p.D;
                    d[k] = halo_finish<true>(hs, raw[i][k], inb, act_mask);
                }
                const f32x4 v[4] = {d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    h4 hi, lo;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float s = fminf(fmaxf(v[j][c] * DDPM3D_X3_ACT_SCALE, -60000.0f), 60000.0f);
                        hi[c] = (_Float16)s;
                        lo[c] = (_Float16)(s - (float)hi[c]);
                    }
                    unsigned char* vrow = lds + (j * RZ + hy * RY + hx * VS) * 16;
                    *reinterpret_cast<h4*>(vrow + q * 8) = hi;
                    *reinterpret_cast<h4*>(vrow + 32 + q * 8) = lo;
                }
            }
        }
        __syncthreads();
        const bool more = chunk + 1 < chunk_end;
        if (more) hs = halo_src<CK>(p, n, chunk + 1, q);

        // ---- 36 taps (j, dy, dx); weight ring of 3 taps, prefetch distance 2
        const unsigned wchunk = (unsigned)chunk * wchunk_stride;
        u32x4 bq[3][2];
        bq[0][0] = buffer_load16(wrsrc, wlane, wchunk);
        bq[0][1] = buffer_load16(wrsrc, wlane, wchunk + wpart);
        bq[1][0] = buffer_load16(wrsrc, wlane, wchunk + wtap_stride);
        bq[1][1] = buffer_load16(wrsrc, wlane, wchunk + wtap_stride + wpart);
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
            if (tap + 2 < NT) {
                bq[(tap + 2) % 3][0] = buffer_load16(wrsrc, wlane, wchunk + (tap + 2) * wtap_stride);
                bq[(tap + 2) % 3][1] = buffer_load16(wrsrc, wlane, wchunk + (tap + 2) * wtap_stride + wpart);
            }
            if (tap == NT - 3 && more) issue_raw(hs);   // after the chunk's last weight loads (vmcnt order)
            const int j = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
            const int tapoff = (j * RZ + dy * RY + dx * VS) * 16;
            const h8 bhi = __builtin_bit_cast(h8, bq[tap % 3][0]);
            const h8 blo = __builtin_bit_cast(h8, bq[tap % 3][1]);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const h8 ahi = *reinterpret_cast<const h8*>(lds + arow[t] + tapoff);
                const h8 alo = *reinterpret_cast<const h8*>(lds + arow[t] + tapoff + 32);
                acc[j][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi, acc[j][t], 0, 0, 0);
                acc[j][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo, acc[j][t], 0, 0, 0);
                acc[j][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi, acc[j][t], 0, 0, 0);
            }
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
    conv_epilogue<1, 1, 4, TXL, TYL>(p, outv, n, z0, y0, x0, tile_in_n, 0, cout, half);
}
