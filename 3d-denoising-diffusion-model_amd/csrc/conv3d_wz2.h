// Winograd-D 3x3x3 convolution (see conv3d_wz.h for the transform), large-register form:
// ONE wave per SIMD, so each wave owns the unified 512-register file -- 256 accumulator
// registers (4 transformed planes x 2 row tiles x 2 cout tiles, in AGPRs) + 256 VGPRs.
//
// Why: in conv3d_wz_kernel the four accumulator sets cap the wave tile at 64 rows x 32 couts,
// and every MFMA triple then needs 2/3 of a 1 KB LDS read and 1/3 of a 1 KB weight load (the
// per-wave weight stream is 295 KB per chunk and workgroup out of L2).  Here the wave tile is
// 64 rows x 64 couts and the workgroup tile 8x8x4 voxels x 128 couts (waves = 2 z-pairs x
// 2 cout halves):
//   - LDS read bytes per MFMA: 2/3 of v1;  weight bytes per MFMA out of L1: 1/2 (and the two
//     z-pair waves of a cout half fetch the same lines, so 1/4 out of L2);
//   - 6 raw input planes feed 4 output planes (v1: 4 feed 2): 3/4 of the SiLU evaluations.
// With one workgroup per CU a barrier idles the whole CU, so the LDS image is double-buffered
// (2 x 70 KB) and the staging of chunk c+1 runs inside the tap loop of chunk c; the weight
// ring runs across chunk boundaries (36 % RING == 0 keeps its phase); the A operands are read
// one tap ahead.  sched_barriers pin both prefetches: the machine scheduler otherwise sinks
// loads down to their first use and the prefetch distance collapses.
//
// Measured on MI355X (r01, ms, v1 -> this kernel): 256->128 @64^3 1.038 -> 0.982,
// 512->256 @64x32x32 0.985 -> 0.946, 512->512 @64x8x8 0.152 -> 0.140; 128->128 @64^3
// 0.545 -> 0.557 (8 chunks: the un-overlapped prologue/epilogue of a lone workgroup weighs
// more).  Once v1 also read its A operands a tap ahead it caught up (0.529 / 0.995 / 0.513 /
// 0.954), so this kernel is opt-in (DDPM3D_WZ2) until the pre-pass below exists.
//
// Known limit (DESIGN.md "next"): one wave per SIMD issues in order, so the staging VALU only
// overlaps the MFMAs if it is interleaved with them instruction by instruction.  Cutting the
// staging into per-tap branch-free pieces does that, but needs ~270 VGPRs beside the 256
// accumulators (48 raw-value registers, the weight ring, staging temporaries) and the
// allocator then evicts an accumulator set to scratch at every plane change: 2x slower.  The
// way out is to take the input transform out of this kernel (a pre-pass writes the
// transformed f16 hi/lo planes once per tensor instead of once per cout block and halo).
//
// Arithmetic per output element is IDENTICAL to conv3d_wz_kernel (same chunk/tap order, same
// three products per tap), so the two kernels agree bit for bit.
#pragma once
#include "conv3d_db.h"

template <int RING>
__global__ __launch_bounds__(256, 1) void conv3d_wz2_kernel(const ConvK p) {
    constexpr int CK = DDPM3D_CONV_CK, NT = 36;
    constexpr int TX = 8, TXL = 3, TYL = 3;
    constexpr int HX = 10, HY = 10, NPI = 6;         // NPI: raw input planes z0-1 .. z0+4
    constexpr int VS = 5;
    constexpr int RY = LdsGeom<TX, HX, HY>::RY;
    constexpr int RZ = LdsGeom<TX, HX, HY>::RZ;
    constexpr int IMG = 4 * RZ * 16;                 // one z-pair's transformed image (4 planes)
    constexpr int BUF = 2 * IMG;                     // both z-pairs
    constexpr int QPV = CK / 4;
    constexpr int HC = HX * HY * QPV;                // staging items: (y, x, channel quad) columns
    constexpr int NL = (HC + 255) / 256;
    constexpr int PF = RING - 1;                     // weight prefetch distance in taps
    static_assert(NT % RING == 0, "ring phase must survive the chunk boundary");
    static_assert(1 + 9 * NL < NT - PF, "staging must finish before the chunk ends");

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int zp = wv >> 1, nh = wv & 1;             // z-pair, cout half of this wave
    const int half = lane >> 5;

    const int tilesZ4 = p.tilesZ >> 1;               // p.tilesZ counts z-pairs; D % 4 == 0 here
    int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tx_i = tile % p.tilesX; tile /= p.tilesX;
    const int ty_i = tile % p.tilesY; tile /= p.tilesY;
    const int tz_i = tile % tilesZ4; tile /= tilesZ4;
    const int n = tile;
    const int x0 = tx_i * TX, y0 = ty_i * 8, z0 = tz_i * 4;

    int arow[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int m = t * 32 + (lane & 31);
        arow[t] = zp * IMG + ((m >> TXL) * RY + (m & (TX - 1)) * VS + half) * 16;
    }

    const int cout0 = blockIdx.y * 128 + nh * 64 + (lane & 31);   // + 32 for the second cout tile
    const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(p.w, p.w_bytes);
    const unsigned wlane = ((unsigned)cout0 * 2 + half) * 16;
    const unsigned wpart = (unsigned)p.CoutPad * 32;             // hi -> lo
    const unsigned wchunk_stride = 2 * wpart;
    const unsigned wtap_stride = (unsigned)(p.CinPad / CK) * wchunk_stride;
    constexpr unsigned WNT = 32 * 32;                             // 32 couts further on

    f32x16 acc[4][2][2];   // [transformed plane j][row tile][cout tile]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[j][t][u][i] = 0.0f;

    const int nchunks = p.CinPad / CK;
    const int chunk_begin = blockIdx.z * p.chunks_per_split;
    const int chunk_end = min(nchunks, chunk_begin + p.chunks_per_split);

    const int q = tid % QPV;
    const int up_shift = p.in_mode == DDPM3D_IN_UP ? 1 : 0;
    const unsigned act_mask = p.act ? 0xFFFFFFFFu : 0u;
    HaloSrc hs = halo_src<CK>(p, n, chunk_begin < chunk_end ? chunk_begin : 0, q);
    const int plane = hs.Hs * hs.Ws;
    int vox0[NL];          // source voxel of raw plane 1 (z = z0) at the item's (y, x), or -1
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int idx = tid + i * 256;
        const int hyx = idx / QPV;
        const int hy = hyx / HX, hx = hyx - hy * HX;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        const bool ok = idx < HC && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        vox0[i] = ok ? ((n * p.D + z0) * hs.Hs + (y >> up_shift)) * hs.Ws + (x >> up_shift) : -1;
    }
    f32x4 raw[NL][NPI];    // raw source values, then (in place) the normalised+activated d_k
    auto issue_raw = [&](const HaloSrc& h) {
        const __amdgpu_buffer_rsrc_t srsrc = make_rsrc(h.src, h.src_bytes);
        const unsigned row_bytes = (unsigned)h.Cs * 4, soff = (unsigned)h.cb * 4;
#pragma unroll
        for (int i = 0; i < NL; ++i)
#pragma unroll
            for (int k = 0; k < NPI; ++k) {
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
        for (int k = 0; k < NPI; ++k) {
            const bool inb = vox0[i] >= 0 && (unsigned)(z0 - 1 + k) < (unsigned)p.D;
            raw[i][k] = halo_finish<true>(hs, raw[i][k], inb, act_mask);
        }
    };
    // transformed plane j of z-pair s of item i: V_j, x8, f16 hi/lo split, into the image at `buf`
    auto store_j = [&](const int i, const int s, const int j, unsigned char* buf) {
        const int idx = tid + i * 256;
        if (idx < HC) {
            const int hyx = idx / QPV;
            const int hy = hyx / HX, hx = hyx - hy * HX;
            const int b = 2 * s;
            const f32x4 v = j == 0 ? raw[i][b] - raw[i][b + 2]
                          : j == 1 ? raw[i][b + 1] + raw[i][b + 2]
                          : j == 2 ? raw[i][b + 2] - raw[i][b + 1]
                                   : raw[i][b + 1] - raw[i][b + 3];
            h4 hi, lo;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float sc = fminf(fmaxf(v[c] * DDPM3D_X3_ACT_SCALE, -60000.0f), 60000.0f);
                hi[c] = (_Float16)sc;
                lo[c] = (_Float16)(sc - (float)hi[c]);
            }
            unsigned char* vrow = buf + s * IMG + (j * RZ + hy * RY + hx * VS) * 16;
            *reinterpret_cast<h4*>(vrow + q * 8) = hi;
            *reinterpret_cast<h4*>(vrow + 32 + q * 8) = lo;
        }
    };

    // weight ring: slot = tap % RING, [cout tile][hi|lo]
    u32x4 bq[RING][2][2];
    auto load_w = [&](const int slot, const unsigned off) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            bq[slot][u][0] = buffer_load16(wrsrc, wlane + u * WNT, off);
            bq[slot][u][1] = buffer_load16(wrsrc, wlane + u * WNT, off + wpart);
        }
    };

    if (chunk_begin < chunk_end) {
        issue_raw(hs);
#pragma unroll
        for (int s = 0; s < PF; ++s) load_w(s, (unsigned)chunk_begin * wchunk_stride + s * wtap_stride);
        // prologue: image 0 <- first chunk, raw <- second chunk
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            finish_d(i);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) store_j(i, s, j, lds);
        }
        if (chunk_begin + 1 < chunk_end) {
            hs = halo_src<CK>(p, n, chunk_begin + 1, q);
            issue_raw(hs);
        }
    }
    __syncthreads();

    int par = 0;
    for (int chunk = chunk_begin; chunk < chunk_end; ++chunk) {
        const bool more = chunk + 1 < chunk_end, more2 = chunk + 2 < chunk_end;
        const unsigned char* bufc = lds + par * BUF;
        unsigned char* bufn = lds + (par ^ 1) * BUF;
        const unsigned wchunk = (unsigned)chunk * wchunk_stride;

        h8 af[2][2][2];   // [slot][row tile][hi|lo]: A operands, read one tap ahead
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            af[0][t][0] = *reinterpret_cast<const h8*>(bufc + arow[t]);
            af[0][t][1] = *reinterpret_cast<const h8*>(bufc + arow[t] + 32);
        }
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
            // Keep each tap's instructions where they are written: left alone, the machine
            // scheduler sinks the weight loads of a straight-line run of taps down to their
            // use (to save registers) and the prefetch distance collapses to nothing.
            __builtin_amdgcn_sched_barrier(0);
            // weights PF taps ahead; past the chunk's end that is the next chunk's first taps
            // (beyond the packed image the buffer load returns zeros, never used)
            {
                const int nt = tap + PF;
                const unsigned off = nt < NT ? wchunk + nt * wtap_stride
                                             : wchunk + wchunk_stride + (nt - NT) * wtap_stride;
                load_w(nt % RING, off);
            }
            if (tap + 1 < NT) {
                const int t1 = tap + 1;
                const int off1 = ((t1 / 9) * RZ + ((t1 / 3) % 3) * RY + (t1 % 3) * VS) * 16;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    af[t1 & 1][t][0] = *reinterpret_cast<const h8*>(bufc + arow[t] + off1);
                    af[t1 & 1][t][1] = *reinterpret_cast<const h8*>(bufc + arow[t] + off1 + 32);
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // prefetches issue BEFORE this tap's MFMAs
            // staging of chunk+1 spread over the taps: item i finishes at tap 1 + 9i, its eight
            // transformed planes follow one per tap, into the other image
            if (more && tap >= 1 && tap < 1 + 9 * NL) {
                const int i = (tap - 1) / 9, ph = (tap - 1) % 9;
                if (ph == 0) finish_d(i);
                else store_j(i, (ph - 1) >> 2, (ph - 1) & 3, bufn);
            }
            if (tap == 1 + 9 * NL && more2) {
                hs = halo_src<CK>(p, n, chunk + 2, q);
                issue_raw(hs);
            }
            const int j = tap / 9;
            h8 bw[2][2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                bw[u][0] = __builtin_bit_cast(h8, bq[tap % RING][u][0]);
                bw[u][1] = __builtin_bit_cast(h8, bq[tap % RING][u][1]);
            }
            // three products per (row tile, cout tile), the four accumulators interleaved so a
            // dependent MFMA is three issues away
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    acc[j][t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[tap & 1][t][1], bw[u][0], acc[j][t][u], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    acc[j][t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[tap & 1][t][0], bw[u][1], acc[j][t][u], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    acc[j][t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[tap & 1][t][0], bw[u][0], acc[j][t][u], 0, 0, 0);
        }
        __syncthreads();   // image (par^1) complete, image par free for chunk+2's staging
        par ^= 1;
    }

    // ---- output transform (register-local), then the common epilogue: this wave's rows are
    // m = zp*128 + (zbit*2 + t)*32 + row  ->  (tz = zp*2 + zbit, ty, tx) of the 8x8x4 tile
    const int tile_in_n = (tz_i * p.tilesY + ty_i) * p.tilesX + tx_i;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        f32x16 outv[4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            outv[t] = acc[0][t][u] + acc[1][t][u] + acc[2][t][u];
            outv[2 + t] = acc[1][t][u] - acc[2][t][u] - acc[3][t][u];
        }
        conv_epilogue<1, 2, 4, TXL, TYL>(p, outv, n, z0, y0, x0, tile_in_n, zp, cout0 + 32 * u, half, blockIdx.z);
    }
}
