// Winograd-D 3x3x3 convolution (transform: conv3d_wz.h), WAVE-SPECIALISED, PERSISTENT form: a
// 512-thread workgroup whose waves 0-3 only compute and whose waves 4-7 only stage, walking
// p.ztiles consecutive z-pairs of one (y, x) tile column.
//
// Why.  In conv3d_wz_kernel every wave alternates  stage -> barrier -> 36 taps of MFMA, two
// workgroups per CU overlapping each other by luck.  PMC (r01, 128->128 @ 64^3): matrix pipe
// busy 62 %, waves parked at s_waitcnt / barrier 21 % -- both waves of a SIMD staging or
// parked at once, and weight waits queued behind the long halo loads because vmcnt retires in
// issue order.  So split the roles:
//   compute wave w (0-3): 32 couts x 64 rows x 4 transformed planes of accumulators, a weight
//       ring that runs across chunk AND tile boundaries, A operands read AHEAD taps ahead; never
//       issues a halo load, never executes staging VALU;
//   loader wave 4+w: while item i = (tile, chunk) is computed it normalises / activates /
//       transforms / splits item i+1 into the other LDS image and issues item i+2's raw loads;
//       never touches the matrix pipe.
// A workgroup's waves go to the SIMDs cyclically, so waves w and w+4 share SIMD w: every SIMD
// hosts exactly one compute and one loader wave, whose VALU co-issues with the other's MFMAs.
// One barrier per item hands an image over.  Registers are allocated per kernel (the compute
// role's ~220 VGPRs), i.e. two waves per SIMD, one workgroup (70 KB of LDS) per CU.
//
// Measured (r01, ms, 128->128 / 256->128 @ 64^3, same box): one tile per workgroup 0.486 /
// 0.892 (conv3d_wz_kernel: 0.473 / 0.892); loaders idle 0.441 / 0.793; 1/8 of the weight
// traffic 0.444 / 0.828; both 0.391 / 0.704 -- i.e. the steady state runs at 88 % of the MFMA
// floor but every workgroup pays ~9.8 us of prologue + epilogue that nothing overlaps when it
// is alone on its CU.  Hence the tile walk: the next tile's first chunk is staged during the
// current tile's last one.
//
// Arithmetic per output element is IDENTICAL to conv3d_wz_kernel (same chunk / tap / product
// order): the kernels agree bit for bit (tests/variant_conv.py).
#pragma once
#include "conv3d_db.h"

// X3: three f16 MFMAs per product on hi/lo-split operands (DDPM3D_PREC_F16X3_WZ); false: one MFMA
// on the hi halves of the same packed image and LDS layout (DDPM3D_PREC_F16_WZ)
template <int RING, int AHEAD, bool X3>
__global__ __launch_bounds__(512, 2) void conv3d_wzs_kernel(const ConvK p) {
    constexpr int CK = DDPM3D_CONV_CK, NT = 36;
    constexpr int TX = 8, TXL = 3, TYL = 3;
    constexpr int HX = 10, HY = 10, NP = 4;
    constexpr int VS = 5;
    constexpr int RY = LdsGeom<TX, HX, HY>::RY;
    constexpr int RZ = LdsGeom<TX, HX, HY>::RZ;
    constexpr int BUF = NP * RZ * 16;
    constexpr int QPV = CK / 4;
    constexpr int HC = HX * HY * QPV;
    constexpr int NL = (HC + 255) / 256;
    constexpr int PF = RING - 1;
    constexpr int AS = AHEAD + 1;                   // A-operand register slots
    static_assert(NT % RING == 0, "the weight ring's phase must survive the chunk boundary");

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 4;

    // tile column: (n, group of p.ztiles z-pairs, ty, tx)
    const int zgroups = (p.tilesZ + p.ztiles - 1) / p.ztiles;
    const WgId wg = wg_id(p);
    int tile = wg.tile;
    const int tx_i = tile % p.tilesX; tile /= p.tilesX;
    const int ty_i = tile % p.tilesY; tile /= p.tilesY;
    const int tg_i = tile % zgroups; tile /= zgroups;
    const int n = tile;
    const int x0 = tx_i * TX, y0 = ty_i * 8;
    const int tz_first = tg_i * p.ztiles;
    const int nzt = min(p.ztiles, p.tilesZ - tz_first);

    const int nchunks = p.CinPad / CK;
    const int chunk_begin = wg.split * p.chunks_per_split;
    const int chunk_end = min(nchunks, chunk_begin + p.chunks_per_split);
    const int nch = max(chunk_end - chunk_begin, 0);
    const int total = nzt * nch;                    // items (tile, chunk), tile-major

    if (loader) {
        // ------------------------------------------------------------------ staging role
        const int lt = tid - 256;
        const int q = lt % QPV;
        const int up_shift = p.in_mode == DDPM3D_IN_UP ? 1 : 0;
        const unsigned act_mask = p.act ? 0xFFFFFFFFu : 0u;
        HaloSrc hs = halo_src<CK>(p, n, nch > 0 ? chunk_begin : 0, q);
        const int plane = hs.Hs * hs.Ws;
        int vox0[NL];      // source voxel of the FIRST tile's input plane 1 (z = 2*tz_first), or -1
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int idx = lt + i * 256;
            const int hyx = idx / QPV;
            const int hy = hyx / HX, hx = hyx - hy * HX;
            const int y = y0 - 1 + hy, x = x0 - 1 + hx;
            const bool ok = idx < HC && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
            vox0[i] = ok ? ((n * p.D + 2 * tz_first) * hs.Hs + (y >> up_shift)) * hs.Ws + (x >> up_shift) : -1;
        }
        // Two raw-value register sets (A: even items, B: odd items), each with its HaloSrc: item
        // i+2's loads are issued BEFORE item i+1 is staged, so their memory latency runs beside a
        // whole item of staging instead of sitting on the loaders' critical path (with one set
        // the loaders took ~6k cycles per item, ~3k of it waiting -- as long as the 6.9k cycles of
        // MFMA they are supposed to hide behind).  The loader role has registers to spare.
        f32x4 rawA[NL][NP], rawB[NL][NP];
        HaloSrc hsB = hs;
        // item `it`'s raw loads into `raw`, addressed by its HaloSrc `hs`
        auto issue_raw = [&](f32x4 (&raw)[NL][NP], const HaloSrc& hs, const int it) {
            const int zi = it / nch;
            const int z0 = 2 * (tz_first + zi);
            const __amdgpu_buffer_rsrc_t srsrc = make_rsrc(hs.src, hs.src_bytes);
            const unsigned row_bytes = (unsigned)hs.Cs * 4, soff = (unsigned)hs.cb * 4;
#pragma unroll
            for (int i = 0; i < NL; ++i)
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const bool zok = (unsigned)(z0 - 1 + k) < (unsigned)p.D;
                    const unsigned voff = (vox0[i] < 0 || !zok)
                                              ? DDPM3D_OOB_OFFSET
                                              : (unsigned)(vox0[i] + (2 * zi + k - 1) * plane) * row_bytes + q * 16;
                    raw[i][k] = __builtin_bit_cast(f32x4, buffer_load16(srsrc, voff, soff));
                }
        };
        // raw -> d_k -> the four transformed planes V_j, x8, f16 hi/lo split, into the image at `buf`
        auto stage = [&](f32x4 (&raw)[NL][NP], const HaloSrc& hs, unsigned char* buf, const int it) {
            const int z0 = 2 * (tz_first + it / nch);
#pragma unroll
            for (int i = 0; i < NL; ++i) {
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const bool inb = vox0[i] >= 0 && (unsigned)(z0 - 1 + k) < (unsigned)p.D;
                    raw[i][k] = halo_finish<true>(hs, raw[i][k], inb, act_mask);
                }
                const int idx = lt + i * 256;
                if (idx < HC) {
                    const int hyx = idx / QPV;
                    const int hy = hyx / HX, hx = hyx - hy * HX;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
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
                        if (X3) *reinterpret_cast<h4*>(vrow + 32 + q * 8) = lo;
                    }
                }
            }
        };

        auto item_src = [&](const int it) { return halo_src<CK>(p, n, chunk_begin + it % nch, q); };
        if (total > 0) {
            issue_raw(rawA, hs, 0);
            if (total > 1) {
                hsB = item_src(1);
                issue_raw(rawB, hsB, 1);
            }
            stage(rawA, hs, lds, 0);
            if (total > 2) {
                hs = item_src(2);
                issue_raw(rawA, hs, 2);
            }
        }
        __syncthreads();                                   // image 0 ready
        // During item `it` (compute waves read image par): stage item it+1 from its set into
        // image par^1, then refill that set with item it+3.  Two items per trip so that the sets
        // are indexed statically.  Exactly `total` barriers, like the compute role.
        int par = 0;
        for (int it = 0; it < total; it += 2) {
            if (it + 1 < total) {
                stage(rawB, hsB, lds + (par ^ 1) * BUF, it + 1);
                if (it + 3 < total) {
                    hsB = item_src(it + 3);
                    issue_raw(rawB, hsB, it + 3);
                }
            }
            __syncthreads();                               // image par^1 ready, image par free
            par ^= 1;
            if (it + 1 >= total) break;
            if (it + 2 < total) {
                stage(rawA, hs, lds + (par ^ 1) * BUF, it + 2);
                if (it + 4 < total) {
                    hs = item_src(it + 4);
                    issue_raw(rawA, hs, it + 4);
                }
            }
            __syncthreads();
            par ^= 1;
        }
        return;
    }

    // ---------------------------------------------------------------------- compute role
    const int half = lane >> 5;
    int arow[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int m = t * 32 + (lane & 31);
        arow[t] = ((m >> TXL) * RY + (m & (TX - 1)) * VS + half) * 16;
    }
    const int cout = wg.cy * 128 + wave * 32 + (lane & 31);
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

    u32x4 bq[RING][X3 ? 2 : 1];  // weight ring: slot = tap % RING, [hi|lo]
    auto load_w = [&](const int slot, const unsigned off) {
        bq[slot][0] = buffer_load16(wrsrc, wlane, off);
        if (X3) bq[slot][X3 ? 1 : 0] = buffer_load16(wrsrc, wlane, off + wpart);
    };
    h8 af[AS][2][X3 ? 2 : 1];    // A operands: slot = tap % AS, [row tile][hi|lo]
    auto load_a = [&](const int slot, const unsigned char* base, const int tap) {
        const int off = ((tap / 9) * RZ + ((tap / 3) % 3) * RY + (tap % 3) * VS) * 16;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            af[slot][t][0] = *reinterpret_cast<const h8*>(base + arow[t] + off);
            if (X3) af[slot][t][X3 ? 1 : 0] = *reinterpret_cast<const h8*>(base + arow[t] + off + 32);
        }
    };

    if (total > 0) {
#pragma unroll
        for (int s = 0; s < PF; ++s) load_w(s, (unsigned)chunk_begin * wchunk_stride + s * wtap_stride);
    }
    __syncthreads();                                       // image 0 ready

    int par = 0;
    int cidx = 0;                                          // chunk index inside the tile
    int zi = 0;
    for (int it = 0; it < total; ++it) {
        const unsigned char* bufc = lds + par * BUF;
        const unsigned wchunk = (unsigned)(chunk_begin + cidx) * wchunk_stride;
        // the item after this one: next chunk of the tile, or the next tile's first chunk
        const unsigned wnext = (unsigned)(chunk_begin + (cidx + 1 == nch ? 0 : cidx + 1)) * wchunk_stride;
#pragma unroll
        for (int s = 0; s < AHEAD; ++s) load_a(s, bufc, s);
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
            // pin every tap's prefetches before its MFMAs (the scheduler otherwise sinks them)
            __builtin_amdgcn_sched_barrier(0);
            {
                // weights PF taps ahead; past the item's end those are the next item's first taps
                // (after the last item they are fetched and never used)
                const int nt = tap + PF;
                load_w(nt % RING, nt < NT ? wchunk + nt * wtap_stride : wnext + (nt - NT) * wtap_stride);
            }
            if (tap + AHEAD < NT) load_a((tap + AHEAD) % AS, bufc, tap + AHEAD);
            __builtin_amdgcn_sched_barrier(0);
            const int j = tap / 9;
            const h8 bhi = __builtin_bit_cast(h8, bq[tap % RING][0]);
            // per accumulator the order stays lo*hi, hi*lo, hi*hi; the two row tiles alternate
            if (X3) {
                const h8 blo = __builtin_bit_cast(h8, bq[tap % RING][X3 ? 1 : 0]);
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    acc[j][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[tap % AS][t][X3 ? 1 : 0], bhi, acc[j][t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    acc[j][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[tap % AS][t][0], blo, acc[j][t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
                acc[j][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[tap % AS][t][0], bhi, acc[j][t], 0, 0, 0);
        }
        __syncthreads();                                   // image par^1 ready, image par free
        par ^= 1;
        if (++cidx == nch) {
            // tile done: output transform (register-local) + the common epilogue, after the
            // barrier so that the loaders are already staging while this wave stores
            f32x16 outv[4];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                outv[t] = acc[0][t] + acc[1][t] + acc[2][t];
                outv[2 + t] = acc[1][t] - acc[2][t] - acc[3][t];
            }
            const int tz = tz_first + zi;
            const int tile_in_n = (tz * p.tilesY + ty_i) * p.tilesX + tx_i;
            conv_epilogue<1, 1, 4, TXL, TYL>(p, outv, n, 2 * tz, y0, x0, tile_in_n, 0, cout, half, wg.split);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[j][t][i] = 0.0f;
            cidx = 0;
            ++zi;
        }
    }
}
