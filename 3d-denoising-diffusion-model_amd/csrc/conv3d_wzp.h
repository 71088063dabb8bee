// Winograd-D 3x3x3 convolution, PLANE-PAIR form for the large layers (64^3 and 64x32x32 levels):
// one 512-thread workgroup per CU owns an 8x8x4 output tile (two z-pairs, 128 GEMM rows per
// transformed plane) x 128 couts and is split into two halves of four waves,
//     half 0: transformed planes j = 0, 1 (V0 = d0 - d2, V1 = d1 + d2)   -> 18 taps
//     half 1: transformed planes j = 2, 3 (V2 = d2 - d1, V3 = d1 - d3)   -> 18 taps
// each wave = 128 rows x 32 couts x 2 planes (128 accumulator registers, as in conv3d_wz.h).
//
// Why (r01/r02 counters): conv3d_wz_kernel's waves each stream their own weight
// fragments for 64 rows -- 2 KB per 6 MFMAs, 4.8 GB of L2 requests per 128->128 @ 64^3 launch,
// ~47 GB/s per CU, which is what a CU can pull from L2 at the clock this kernel runs at.  The
// MFMA pipes idle behind that stream.  Here a weight fragment serves 128 rows (2 KB per 12 MFMAs:
// half the stream) while the LDS traffic per MFMA and the accumulator count per wave stay the same:
// the two halves need DIFFERENT weights (U_j of their own planes) and different LDS planes, so
// nothing is fetched twice.  The four M_j of an output meet in the epilogue, through LDS.
//
// The halves run in ANTI-PHASE, one barrier per phase: while half 0 runs the taps of chunk c,
// half 1 stages its planes of chunk c; then half 1 runs its taps while half 0 stages chunk c+1.
// Each SIMD holds one wave of either half, so at any time exactly one wave per SIMD issues MFMAs
// and the other one's loads / VALU fill the issue slots in between -- what two co-resident
// workgroups of conv3d_wz_kernel do by chance, here by construction.
//
// Same arithmetic, value for value, as conv3d_wz_kernel (every accumulator sees its products in
// the same order; the output transform adds in the same order): the two forms are bit-identical
// (tests/test_gpu_ops.py), the launcher picks by shape (conv3d.hip).
#pragma once
#include "conv3d_stage.h"

// raw data of one staging slot of one half: five input planes z0 - 1 + HP .. z0 + 3 + HP
struct PairRaw {
    u32x4 v[5];
    unsigned zmask;
    bool b16;
};

// Slot I of the thread: issue its five plane loads.  Every scalar of the address is forced uniform
// (readfirstlane): the values are, but the compiler keeps some of them in VGPRs here and would wrap
// each load in a waterfall loop.
template <int HP, int I>
__device__ __forceinline__ void pair_issue(const ConvK& p, const StageLane& s, PairRaw& r, int z0, int chunk) {
    const int c0 = chunk * WzGeom::CK;
    const bool from0 = c0 < p.C0;
    const __amdgpu_buffer_rsrc_t rs = from0 ? make_rsrc(p.src0, p.src0_bytes) : make_rsrc(p.src1, p.src1_bytes);
    // (fields read into locals first: `c ? s.a : s.b` on members becomes a select of ADDRESSES and
    // then the whole StageLane lives in scratch)
    const unsigned pl0 = s.plane0, pl1 = s.plane1, e0 = s.es0, e1 = s.es1;
    const unsigned plane = __builtin_amdgcn_readfirstlane(from0 ? pl0 : pl1);
    const unsigned es = __builtin_amdgcn_readfirstlane(from0 ? e0 : e1);
    const unsigned cb4 = (unsigned)(from0 ? c0 : c0 - p.C0) * es;
    const int zb = __builtin_amdgcn_readfirstlane(s.zb);
    r.b16 = es == 2u;
    r.zmask = 0;
    const unsigned a = s.vo0[I], b = s.vo1[I];
    const unsigned vo = from0 ? a : b;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int z = z0 - 1 + HP + k;
        if ((unsigned)z < (unsigned)p.D) {
            r.zmask |= 1u << k;
            r.v[k] = buffer_load_quad(rs, vo, cb4 + (unsigned)(z - zb) * plane, r.b16);
        }
    }
}

// the chunk's GroupNorm/FiLM affine of the thread's channel quad, times the activation scale
struct PairAff { float sa[4], sb[4]; };
__device__ __forceinline__ PairAff pair_affine(const ConvK& p, const StageLane& s, int n, int chunk) {
    PairAff f;
    f32x4 ga = {1.f, 1.f, 1.f, 1.f}, gb = {0.f, 0.f, 0.f, 0.f};
    if (p.affA != nullptr) {
        ga = *reinterpret_cast<const f32x4*>(p.affA + (size_t)n * p.Cin + chunk * WzGeom::CK + s.q * 4);
        gb = *reinterpret_cast<const f32x4*>(p.affB + (size_t)n * p.Cin + chunk * WzGeom::CK + s.q * 4);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        f.sa[c] = ga[c] * s.scale;
        f.sb[c] = gb[c] * s.scale;
    }
    return f;
}

// slot I: raw -> the half's image, planes (z-pair zp, jj) at img + (zp * 2 + jj) * RZ * 16
template <int MODE, int HP, int I>
__device__ __forceinline__ void pair_write(const StageLane& s, const PairRaw& r, const PairAff& f, unsigned char* img) {
    float e[5][4];   // S * act(A x + B) of planes z0 - 1 + HP + k
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        if (r.zmask & (1u << k)) {
            const f32x4 x = quad_bits_expand(r.v[k], r.b16);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float ys = __builtin_fmaf(x[c], f.sa[c], f.sb[c]);
                const float ex = __builtin_amdgcn_exp2f(__builtin_fmaf(ys, s.km, s.ka));
                e[k][c] = ys * __builtin_amdgcn_rcpf(1.0f + ex);
            }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) e[k][c] = 0.0f;
        }
    }
    if (s.ok[I]) {
#pragma unroll
        for (int zp = 0; zp < 2; ++zp)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int b = 2 * zp;
                float v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    v[c] = HP == 0 ? (jj == 0 ? e[b][c] - e[b + 2][c] : e[b + 1][c] + e[b + 2][c])
                                   : (jj == 0 ? e[b + 1][c] - e[b][c] : e[b][c] - e[b + 2][c]);
                unsigned char* vrow = img + (zp * 2 + jj) * WzGeom::RZ * 16 + s.lds[I];
                if constexpr (MODE == WZ_BF16) {
                    *reinterpret_cast<u32x2*>(vrow) = u32x2{bf16_pack(v[0], v[1]), bf16_pack(v[2], v[3])};
                } else {
                    unsigned h0, h1, l0, l1;
                    split_pair(v[0], v[1], h0, l0);
                    split_pair(v[2], v[3], h1, l1);
                    *reinterpret_cast<u32x2*>(vrow) = u32x2{h0, h1};
                    if (MODE == WZ_F16X3) *reinterpret_cast<u32x2*>(vrow + 32) = u32x2{l0, l1};
                }
            }
    }
}

// weight stream of one wave: uniform descriptor + lane offset + running scalar offset
struct WzpW {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned wlane, wpart, wtap_stride, woff;
};

// the first two weight fragments of a chunk: issued BEFORE the barrier that opens its tap phase
template <bool X3, int NB, int R>
__device__ __forceinline__ void wzp_weights_head(WzpW& w, unsigned start, u32x4 (&bq)[R][NB]) {
    w.woff = start;
#pragma unroll
    for (int s = 0; s < R - 1; ++s) {
        bq[s][0] = buffer_load16(w.rsrc, w.wlane, w.woff);
        if (X3) bq[s][NB - 1] = buffer_load16(w.rsrc, w.wlane, w.woff + w.wpart);
        w.woff += w.wtap_stride;
    }
}

// the 18 taps (jj, dy, dx) of one chunk: 4 row tiles x (3 | 1) MFMAs each
template <int MODE, int NB, int DBG = 0>
__device__ __forceinline__ void wzp_taps(const unsigned char* img, const int (&arow)[4], WzpW& w,
                                         u32x4 (&bq)[(DBG & 8) ? 4 : 3][NB], f32x16 (&acc)[2][4]) {
    constexpr int R = (DBG & 8) ? 4 : 3;   // weight ring: prefetch distance R - 1 taps
    constexpr bool X3 = MODE == WZ_F16X3;
    constexpr int NT = 18, L = NB - 1;
    constexpr int VS = WzGeom::VS, RY = WzGeom::RY, RZ = WzGeom::RZ;
    h8 af[2][4][NB];   // [slot][row tile][hi|lo]
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        af[0][t][0] = *reinterpret_cast<const h8*>(img + arow[t]);
        if (X3) af[0][t][L] = *reinterpret_cast<const h8*>(img + arow[t] + 32);
    }
#pragma unroll
    for (int tap = 0; tap < NT; ++tap) {
        __builtin_amdgcn_sched_barrier(0);
        if (tap + 1 < NT) {
            const int t1 = tap + 1;
            const int off1 = ((t1 / 9) * RZ + ((t1 / 3) % 3) * RY + (t1 % 3) * VS) * 16;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if constexpr (DBG & 16) {   // measurement: no LDS reads behind tap 0
                    af[t1 & 1][t][0] = af[tap & 1][t][0];
                    if (X3) af[t1 & 1][t][L] = af[tap & 1][t][L];
                    continue;
                }
                af[t1 & 1][t][0] = *reinterpret_cast<const h8*>(img + arow[t] + off1);
                if (X3) af[t1 & 1][t][L] = *reinterpret_cast<const h8*>(img + arow[t] + off1 + 32);
            }
        }
        if constexpr (DBG & 32) {           // measurement: no weight loads behind the head
            if (tap + R - 1 < NT) {
                bq[(tap + R - 1) % R][0] = bq[tap % R][0];
                if (X3) bq[(tap + R - 1) % R][L] = bq[tap % R][L];
            }
        } else
        if (tap + R - 1 < NT) {
            bq[(tap + R - 1) % R][0] = buffer_load16(w.rsrc, w.wlane, w.woff);
            if (X3) bq[(tap + R - 1) % R][L] = buffer_load16(w.rsrc, w.wlane, w.woff + w.wpart);
            w.woff += w.wtap_stride;
        }
        if constexpr (!(DBG & 4)) __builtin_amdgcn_sched_barrier(0);
        const int jj = tap / 9;
        const h8 bhi = __builtin_bit_cast(h8, bq[tap % R][0]);
        if (X3) {
            const h8 blo = __builtin_bit_cast(h8, bq[tap % R][L]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[jj][t] = mfma16<false>(af[tap & 1][t][L], bhi, acc[jj][t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[jj][t] = mfma16<false>(af[tap & 1][t][0], blo, acc[jj][t]);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[jj][t] = mfma16<MODE == WZ_BF16>(af[tap & 1][t][0], bhi, acc[jj][t]);
        if constexpr (DBG & 4) {
        // one wave per SIMD issues the MFMAs: spread the next tap's LDS reads and the weight
        // loads between them instead of in a clump in front (the pipe would drain behind it)
#pragma unroll
        for (int i = 0; i < (X3 ? 8 : 4); ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
        }
#pragma unroll
        for (int i = 0; i < (X3 ? 2 : 1); ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
        }
        }
    }
}

// One staging item of a half.  The two slots of a thread go one after the other (loads of slot 1
// behind the stores of slot 0): the phase has time to spare (the other half's taps take twice as
// long) and 20 raw registers instead of 40 keep the 128 accumulators out of scratch.
template <int MODE, int HP>
__device__ __forceinline__ void wzp_stage(const ConvK& p, const StageLane& sl, unsigned char* img, int n, int z0,
                                          int chunk) {
    static_assert(WzGeom::NL == 2, "two staging slots per thread");
    const PairAff f = pair_affine(p, sl, n, chunk);
    {
        PairRaw raw;
        pair_issue<HP, 0>(p, sl, raw, z0, chunk);
        pair_write<MODE, HP, 0>(sl, raw, f, img);
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        PairRaw raw;
        pair_issue<HP, 1>(p, sl, raw, z0, chunk);
        pair_write<MODE, HP, 1>(sl, raw, f, img);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// One half's life: staging and tap phases of every chunk, in the phase order of its side.
//   half 0:  [stage 0] | taps 0 | stage 1 | taps 1 | ...          ("|" = workgroup barrier)
//   half 1:  [  idle ] | stage 0 | taps 0 | stage 1 | ...
template <int MODE, int HP, int DBG = 0>
__device__ __forceinline__ void wzp_half(const ConvK& p, unsigned char* img, const StageLane& sl, int n, int z0,
                                         const int (&arow)[4], unsigned wlane, f32x16 (&acc)[2][4]) {
    constexpr bool X3 = MODE == WZ_F16X3;
    constexpr int CK = DDPM3D_CONV_CK, NB = X3 ? 2 : 1;
    WzpW w;
    w.rsrc = make_rsrc(p.w, p.w_bytes);
    w.wlane = wlane;
    w.wpart = (unsigned)p.CoutPad * 32;
    const unsigned wchunk_stride = 2 * w.wpart;
    w.wtap_stride = (unsigned)(p.CinPad / CK) * wchunk_stride;
    w.woff = 0;
    const unsigned wbase = (unsigned)(HP * 18) * w.wtap_stride;   // this half's taps in the packed image
    const int nchunks = p.CinPad / CK;
    constexpr int R = (DBG & 8) ? 4 : 3;
    u32x4 bq[R][NB];

    // Straight-line loop bodies (taps and staging both unconditional, like conv3d_wz.h): with the
    // two behind a phase-parity branch the accumulators' live ranges split at every merge and half
    // of them ended up in scratch.
    if constexpr (HP == 0) {
        wzp_stage<MODE, HP>(p, sl, img, n, z0, 0);
        wzp_weights_head<X3, NB, R>(w, wbase, bq);
        __syncthreads();
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            if constexpr (!(DBG & 2)) wzp_taps<MODE, NB, DBG>(img, arow, w, bq, acc);
            __syncthreads();
            // behind the last chunk this restages it (nothing reads it): keeps the body branch-free
            const int next = min(chunk + 1, nchunks - 1);
            if (!(DBG & 1)) wzp_stage<MODE, HP>(p, sl, img, n, z0, next);
            wzp_weights_head<X3, NB, R>(w, (unsigned)next * wchunk_stride + wbase, bq);
            __syncthreads();
        }
    } else {
        __syncthreads();
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            if (!(DBG & 1) || chunk == 0) wzp_stage<MODE, HP>(p, sl, img, n, z0, chunk);
            wzp_weights_head<X3, NB, R>(w, (unsigned)chunk * wchunk_stride + wbase, bq);
            __syncthreads();
            if constexpr (!(DBG & 2)) wzp_taps<MODE, NB, DBG>(img, arow, w, bq, acc);
            __syncthreads();
        }
    }
}

// DBG (measurement builds only, -DDDPM3D_WZP_DEBUG): 1 = stage only the first chunk, 2 = no taps,
// 4 = LDS reads / weight loads interleaved with the MFMAs by sched_group_barrier
template <int MODE, int DBG = 0>
__global__ __launch_bounds__(512, 2) void conv3d_wzp_kernel(const ConvK p) {
    constexpr int TX = 8, TXL = 3, TYL = 3;
    constexpr int VS = WzGeom::VS, RY = WzGeom::RY;

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hp = wave >> 2, wn = wave & 3;
    const int half = lane >> 5;
    const int lt = tid & 255;   // staging thread inside the half

    const int tilesZ = p.D >> 2;
    const WgId wg = wg_id(p);
    int tile = wg.tile;
    const int tx_i = tile % p.tilesX; tile /= p.tilesX;
    const int ty_i = tile % p.tilesY; tile /= p.tilesY;
    const int tz_i = tile % tilesZ; tile /= tilesZ;
    const int n = tile;
    const int x0 = tx_i * TX, y0 = ty_i * 8, z0 = tz_i * 4;

    unsigned char* img = lds + hp * WzGeom::BUF;   // this half's image: (z-pair, jj) planes

    // GEMM rows of a wave: row tile t = (z-pair t >> 1, 32 of the 64 (y, x) positions)
    int arow[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int m = (t & 1) * 32 + (lane & 31);
        arow[t] = (t >> 1) * 2 * WzGeom::RZ * 16 + ((m >> TXL) * RY + (m & (TX - 1)) * VS + half) * 16;
    }
    const int cout = wg.cy * 128 + wn * 32 + (lane & 31);
    const unsigned wlane = ((unsigned)cout * 2 + half) * 16;

    f32x16 acc[2][4];   // [plane jj of the half][row tile]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[j][t][i] = 0.0f;

    ActScale asc = {1.0f, 1.0f};
    if constexpr (MODE != WZ_BF16) asc = act_scale(p, n, 2.0f);
    const StageLane sl = stage_lane(p, lt, n, y0, x0, max(z0 - 1, 0), asc.s);
    stage_zero_border(sl, img, 1, lt);

    if (hp == 0)
        wzp_half<MODE, 0, DBG>(p, img, sl, n, z0, arow, wlane, acc);
    else
        wzp_half<MODE, 1, DBG>(p, img, sl, n, z0, arow, wlane, acc);

    // ---- the halves swap one plane through LDS (the images are dead: last barrier passed):
    //   half 0 holds M0, M1 and writes out(z)   = (M0 + M1) + M2   for z = z0, z0 + 2
    //   half 1 holds M2, M3 and writes out(z+1) = (M1 - M2) - M3   for z = z0 + 1, z0 + 3
    float* xch = reinterpret_cast<float*>(lds);
    {
        float* mine = xch + wave * 4096 + lane;
        // half 0 gives M1, half 1 gives M2 (a uniform branch: a runtime plane index would put the
        // accumulators in scratch)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float m1 = acc[1][t][i], m2 = acc[0][t][i];
                mine[(t * 16 + i) * 64] = hp == 0 ? m1 : m2;     // (values, not a select of addresses)
            }
        }
    }
    __syncthreads();
    f32x16 outv[4];
    {
        const float* theirs = xch + (wave ^ 4) * 4096 + lane;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x16 x;
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = theirs[(t * 16 + i) * 64];
            if (hp == 0)
                outv[t] = acc[0][t] + acc[1][t] + x;    // (M0 + M1) + M2
            else
                outv[t] = x - acc[0][t] - acc[1][t];    // (M1 - M2) - M3
        }
    }
    // statistics row: one per (8x8 column, output plane pair) -- (D/2) * tilesY * tilesX rows per sample
    const int row = ((tz_i * 2 + hp) * p.tilesY + ty_i) * p.tilesX + tx_i;
    if constexpr (DBG & 64) {   // measurement: no epilogue (one store keeps the accumulators alive)
        float sink = 0.0f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) sink += outv[t][i];
        if (sink == 12345.678f) p.out[0] = sink;
        return;
    }
    conv_epilogue<1, 1, 4, TXL, TYL, 2>(p, outv, n, z0 + hp, y0, x0, row, 0, cout, half, 0, asc.inv);
}
