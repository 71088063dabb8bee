// 3x3x3 convolution with Winograd F(2,3) along the DEPTH axis, split-f16 arithmetic
// (DDPM3D_PREC_F16X3_WZ), for the layers that carry the network's FLOPs: 128-voxel tiles of one, two
// or four z-pairs (8x8x2, 8x4x4, 4x4x8: conv3d_stage.h WzGeomT), 128-cout workgroups with every wave
// active, pipelined inputs (IN_SAME / IN_UP).
//
// For one output pair (z, z+1) at a fixed (y, x) and one (dy, dx):
//     V0 = d0 - d2   V1 = d1 + d2   V2 = d2 - d1   V3 = d1 - d3        (d_k = input plane z-1+k)
//     U0 = g0        U1 = (g0+g1+g2)/2   U2 = (g0-g1+g2)/2   U3 = g2    (g_k = weight at dz = k)
//     M_j = sum over (ci, dy, dx) of U_j * V_j
//     out(z) = M0 + M1 + M2        out(z+1) = M1 - M2 - M3
// i.e. 4 products per two outputs instead of 6: 36 "taps" (j, dy, dx) per 16-channel chunk
// feeding four accumulator sets, 216 MFMAs per wave and chunk instead of 324.
//
// Why depth: D is never strided in this network (unet.py:129), the workgroup tile is whole
// z-pairs per (y, x), and a pair's four halo planes are exactly the transform's four inputs --
// so the transformed image V of a pair has the SAME LDS footprint as its plain halo image (4
// planes) and a tap is, as before, a compile-time LDS offset (plane j instead of plane dz).
// Consecutive pairs of a tile share two input planes (two new planes per pair).  The input transform is done in fp32 on the
// normalised+activated values while staging (before the f16 hi/lo split); the weight
// transform at pack time (ops.hip); the output transform is register-local in the epilogue
// (the four M_j of an output live in the same lane and register index).
//
// Staging is the lean form of conv3d_stage.h (r02).
//
// The four accumulator sets cost 128 VGPRs, so this kernel runs two workgroups per CU, whose
// stage / compute phases overlap each other.  (A variant with a double-buffered LDS image and
// the staging of chunk c+1 spread over the tap loop of chunk c measured 3-5 % slower, r01.)
#pragma once
#include "conv3d_stage.h"

// MODE WZ_F16X3: three f16 MFMAs per product on hi/lo-split operands (DDPM3D_PREC_F16X3_WZ);
// WZ_F16: one MFMA on the hi halves of the same packed image and LDS layout (DDPM3D_PREC_F16_WZ);
// WZ_BF16: one bf16 MFMA on bf16-rounded operands, no scaling (DDPM3D_PREC_BF16_WZ)
// IL: the next tap's LDS reads and weight loads are spread between this tap's MFMAs
// (sched_group_barrier) instead of issued in front of them
// Measurement build only (-DDDPM3D_WZ_STAMPS, tools/wz_stamps.py; never in the shipped library): s_memtime
// stamps of every phase of every wave, kept in LDS behind the image and dumped to ddpm3d_conv_desc.workspace
// when the wave ends: [workgroup][wave][WZ_NSTAMP] uint64.
#ifdef DDPM3D_WZ_STAMPS
#define WZ_NSTAMP 48
#define WZ_STAMP(i) do { if (lane == 0) wz_stamps[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WZ_STAMP(i) do { } while (0)
#endif

// Tile forms (conv3d_stage.h WzGeomT): 8 x 8 x 2 (one z-pair), 8 x 4 x 4 (two; r03), 4 x 4 x 8 (four: the levels
// below 8x8, which the direct kernel's 4x4 tiles ran at 1.5x the MFMAs; r03).  A 32-row MFMA tile is 32 positions of
// one transformed plane -- half a pair, one pair or two pairs' worth; the tap loop below is the same code for all.
template <int MODE, int IL = 0, int TX = 8, int TY = TX>
__global__ __launch_bounds__(256, 2) void conv3d_wz_kernel(const ConvK p) {
    typedef WzGeomT<TX, TY> G;
    constexpr bool X3 = MODE == WZ_F16X3;
    constexpr int CK = DDPM3D_CONV_CK, NT = 36;
    constexpr int TXL = TX == 8 ? 3 : 2, TYL = TY == 8 ? 3 : 2;
    constexpr int VS = G::VS, RY = G::RY, RZ = G::RZ;

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
#ifdef DDPM3D_WZ_STAMPS
    unsigned long long* wz_stamps = reinterpret_cast<unsigned long long*>(lds + G::BUF) + wn * WZ_NSTAMP;
    WZ_STAMP(0);
#endif

    const WgId wg = wg_id(p);
    int tile = wg.tile;
    const int tx_i = tile % p.tilesX; tile /= p.tilesX;
    const int ty_i = tile % p.tilesY; tile /= p.tilesY;
    const int tz_i = tile % p.tilesZ; tile /= p.tilesZ;
    const int n = tile;
    const int x0 = tx_i * TX, y0 = ty_i * TY, z0 = tz_i * 2 * G::NZP;

    // GEMM rows of this wave, two 32-row MFMA tiles: row m = position m % RP of z-pair m / RP
    int arow[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int m = t * 32 + (lane & 31);
        const int pos = m % G::RP;
        arow[t] = (m / G::RP) * G::PAIR + ((pos >> TXL) * RY + (pos & (TX - 1)) * VS + half) * 16;
    }

    const int cout = wg.cy * 128 + wn * 32 + (lane & 31);
    // the epilogue's per-cout scale and bias, requested now (two registers for the kernel's life)
    const float pre_ws = cout < p.Cout ? p.wscale[cout] : 1.0f;
    const float pre_bias = cout < p.Cout ? p.bias[(size_t)n * p.bias_stride_n + cout] : 0.0f;
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

    // the range bound is REQUESTED first and folded behind the first item's loads (r04): no address depends on
    // the scale, and a dependent round trip in front of the first loads is 3-4 % of a two-chunk workgroup's life
    // on the split levels (profiles/r04_wz_stamps_*: prologue 5.5 k cycles of 53 k at 512 -> 512 @ 64x4x4)
    ActScale asc = {1.0f, 1.0f};
    float bound_raw = 0.0f;
    if constexpr (MODE != WZ_BF16) bound_raw = act_scale_load(p, n);
    StageLaneT<G> sl = stage_lane<G>(p, tid, n, y0, x0, max(z0 - 1, 0), 1.0f);
    stage_zero_border(sl, lds, 1, tid);
    StageRawT<G> raw;
    if (chunk_begin < chunk_end) stage_issue(p, sl, raw, n, z0, chunk_begin);
    if constexpr (MODE != WZ_BF16) {
        asc = act_scale_finish(bound_raw, 2.0f);   // the input transform adds two planes
        stage_lane_scale(p, sl, asc.s);
    }
    WZ_STAMP(1);

    constexpr int L = X3 ? 1 : 0;   // index of the lo halves (unused slot 0 alias in the f16 form)
    // f16x3: ring of 4 taps (r04).  The r01-r03 ring of 3 "because the registers are full" overlooked that the ring empties
    // while the next chunk's raw staging registers fill (taps 33-35): with the staging loads pinned at tap NT - 3 a deeper
    // ring costs no register at the point of highest pressure (240 VGPRs at depth 4 on 8x4x4 tiles, no spill; 248 at 5;
    // the 4x4x8 form spills from 5).  Same box (profiles/r04_lib_ab_weight_ring_f16x3.txt): depth 3 / 4 / 5 / 6 =
    // 12.95 / 12.75 / 12.79 / 12.88 ms per forward; the 4x4x8 family, whose weights come from HBM, 0.842 -> 0.788.
#ifndef DDPM3D_WZ_RING_X3
#define DDPM3D_WZ_RING_X3 4
#endif
#ifndef DDPM3D_WZ_RING_X1
#define DDPM3D_WZ_RING_X1 8
#endif
    constexpr int R = X3 ? DDPM3D_WZ_RING_X3 : DDPM3D_WZ_RING_X1;
    // the next chunk's raw loads are issued behind the chunk's LAST weight loads (vmcnt retires in order): at tap NT - R
    // in the one-MFMA forms; a deeper f16x3 ring stops loading earlier, and its staging loads still go out at tap NT - 3,
    // when ring slots have started to free up (the raw registers take their place)
    constexpr int STAGE_TAP = X3 ? NT - 3 : NT - 8;
    for (int chunk = chunk_begin; chunk < chunk_end; ++chunk) {
        const bool more = chunk + 1 < chunk_end;
        const unsigned char* bufc = lds;
        WZ_STAMP(2 + (chunk - chunk_begin) * 5);
        __syncthreads();
        WZ_STAMP(3 + (chunk - chunk_begin) * 5);
        stage_write<MODE, G>(sl, raw, lds);
        WZ_STAMP(4 + (chunk - chunk_begin) * 5);
        __syncthreads();
        WZ_STAMP(5 + (chunk - chunk_begin) * 5);

        // ---- 36 taps (j, dy, dx); weight ring of R taps, prefetch distance R - 1: 3 in the f16x3 form
        // (r04, see R above; 2 until r03), 7 in the one-MFMA-per-product forms, whose taps are three
        // times shorter against the same L2 latency (r02, same box, bf16 DDIM-50: R = 3 / 4 / 6 / 8 / 9
        // -> 2.19 / 2.34 / 2.42 / 2.48 / 2.39 volumes/s; r04: 10 / 12 with the staging loads pinned at
        // tap 28: no better; the A operands one tap ahead as below: two ahead -1.5 %, three -16 %).
        // The stream's byte offset is one running scalar
        if constexpr (IL == 3 || IL == 4 || IL == 6) __builtin_amdgcn_s_setprio(1);
        unsigned woff = (unsigned)chunk * wchunk_stride;
        auto bump = [&]() {
            woff += wtap_stride;
            asm volatile("" : "+s"(woff));
        };
        u32x4 bq[R][X3 ? 2 : 1];
#pragma unroll
        for (int s = 0; s < R - 1; ++s) {
            bq[s][0] = buffer_load16(wrsrc, wlane, woff);
            if (X3) bq[s][L] = buffer_load16(wrsrc, wlane, woff + wpart);
            bump();
        }
#ifndef DDPM3D_WZ_A_AHEAD
#define DDPM3D_WZ_A_AHEAD 1     // taps the A operands are read ahead of their use (measurement: 2)
#endif
        constexpr int AH = DDPM3D_WZ_A_AHEAD, AS = AH + 1;
        h8 af[AS][2][X3 ? 2 : 1];   // [slot][row tile][hi|lo]: A operands, read AH tap(s) ahead
        auto aread = [&](const int t1) {
            const int off1 = ((t1 / 9) * RZ + ((t1 / 3) % 3) * RY + (t1 % 3) * VS) * 16;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                af[t1 % AS][t][0] = *reinterpret_cast<const h8*>(bufc + arow[t] + off1);
                if (X3) af[t1 % AS][t][L] = *reinterpret_cast<const h8*>(bufc + arow[t] + off1 + 32);
            }
        };
#pragma unroll
        for (int t1 = 0; t1 < AH; ++t1) aread(t1);
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
            // pin the tap boundary: the scheduler otherwise moves the LDS reads (and, in long
            // straight-line runs, the weight loads) down to their first use
            __builtin_amdgcn_sched_barrier(0);
            if (tap + AH < NT) aread(tap + AH);
            if (tap + R - 1 < NT) {
                bq[(tap + R - 1) % R][0] = buffer_load16(wrsrc, wlane, woff);
                if (X3) bq[(tap + R - 1) % R][L] = buffer_load16(wrsrc, wlane, woff + wpart);
                bump();
            }
            if constexpr (!IL) __builtin_amdgcn_sched_barrier(0);   // prefetches issue BEFORE this tap's MFMAs
            // next chunk's raw loads: after the chunk's last weight loads (vmcnt retires in order)
            if (tap == STAGE_TAP && more) stage_issue(p, sl, raw, n, z0, chunk + 1);
            const int j = tap / 9;
            const h8 bhi = __builtin_bit_cast(h8, bq[tap % R][0]);
            // per accumulator the order stays lo*hi, hi*lo, hi*hi; the two row tiles alternate
            if (X3) {
                const h8 blo = __builtin_bit_cast(h8, bq[tap % R][L]);
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    acc[j][t] = mfma16<false>(af[tap % AS][t][L], bhi, acc[j][t]);
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    acc[j][t] = mfma16<false>(af[tap % AS][t][0], blo, acc[j][t]);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
                acc[j][t] = mfma16<MODE == WZ_BF16>(af[tap % AS][t][0], bhi, acc[j][t]);
            if constexpr (IL == 1 || IL == 3) {
#pragma unroll
                for (int i = 0; i < (X3 ? 4 : 2); ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
#pragma unroll
                for (int i = 0; i < (X3 ? 2 : 1); ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
            } else if constexpr (IL == 5 || IL == 6) {
#pragma unroll
                for (int i = 0; i < (X3 ? 2 : 1); ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
#pragma unroll
                for (int i = 0; i < (X3 ? 2 : 1); ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
            } else if constexpr (IL == 2 || IL == 4) {
#pragma unroll
                for (int i = 0; i < (X3 ? 2 : 1); ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
#pragma unroll
                for (int i = 0; i < (X3 ? 4 : 2); ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
        }
        if constexpr (IL == 3 || IL == 4 || IL == 6) __builtin_amdgcn_s_setprio(0);
        WZ_STAMP(6 + (chunk - chunk_begin) * 5);
    }
    WZ_STAMP(42);

    // ---- output transform (register-local), then the common epilogue on the tile:
    // virtual accumulator u = zbit*2 + t covers rows m = u*32 + row: output plane zbit of z-pair (t*32 + row) / RP,
    // i.e. depth 2 * pair + zbit (conv_epilogue's ZPAIRS map; for 8x8x2 that is the plain tz = zbit)
    f32x16 outv[4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        outv[t] = acc[0][t] + acc[1][t] + acc[2][t];
        outv[2 + t] = acc[1][t] - acc[2][t] - acc[3][t];
    }
    const int tile_in_n = (tz_i * p.tilesY + ty_i) * p.tilesX + tx_i;
    #ifndef DDPM3D_WZ_WIDE_X3
#define DDPM3D_WZ_WIDE_X3 0     // measurement: the 16-byte epilogue in the f16x3 form too
#endif
    // a split launch leaves raw slabs for the reduce launch: 16-byte stores in every mode (r04; the f16x3 form keeps
    // its four-byte stores for finished outputs, whose epilogue hides behind 3x the MFMA work -- a split workgroup is
    // two to eight chunks long and nothing hides its slab stores: profiles/r04_tree_ab_wide_slabs_prologue_*.txt)
    // conv_epilogue_lean: the same arithmetic without conv_epilogue's run-time generality (false = not its case), in
    // the ONE-MFMA modes: bf16 forward -0.8 % on the same box.  The f16x3 kernels gain nothing from it (their epilogue
    // hides behind three times the MFMA work and the board's power cap: profiles/r04_lib_ab_lean_epilogue.txt) and keep
    // conv_epilogue.  (Both forms store through epi_store_b64 / _b128: see the store-data hazard note there.)
#ifndef DDPM3D_WZ_LEAN
#define DDPM3D_WZ_LEAN 1        // measurement: 0 = conv_epilogue everywhere
#endif
    constexpr bool LEAN = DDPM3D_WZ_LEAN && MODE != WZ_F16X3;
    const float ws1[1] = {pre_ws}, bs1[1] = {pre_bias};
    if (p.ksplit > 1) {
        if (LEAN && conv_epilogue_lean<1, 1, 4, 1, TXL, TYL, G::NZP != 1, true>(p, outv, n, z0, y0, x0, tile_in_n, 0, cout, half,
                                                                                         wg.split, asc.inv, ws1, bs1)) {
        } else
        conv_epilogue<1, 1, 4, TXL, TYL, true, G::NZP != 1, true>(p, outv, n, z0, y0, x0, tile_in_n, 0, cout, half, wg.split,
                                                                 asc.inv, true, pre_ws, pre_bias);
    } else if (LEAN &&
               conv_epilogue_lean<1, 1, 4, 1, TXL, TYL, G::NZP != 1, false>(p, outv, n, z0, y0, x0, tile_in_n, 0, cout, half, wg.split,
                                                                           asc.inv, ws1, bs1)) {
    } else
    conv_epilogue<1, 1, 4, TXL, TYL, MODE != WZ_F16X3 || DDPM3D_WZ_WIDE_X3, G::NZP != 1>(p, outv, n, z0, y0, x0, tile_in_n, 0, cout, half, wg.split,
                                                                 asc.inv, true, pre_ws, pre_bias);
#ifdef DDPM3D_WZ_STAMPS
    WZ_STAMP(43);
    if (lane == 0) {
        wz_stamps[44] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
        wz_stamps[45] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
        wz_stamps[46] = __builtin_amdgcn_s_memrealtime();
    }
    if (p.partial != nullptr && lane < WZ_NSTAMP) {
        const size_t wgl = blockIdx.x + (size_t)gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z);
        // a split launch keeps its slabs in front; the dump sits behind them (tools/wz_stamps.py sizes the workspace)
        float* dump = p.partial + (p.ksplit > 1 ? (size_t)p.ksplit * p.N * p.D * p.H * p.W * p.Cout : (size_t)0);
        reinterpret_cast<unsigned long long*>(dump)[(wgl * 4 + wn) * WZ_NSTAMP + lane] = wz_stamps[lane];
    }
#endif
}
