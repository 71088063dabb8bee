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
//   B operand      packed weights streamed L2 -> VGPR (buffer_load_dwordx4),
//                  private to the wave's 32 couts, 3-tap register ring
//   split-K        blockIdx.z owns a range of the ci-chunks (low-res levels)
//
// This file: the direct form (all shapes and input modes), the split-K reduce and the
// dispatcher.  The 3x3x3 layers that carry the network's FLOPs run the Winograd-along-depth
// form instead (conv3d_wz.h, staging in conv3d_stage.h).
//
// Arithmetic modes (template PREC), same fp32 inputs, outputs and
// accumulators:
//   PREC 0  v_mfma_f32_32x32x2_f32: exact fp32 products (bitwise an fmaf chain),
//           64 FLOP/clk/SIMD.
//   PREC 1  each fp32 operand is split x = hi + lo into two f16 (after a
//           power-of-two scaling that keeps lo out of the f16 subnormals) and
//           a*b is evaluated as hi*hi + hi*lo + lo*hi on
//           v_mfma_f32_32x32x16_f16 (every f16 x f16 product is exact in fp32).
//           Operand representation error ~2^-23 relative -- below the fp32
//           accumulation error both modes share -- at 16/3 the MFMA rate.
//   PREC 2  one product on the hi halves only (operands rounded to f16): the reference's
//           --use_fp16 analogue.
//   PREC 5  one bf16 MFMA per product (operands rounded to bf16, no scaling): BASELINE config 4.
//
// K order inside a channel block is permuted in PREC 0 (lane half h supplies
// channel 4h+s at step s) so both operands are 16-byte loads; the weight packer
// (ops.hip) stores the order each mode reads.
#include <hip/hip_runtime.h>
#include "conv3d_load.h"
#include "conv3d_epilogue.h"  // LdsGeom, conv_epilogue

// PIPEM = 1: IN_SAME / IN_UP inputs, staging software-pipelined; 0: IN_POOL / IN_PLANAR2;
// 2: IN_STRIDE2 (stride (1,2,2): the halo tile is laid out in the source grid, rows 2 voxels apart).
template <int PREC, int PIPEM, int KS, int WN, int MT, int TXL, int TYL>
__global__ __launch_bounds__(256, (PIPEM == 2 ? 1 : (TXL == 2 ? 2 : 3))) void conv3d_kernel(const ConvK p) {
    constexpr int CK = DDPM3D_CONV_CK;
    constexpr int PIPE = PIPEM == 1 ? 1 : 0;
    constexpr int SXY = PIPEM == 2 ? 2 : 1;           // stride along H and W
    constexpr int WM = 4 / WN;
    // MT = 32-row accumulators per wave; workgroup tile = WM * MT * 32 voxels = 128
    constexpr int TX = 1 << TXL, TY = 1 << TYL;
    constexpr int TZ = WM * MT * 32 / (TX * TY);
    constexpr int PAD = KS / 2;
    constexpr int HX = SXY * (TX - 1) + KS, HY = SXY * (TY - 1) + KS, HZ = TZ + 2 * PAD;
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
    constexpr int NL = (HV * QPV + 255) / 256;                  // staging items per thread

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5;

    const WgId wg = wg_id(p);
    int tile = wg.tile;
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
        arow[t] = (tz * RZ + SXY * ty * RY + SXY * tx * VS + half) * 16;
    }

    const int cout = wg.cy * (32 * WN) + wn * 32 + (lane & 31);
    const bool wave_active = (wg.cy * (32 * WN) + wn * 32) < p.CoutPad;
    const int cout_ld = wave_active ? cout : 0;
    // Weight stream in 16-byte units.
    //   PREC 0: [tap][ci/8][CoutPad][8 f32]            -> 2 units per (8-ci block, cout)
    //   PREC 1: [tap][ci/16][hi|lo][CoutPad][16 f16]   -> 2 units per (16-ci block, part, cout)
    // Addressed as buffer loads: uniform descriptor + ONE per-lane byte offset + a scalar
    // (chunk, tap, part) byte offset, so nothing per-tap lives in VGPRs.
    const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(p.w, p.w_bytes);
    const unsigned wlane = ((unsigned)cout_ld * 2 + half) * 16;
    const unsigned wpart = (unsigned)p.CoutPad * 32;                  // PREC0: next 8-ci block; PREC1: hi -> lo
    const unsigned wchunk_stride = 2 * wpart;                         // one 16-ci chunk
    const unsigned wtap_stride = (unsigned)(p.CinPad / CK) * wchunk_stride;

    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;

    const int nchunks = p.CinPad / CK;
    // split-K: blockIdx.z owns a contiguous range of the Cin chunks
    const int chunk_begin = wg.split * p.chunks_per_split;
    const int chunk_end = min(nchunks, chunk_begin + p.chunks_per_split);

    // Software pipeline of the staging (IN_SAME / IN_UP, one 16-byte load per item): the
    // raw loads of chunk c+1 are issued near the end of chunk c's tap loop and finished
    // (affine + activation + LDS store) after the barrier.  vmcnt retires in issue order,
    // so they are issued only AFTER the chunk's last weight loads: a weight wait must never
    // sit behind the long-latency halo loads.
    const int q = tid % QPV;  // fixed per thread: 256 % QPV == 0
    const int up_shift = p.in_mode == DDPM3D_IN_UP ? 1 : 0;
    const unsigned act_mask = p.act ? 0xFFFFFFFFu : 0u;
    u32x4 raw[PIPE ? NL : 1];   // raw bits of the prefetched quads (fp32, or bf16 in the low half)
    ActScale asc = {1.0f, 1.0f};
    if constexpr (PREC == 1 || PREC == 2) asc = act_scale(p, n, 1.0f);
    HaloSrc hs = halo_src<CK>(p, n, chunk_begin < chunk_end ? chunk_begin : 0, q);
    // source voxel of each staging item (-1 = zero padding): launch-invariant, one VGPR each
    int vox[PIPE ? NL : 1];
#pragma unroll
    for (int i = 0; i < (PIPE ? NL : 0); ++i) {
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
        const __amdgpu_buffer_rsrc_t srsrc = make_rsrc(h.src, h.src_bytes);  // uniform: src0 or src1
        const unsigned es = h.b16 ? 2u : 4u;                                 // bytes per element
        const unsigned row_bytes = (unsigned)h.Cs * es, soff = (unsigned)h.cb * es;
#pragma unroll
        for (int i = 0; i < (PIPE ? NL : 0); ++i) {
            // out-of-range offset -> the buffer load returns 0 = the conv's zero padding
            const unsigned voff = vox[i] < 0 ? DDPM3D_OOB_OFFSET : (unsigned)vox[i] * row_bytes + q * 4 * es;
            raw[i] = buffer_load_quad(srsrc, voff, soff, h.b16);
        }
    };
    if (PIPE && chunk_begin < chunk_end) issue_raw(hs);

    // (Tried, r01, for the 1x1 convs whose single tap leaves the weight ring nothing to run ahead
    // of: (a) the tap fetched a whole chunk ahead in loop-carried registers -- hipcc then waits
    // vmcnt(0) at the top of the tap section, ahead of the new loads: 0.131 -> 0.318 ms on 256->128
    // @ 64^3; (b) the tap requested before the staging barriers: 0.128 -> 0.135 ms.  With three
    // workgroups per CU the latency is already covered; those layers just move 400 MB.)
    for (int chunk = chunk_begin; chunk < chunk_end; ++chunk) {
        __syncthreads();  // everyone done reading the previous chunk's tile
        // ------------------------------------------------ stage the halo tile
        {
            auto store_item = [&](int hz, int hy, int hx, const f32x4 v) {
                unsigned char* vrow = lds + (hz * RZ + hy * RY + hx * VS) * 16;
                if constexpr (PREC == 0) {
                    *reinterpret_cast<f32x4*>(vrow + q * 16) = v;
                } else if constexpr (PREC == 5) {
                    *reinterpret_cast<u32x2*>(vrow + q * 8) = u32x2{bf16_pack(v[0], v[1]), bf16_pack(v[2], v[3])};
                } else {
                    h4 hi, lo;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float s = v[c] * asc.s;   // |s| < 2^15 by the choice of the scale
                        hi[c] = (_Float16)s;
                        lo[c] = (_Float16)(s - (float)hi[c]);
                    }
                    *reinterpret_cast<h4*>(vrow + q * 8) = hi;
                    if constexpr (PREC == 1) *reinterpret_cast<h4*>(vrow + 32 + q * 8) = lo;
                }
            };
            if constexpr (PIPE) {
                // finish the prefetched items (no loads here: unrolled for static raw[] indexing)
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    const int idx = tid + i * 256;
                    if (idx < HV * QPV) {
                        const int hv = idx / QPV;
                        const int hz = hv / (HY * HX);
                        const int rem = hv - hz * (HY * HX);
                        const int hy = rem / HX;
                        const int hx = rem - hy * HX;
                        const bool inb = halo_inb(p, z0 - PAD + hz, y0 - PAD + hy, x0 - PAD + hx);
                        store_item(hz, hy, hx, halo_finish<PREC != 0>(hs, quad_bits_expand(raw[i], hs.b16, hs.f16), inb, act_mask));
                    }
                }
            } else {
                // pool / planar inputs: fetch + transform in place (rolled loop: an item is up to 4 loads)
#pragma unroll 1
                for (int idx = tid; idx < HV * QPV; idx += 256) {
                    const int hv = idx / QPV;
                    const int hz = hv / (HY * HX);
                    const int rem = hv - hz * (HY * HX);
                    const int hy = rem / HX;
                    const int hx = rem - hy * HX;
                    store_item(hz, hy, hx, halo_fetch<PREC != 0>(p, hs, n, z0 - PAD + hz, SXY * y0 - PAD + hy,
                                                                 SXY * x0 - PAD + hx, q, chunk == 0));
                }
            }
        }
        __syncthreads();
        const bool more = chunk + 1 < chunk_end;
        if (more) hs = halo_src<CK>(p, n, chunk + 1, q);  // also loads the next chunk's affine quad
        if (!wave_active) {  // (only with a cout tile beyond CoutPad) still owns staging items
            if (PIPE && more) issue_raw(hs);
            continue;
        }

        // ------------------------------------------------ taps x k-steps
        // weight ring of R taps (prefetch distance R - 1), statically indexed by the unrolled tap
        const unsigned wchunk = (unsigned)chunk * wchunk_stride;  // scalar byte offset of this chunk
        constexpr bool LO = PREC != 2 && PREC != 5;  // PREC 2 / 5 stream only the hi halves
        constexpr int R = 3;   // (deeper: no gain in the f16x3 form, -3 % on one family in the bf16 form)
        u32x4 bq[R][2];
#pragma unroll
        for (int s = 0; s < R - 1 && s < NT; ++s) {
            bq[s][0] = buffer_load16(wrsrc, wlane, wchunk + s * wtap_stride);
            if (LO) bq[s][1] = buffer_load16(wrsrc, wlane, wchunk + s * wtap_stride + wpart);
        }
        // split-f16 modes, 4x4 tiles and 1x1 convs: A operands read one tap ahead into a second register
        // set (as in conv3d_wz.h), the tap boundary pinned so the scheduler does not sink the reads to
        // their use: -19 % on the 4x4-level 3x3x3 convs, -2 % on the 1x1 convs (r02, same box).  Not on
        // the 8x8-tile 3x3x3 form: three workgroups per CU leave 170 registers, it spills (+4 %).
        constexpr bool AHEAD = PREC != 0 && (TXL == 2 || KS == 1);
        h8 af[2][MT][2];
        if constexpr (AHEAD) {
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                af[0][t][0] = *reinterpret_cast<const h8*>(lds + arow[t]);
                if (LO) af[0][t][1] = *reinterpret_cast<const h8*>(lds + arow[t] + 32);
            }
        }
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
            if constexpr (AHEAD) {
                __builtin_amdgcn_sched_barrier(0);
                if (tap + 1 < NT) {
                    const int t1 = tap + 1;
                    const int off1 = ((t1 / (KS * KS)) * RZ + ((t1 / KS) % KS) * RY + (t1 % KS) * VS) * 16;
#pragma unroll
                    for (int t = 0; t < MT; ++t) {
                        af[t1 & 1][t][0] = *reinterpret_cast<const h8*>(lds + arow[t] + off1);
                        if (LO) af[t1 & 1][t][1] = *reinterpret_cast<const h8*>(lds + arow[t] + off1 + 32);
                    }
                }
            }
            if (tap + R - 1 < NT) {
                bq[(tap + R - 1) % R][0] = buffer_load16(wrsrc, wlane, wchunk + (tap + R - 1) * wtap_stride);
                if (LO) bq[(tap + R - 1) % R][1] = buffer_load16(wrsrc, wlane, wchunk + (tap + R - 1) * wtap_stride + wpart);
            }
            if constexpr (AHEAD) __builtin_amdgcn_sched_barrier(0);
            // all of this chunk's weight loads are in flight: now the next chunk's halo loads
            if (PIPE && tap == (NT >= R ? NT - R : 0) && more) issue_raw(hs);
            const int dz = tap / (KS * KS), dy = (tap / KS) % KS, dx = tap % KS;
            const int tapoff = (dz * RZ + dy * RY + dx * VS) * 16;
            const u32x4 b0 = bq[tap % R][0], b1 = LO ? bq[tap % R][1] : b0;
            if constexpr (PREC == 0) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    f32x4 a[MT];
#pragma unroll
                    for (int t = 0; t < MT; ++t)
                        a[t] = *reinterpret_cast<const f32x4*>(lds + arow[t] + tapoff + kk * 32);
                    const f32x4 b = __builtin_bit_cast(f32x4, kk == 0 ? b0 : b1);
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int t = 0; t < MT; ++t)
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][s], b[s], acc[t], 0, 0, 0);
                }
            } else if constexpr (PREC == 2 || PREC == 5) {
                // single product on the f16- / bf16-rounded operands
                const h8 bhi = __builtin_bit_cast(h8, b0);
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const h8 ahi = AHEAD ? af[tap & 1][t][0] : *reinterpret_cast<const h8*>(lds + arow[t] + tapoff);
                    acc[t] = mfma16<PREC == 5>(ahi, bhi, acc[t]);
                }
            } else {
                const h8 bhi = __builtin_bit_cast(h8, b0);
                const h8 blo = __builtin_bit_cast(h8, b1);
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const h8 ahi = AHEAD ? af[tap & 1][t][0] : *reinterpret_cast<const h8*>(lds + arow[t] + tapoff);
                    const h8 alo = AHEAD ? af[tap & 1][t][1] : *reinterpret_cast<const h8*>(lds + arow[t] + tapoff + 32);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi, acc[t], 0, 0, 0);
                }
            }
        }
    }

    if (!wave_active) return;

    const int tile_in_n = (tz_i * p.tilesY + ty_i) * p.tilesX + tx_i;
    conv_epilogue<PREC, WM, MT, TXL, TYL, PREC == 2 || PREC == 5>(p, acc, n, z0, y0, x0, tile_in_n, wm, cout, half, wg.split,
                                                                 asc.inv);
}

// This file is compiled once per arithmetic mode (-DDDPM3D_PREC_ONLY=0|1|2|5, and 3 = the
// Winograd-D kernels of every mode; see the Makefile) so the instantiations build in parallel; the
// mode-independent parts below (split-K reduction, dispatcher) live in the PREC 0 object only.
#ifndef DDPM3D_PREC_ONLY
#error "compile with -DDDPM3D_PREC_ONLY=0, 1, 2, 3 or 5"
#endif
#if DDPM3D_PREC_ONLY == 0

// ------------------------------------------------------- split-K reduction
// out = sum_s slab[s] + bias + residual, plus the GroupNorm partial sums the
// conv epilogue would have written.  One workgroup per DDPM3D_REDUCE_VOX
// voxels of one sample (= one statistics row); threads run along Cout, so every
// slab / residual / output access is a contiguous row segment.
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const ConvK p) {
    const size_t DHW = (size_t)p.D * p.H * p.W;
    const int rows = p.stats_rows;
    const int n = blockIdx.x / rows, r = blockIdx.x % rows;
    const size_t v0 = (size_t)r * p.reduce_vox;
    const size_t v1 = min(v0 + (size_t)p.reduce_vox, DHW);
    const size_t slab_stride = (size_t)p.N * DHW * p.Cout;
    for (int cout = threadIdx.x; cout < p.Cout; cout += 256) {
        const float bias = p.bias[(size_t)n * p.bias_stride_n + cout];
        GnAcc gs;
        gs.init(0.0f);
        float cnt = 0.0f;
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
                ddpm3d_act_store(p.out, e, val, (p.io & DDPM3D_IO_OUT_BF16) != 0, (p.io & DDPM3D_IO_HALF_IS_F16) != 0);
            else
                p.out[((size_t)n * p.Cout + cout) * DHW + v] = val;
            if (cnt == 0.0f) gs.init(val);
            gs.add(val);
            cnt += 1.0f;
        }
        if (p.stats != nullptr)
            *reinterpret_cast<double2*>(p.stats + (((size_t)n * p.Cout + cout) * rows + r) * 2) =
                make_double2(gs.sum1(cnt), gs.sum2(cnt));
    }
}

// Vectorised form for Cout % 4 == 0 and NDHWC output (every split conv of the network):
// workgroup = RV voxels (one statistics row) x 64 channel quads (blockIdx.y); thread = (voxel
// lane vl = tid/64, channel quad cq = tid%64) sums the S slabs for RV/4 voxels x 4 channels
// with 16-byte loads, four slabs in flight (added in slab order, like the scalar kernel),
// then the 4 voxel lanes are folded through LDS for the statistics row.  RV shrinks with
// the level (ddpm3d_reduce_vox) so that the 64x4x4 level still launches 512 workgroups.
template <int RV>
__global__ __launch_bounds__(256) void conv_splitk_reduce_v4_kernel(const ConvK p) {
    const size_t DHW = (size_t)p.D * p.H * p.W;
    const int rows = p.stats_rows;
    const int n = blockIdx.x / rows, r = blockIdx.x % rows;
    const size_t v0 = (size_t)r * RV;
    const size_t slab_stride = (size_t)p.N * DHW * p.Cout;
    const int quads = p.Cout / 4;
    const int cq = threadIdx.x & 63, vl = threadIdx.x >> 6;
    __shared__ double red[2][4][64 * 4];
    {
        const int q = blockIdx.y * 64 + cq;
        const bool qok = q < quads;
        GnAcc gs[4];
        float cnt = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) gs[c].init(0.0f);
        if (qok) {
            const f32x4 bias = *reinterpret_cast<const f32x4*>(p.bias + (size_t)n * p.bias_stride_n + q * 4);
#pragma unroll
            for (int i = 0; i < RV / 4; ++i) {
                const size_t v = v0 + vl + 4 * i;
                if (v < DHW) {
                    const size_t e = ((size_t)n * DHW + v) * p.Cout + q * 4;
                    f32x4 val = *reinterpret_cast<const f32x4*>(p.partial + e);
                    int s = 1;
                    for (; s + 3 < p.ksplit; s += 4) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(p.partial + e + (size_t)s * slab_stride);
                        const f32x4 b = *reinterpret_cast<const f32x4*>(p.partial + e + (size_t)(s + 1) * slab_stride);
                        const f32x4 c = *reinterpret_cast<const f32x4*>(p.partial + e + (size_t)(s + 2) * slab_stride);
                        const f32x4 d = *reinterpret_cast<const f32x4*>(p.partial + e + (size_t)(s + 3) * slab_stride);
                        val += a; val += b; val += c; val += d;
                    }
                    for (; s < p.ksplit; ++s)
                        val += *reinterpret_cast<const f32x4*>(p.partial + e + (size_t)s * slab_stride);
                    val += bias;
                    if (p.res_mode != DDPM3D_RES_NONE) {
                        const int x = (int)(v % p.W), y = (int)((v / p.W) % p.H), z = (int)(v / ((size_t)p.W * p.H));
#pragma unroll
                        for (int c = 0; c < 4; ++c) val[c] += ddpm3d_residual(p, n, z, y, x, q * 4 + c);
                    }
                    if (p.io & DDPM3D_IO_OUT_BF16) {
                        const bool f16 = (p.io & DDPM3D_IO_HALF_IS_F16) != 0;
                        u32x2 pk = u32x2{half_pack(val[0], val[1], f16), half_pack(val[2], val[3], f16)};
                        *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(p.out) + e) = pk;
                        // (8-byte store in a loop: data registers pinned behind it, conv3d_epilogue.h epi_store_b64)
                        asm volatile("s_nop 7" : "+v"(pk) : : "memory");
                    }
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
        }
        if (p.stats != nullptr) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                red[0][vl][cq * 4 + c] = gs[c].sum1(cnt);
                red[1][vl][cq * 4 + c] = gs[c].sum2(cnt);
            }
            __syncthreads();
            if (vl == 0 && qok) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const double a = (red[0][0][cq * 4 + c] + red[0][1][cq * 4 + c]) +
                                     (red[0][2][cq * 4 + c] + red[0][3][cq * 4 + c]);
                    const double b = (red[1][0][cq * 4 + c] + red[1][1][cq * 4 + c]) +
                                     (red[1][2][cq * 4 + c] + red[1][3][cq * 4 + c]);
                    *reinterpret_cast<double2*>(p.stats + (((size_t)n * p.Cout + q * 4 + c) * rows + r) * 2) =
                        make_double2(a, b);
                }
            }
            __syncthreads();
        }
    }
}

hipError_t ddpm3d_launch_splitk_reduce(const ConvK& k, hipStream_t st) {
    if (k.Cout % 4 == 0 && k.out_layout == DDPM3D_OUT_NDHWC) {
        const dim3 grid(k.N * k.stats_rows, (k.Cout / 4 + 63) / 64);
        switch (k.reduce_vox) {
            case 4: hipLaunchKernelGGL(conv_splitk_reduce_v4_kernel<4>, grid, dim3(256), 0, st, k); break;
            case 8: hipLaunchKernelGGL(conv_splitk_reduce_v4_kernel<8>, grid, dim3(256), 0, st, k); break;
            case 16: hipLaunchKernelGGL(conv_splitk_reduce_v4_kernel<16>, grid, dim3(256), 0, st, k); break;
            default: return hipErrorInvalidValue;
        }
    } else
        hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3(k.N * k.stats_rows), dim3(256), 0, st, k);
    return hipGetLastError();
}

hipError_t ddpm3d_launch_conv_p0(const ConvK& k, const ConvCfg& c, hipStream_t st);
hipError_t ddpm3d_launch_conv_p1(const ConvK& k, const ConvCfg& c, hipStream_t st);
hipError_t ddpm3d_launch_conv_p2(const ConvK& k, const ConvCfg& c, hipStream_t st);
hipError_t ddpm3d_launch_conv_p5(const ConvK& k, const ConvCfg& c, hipStream_t st);
hipError_t ddpm3d_launch_conv_wz(const ConvK& k, const ConvCfg& c, hipStream_t st);

hipError_t ddpm3d_launch_conv(const ConvK& k, const ConvCfg& c, hipStream_t st) {
    switch (c.PREC) {
        case 0: return ddpm3d_launch_conv_p0(k, c, st);
        case 1: return ddpm3d_launch_conv_p1(k, c, st);
        case 2: return ddpm3d_launch_conv_p2(k, c, st);
        case 3:
        case 4:
        case 6: return ddpm3d_launch_conv_wz(k, c, st);
        case 5: return ddpm3d_launch_conv_p5(k, c, st);
    }
    return hipErrorInvalidValue;
}
#endif  // DDPM3D_PREC_ONLY == 0

#if DDPM3D_PREC_ONLY == 3
// Winograd-D forms of the f16x3 / f16 / bf16 3x3x3 conv, an object of their own (conv3d_p3.o;
// eligibility is checked by the C ABI)
#include "conv3d_wz.h"
hipError_t ddpm3d_launch_conv_wz(const ConvK& k, const ConvCfg& c, hipStream_t st) {
    const int gy = k.CoutPad / 128;
    const int gx = k.N * k.tilesZ * k.tilesY * k.tilesX;
#ifdef DDPM3D_WZ_STAMPS
    constexpr size_t stamp_lds = 4 * WZ_NSTAMP * 8;     // measurement build: the stamps sit behind the image
#else
    constexpr size_t stamp_lds = 0;
#endif
    constexpr size_t lds = (size_t)WzGeom::BUF + stamp_lds;
    // One kernel form.  Measured and dropped in r02 (profiles/r02_layer_ab_*.txt, r02_wzp_plane_pair_*.txt,
    // DESIGN.md 3.1b): a wave-specialised persistent form (compute waves + loader waves, tile walk,
    // epilogue hand-off through LDS); a 128-row wave tile with one wave per SIMD (plain, and with the
    // staging interleaved into the tap loop); a plane-pair form (8x8x4 tiles, two anti-phase halves of
    // two transformed planes each, half the L2 weight stream) -- 0-15 % behind this one or equal to it.
    // r03: the plane-pair form rebuilt for the one-MFMA modes (f16, bf16), where the L2 weight stream
    // was suspected to be the bound, at weight-ring depths 3 / 4 / 5 / 6 / 8, bit-identical:
    // 15-30 % SLOWER than this form on the 64^3 and 64x32x32 layers, equal below
    // (profiles/r03_layer_ab_plane_pair_{bf16,f16}.txt) -- with a third of the MFMAs its tap phase
    // (72 MFMAs) is shorter than the other half's staging phase, so the anti-phase halves wait for each
    // other.  Not kept (the form is commit 8797410's conv3d_wzp.h with the weight-ring depth made a template
    // parameter).
    // f16x3: the issue order of a tap (conv3d_wz.h, IL): one order for every shape since r03
    // (profiles/r03_layer_ab_wz_issue_order.txt; r02 had picked between 0 and 1 by shape).
    if (c.TXL == 2) {
        // 4x4x8 tiles (the levels below 8x8: r03).  One issue order; the image is 56 KB
        constexpr size_t lds4 = (size_t)WzGeomT<4>::BUF + stamp_lds;
        if (c.PREC == DDPM3D_PREC_F16_WZ)
            hipLaunchKernelGGL((conv3d_wz_kernel<WZ_F16, 0, 4>), dim3(gx, gy, k.ksplit), dim3(256), lds4, st, k);
        else if (c.PREC == DDPM3D_PREC_BF16_WZ)
            hipLaunchKernelGGL((conv3d_wz_kernel<WZ_BF16, 0, 4>), dim3(gx, gy, k.ksplit), dim3(256), lds4, st, k);
        else
            hipLaunchKernelGGL((conv3d_wz_kernel<WZ_F16X3, 4, 4>), dim3(gx, gy, k.ksplit), dim3(256), lds4, st, k);
        return hipGetLastError();
    }
#ifndef DDPM3D_WZ_T84
#define DDPM3D_WZ_T84 1
#endif
    // 8x4x4 tiles (two z-pairs per workgroup; r03) where they tile the volume like the 8x8x2 grid the shape rule
    // counted (H % 8 == 0, D % 4 == 0: same number of tiles, so statistics rows and workspace are unchanged); the
    // default issue order only
    if (DDPM3D_WZ_T84 && k.H % 8 == 0 && k.D % 4 == 0 && !(k.hint & DDPM3D_HINT_WZ_ORDER_MASK)) {
        ConvK k2 = k;
        k2.tilesY = 2 * k.tilesY;
        k2.tilesZ = k.tilesZ / 2;
        constexpr size_t lds84 = (size_t)WzGeomT<8, 4>::BUF + stamp_lds;
        if (c.PREC == DDPM3D_PREC_F16_WZ)
            hipLaunchKernelGGL((conv3d_wz_kernel<WZ_F16, 0, 8, 4>), dim3(gx, gy, k.ksplit), dim3(256), lds84, st, k2);
        else if (c.PREC == DDPM3D_PREC_BF16_WZ)
            hipLaunchKernelGGL((conv3d_wz_kernel<WZ_BF16, 0, 8, 4>), dim3(gx, gy, k.ksplit), dim3(256), lds84, st, k2);
        else
            hipLaunchKernelGGL((conv3d_wz_kernel<WZ_F16X3, 4, 8, 4>), dim3(gx, gy, k.ksplit), dim3(256), lds84, st, k2);
        return hipGetLastError();
    }
    if (c.PREC == DDPM3D_PREC_F16_WZ)
        hipLaunchKernelGGL(conv3d_wz_kernel<WZ_F16>, dim3(gx, gy, k.ksplit), dim3(256), lds, st, k);
    else if (c.PREC == DDPM3D_PREC_BF16_WZ)
        hipLaunchKernelGGL(conv3d_wz_kernel<WZ_BF16>, dim3(gx, gy, k.ksplit), dim3(256), lds, st, k);
    else {
        // kernel_hint bits 12..14: force an issue order (A/B measurements; identical arithmetic), 0 = the default.
        // Built: 0, 1 (r02's two), 2, 4 (r03); 3, 5, 6 of conv3d_wz.h were measured and are not instantiated
        // (every instantiation is ~1 MB of code object to load at the first launch)
        int order = (k.hint & DDPM3D_HINT_WZ_ORDER_MASK) >> DDPM3D_HINT_WZ_ORDER_SHIFT;
        if (order == 0) order = 5;   // (value = IL + 1): weight loads first, reads behind, wave priority 1
        switch (order - 1) {
            case 0: hipLaunchKernelGGL((conv3d_wz_kernel<WZ_F16X3, 0>), dim3(gx, gy, k.ksplit), dim3(256), lds, st, k); break;
            case 1: hipLaunchKernelGGL((conv3d_wz_kernel<WZ_F16X3, 1>), dim3(gx, gy, k.ksplit), dim3(256), lds, st, k); break;
            case 2: hipLaunchKernelGGL((conv3d_wz_kernel<WZ_F16X3, 2>), dim3(gx, gy, k.ksplit), dim3(256), lds, st, k); break;
            case 4: hipLaunchKernelGGL((conv3d_wz_kernel<WZ_F16X3, 4>), dim3(gx, gy, k.ksplit), dim3(256), lds, st, k); break;
            default: return hipErrorInvalidValue;
        }
    }
    return hipGetLastError();
}
#else   // the direct kernels of one arithmetic mode

// ---------------------------------------------------------------- dispatch
template <int PREC, int PIPE, int KS, int WN, int MT, int TXL, int TYL>
static hipError_t launch_cfg(const ConvK& k, int grid_x, int grid_y, hipStream_t st) {
    constexpr int TX = 1 << TXL, TY = 1 << TYL, TZ = (4 / WN) * MT * 32 / (TX * TY);
    constexpr int PAD = KS / 2, SXY = PIPE == 2 ? 2 : 1;
    constexpr int HX = SXY * (TX - 1) + KS, HY = SXY * (TY - 1) + KS, HZ = TZ + 2 * PAD;
    constexpr size_t lds_bytes = (size_t)HZ * LdsGeom<TX, HX, HY>::RZ * 16;
    if constexpr (lds_bytes > 65536) {
        static DynLdsOnce once = {};
        const hipError_t attr = ddpm3d_allow_dynamic_lds(
            once, reinterpret_cast<const void*>(&conv3d_kernel<PREC, PIPE, KS, WN, MT, TXL, TYL>), (int)lds_bytes);
        if (attr != hipSuccess) return attr;
    }
    hipLaunchKernelGGL((conv3d_kernel<PREC, PIPE, KS, WN, MT, TXL, TYL>), dim3(grid_x, grid_y, k.ksplit), dim3(256),
                       lds_bytes, st, k);
    return hipGetLastError();
}

#define DDPM3D_CAT2(a, b) a##b
#define DDPM3D_CAT(a, b) DDPM3D_CAT2(a, b)
hipError_t DDPM3D_CAT(ddpm3d_launch_conv_p, DDPM3D_PREC_ONLY)(const ConvK& k, const ConvCfg& c, hipStream_t st) {
    const int gx = k.N * k.tilesZ * k.tilesY * k.tilesX;
    const int gy = (k.CoutPad + 32 * c.WN - 1) / (32 * c.WN);
    const int pipe = k.in_mode == DDPM3D_IN_STRIDE2 ? 2 : ((k.in_mode == DDPM3D_IN_SAME || k.in_mode == DDPM3D_IN_UP) ? 1 : 0);
#define CASE(P_, PI_, KS_, WN_, MT_, TXL_, TYL_)                                                      \
    if (pipe == PI_ && c.KS == KS_ && c.WN == WN_ && c.MT == MT_ && c.TXL == TXL_ && c.TYL == TYL_)   \
        return launch_cfg<P_, PI_, KS_, WN_, MT_, TXL_, TYL_>(k, gx, gy, st);
#define CASES(P_, PI_)                                                                          \
    CASE(P_, PI_, 3, 4, 4, 3, 3) CASE(P_, PI_, 3, 2, 2, 3, 3) CASE(P_, PI_, 3, 1, 1, 3, 3)      \
    CASE(P_, PI_, 3, 4, 4, 2, 2) CASE(P_, PI_, 3, 2, 2, 2, 2) CASE(P_, PI_, 3, 1, 1, 2, 2)      \
    CASE(P_, PI_, 1, 4, 4, 3, 3) CASE(P_, PI_, 1, 2, 2, 3, 3) CASE(P_, PI_, 1, 1, 1, 3, 3)      \
    CASE(P_, PI_, 1, 4, 4, 2, 2) CASE(P_, PI_, 1, 2, 2, 2, 2) CASE(P_, PI_, 1, 1, 1, 2, 2)
    CASES(DDPM3D_PREC_ONLY, 1) CASES(DDPM3D_PREC_ONLY, 0)
    // stride-(1,2,2) 3x3x3 convs (Downsample with use_conv)
    CASE(DDPM3D_PREC_ONLY, 2, 3, 4, 4, 3, 3) CASE(DDPM3D_PREC_ONLY, 2, 3, 2, 2, 3, 3) CASE(DDPM3D_PREC_ONLY, 2, 3, 1, 1, 3, 3)
    CASE(DDPM3D_PREC_ONLY, 2, 3, 4, 4, 2, 2) CASE(DDPM3D_PREC_ONLY, 2, 3, 2, 2, 2, 2) CASE(DDPM3D_PREC_ONLY, 2, 3, 1, 1, 2, 2)
#undef CASES
#undef CASE
    return hipErrorInvalidValue;
}
#endif  // DDPM3D_PREC_ONLY != 3
