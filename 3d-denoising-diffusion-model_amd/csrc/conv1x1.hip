// The 1x1x1 convolutions on raw inputs -- the ResBlock skip connections (unet.py:173-186,
// `skip_connection = conv_nd(dims, channels, out_channels, 1)`), whose input is the block input itself,
// the decoder's being the virtual concat [h, skip] -- as a GEMM whose A operand never touches LDS.
//
// The general kernel (conv3d.hip) stages every 16-channel chunk through LDS behind two barriers for ONE
// tap = 12 MFMAs per wave: the 1x1 layers ran at the latency of that round trip, not at any bandwidth
// (r03 layer table: 25-50 us for the 14 low-resolution ones, 2-5 GFLOP each; 3.5 TB/s on the three
// 64^3 ones that move 400 MB).  Here:
//
//   workgroup  256 threads = 4 waves, tile = 128 voxels (the same 8x8x2 / 4x4x8 tile) x 128 couts
//   wave w     the tile's rows 32w .. 32w+31 x all 128 couts: four 32x32 accumulators
//   A operand  straight from global memory into the MFMA layout.  K is walked in blocks of 32 input
//              channels; lane (row, g = lane / 32) owns channels 16g .. 16g+15 of the block -- 64
//              contiguous bytes of its voxel's row (fp32; 32 bytes of a 16-bit tensor), a whole 128-byte
//              line per row and block -- and feeds channels 16g + 8s .. + 7 to the block's MFMA step s.
//              The weights are read in the same permuted order out of the unchanged packed image
//              (chunk 2b + g, k-half s).
//   B operand  the block's 128 couts x 32 channels (x hi | lo) are the same for the four waves: staged
//              once per workgroup through LDS (16 KB a block in the split-f16 form), in the order the
//              MFMA reads it, so a wave's read is 1 KB contiguous
//   pipeline   two (split-f16) or four blocks in flight per thread (its A rows and its share of the weight stage), ONE
//              barrier per block; every load of a block has the same distance, so the in-order vmcnt
//              never waits for a younger load than it needs
//   epilogue   conv_epilogue (bias, residual, 16-bit stores, split-K slabs) per 32-cout accumulator
//
// Eligibility (ddpm3d_pw_ok, checked by the C ABI): ksize 1, input mode SAME, no affine / activation
// prologue, no statistics, Cout % 128 == 0, Cin % 32 == 0 (and C0 % 32 == 0 for a concat), both sources
// of one element width, precision F16X3 / F16 / BF16.  Everything else stays on conv3d.hip's form.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "conv3d_load.h"
#include "conv3d_epilogue.h"

namespace {

// blocks in flight per thread: its registers hold that many blocks of A rows and weight-stage units (the
// split-f16 form, 32 registers a block, spills into the loop at three)
#ifndef DDPM3D_PW_DEPTH_X3
#define DDPM3D_PW_DEPTH_X3 2
#endif
#ifndef DDPM3D_PW_DEPTH_ONE
#define DDPM3D_PW_DEPTH_ONE 4
#endif
#ifndef DDPM3D_PW_WGS
#define DDPM3D_PW_WGS 2          // workgroups per CU the register budget is held to
#endif
#ifndef DDPM3D_PW_WIDE
#define DDPM3D_PW_WIDE 1         // conv_epilogue's 16-byte store form
#endif
constexpr int pw_depth(int prec) { return prec == 1 ? DDPM3D_PW_DEPTH_X3 : DDPM3D_PW_DEPTH_ONE; }

template <int I, int N, class F>
__device__ __forceinline__ void pw_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        pw_static_for<I + 1, N>(f);
    }
}

// 8 fp32 -> the MFMA operand(s) of one step
template <int PREC>
__device__ __forceinline__ void pw_operand(const float (&v)[8], float s, h8& hi, h8& lo) {
    if constexpr (PREC == 5) {
        u32x4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = bf16_pack(v[2 * i], v[2 * i + 1]);
        hi = __builtin_bit_cast(h8, r);
        lo = hi;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float sc = v[i] * s;                 // |sc| < 2^15 by the choice of the scale
            hi[i] = (_Float16)sc;
            lo[i] = PREC == 1 ? (_Float16)(sc - (float)hi[i]) : hi[i];
        }
    }
}

template <int PREC, int S16, int TXL, int TYL, bool WIDE>
__global__ __launch_bounds__(256, DDPM3D_PW_WGS) void conv3d_pw_kernel(const ConvK p) {
    constexpr int TX = 1 << TXL, TY = 1 << TYL, TZ = 128 / (TX * TY);
    constexpr bool LO = PREC == 1;
    constexpr int NP = LO ? 2 : 1;                   // weight parts (hi | lo)
    constexpr int NA = S16 ? 2 : 4;                  // 16-byte loads of a lane's 16 channels
    constexpr int NB = 2 * NP;                       // 16-byte units of the weight stage per thread
    constexpr int STAGE = 512 * NP * 16;             // bytes: [cout tile 4][part NP][step 2][g 2][cout 32][16 B]
    constexpr int D = pw_depth(PREC);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Measurement build only (-DDDPM3D_PW_STAMPS, tools/pw_stamps.py; never in the shipped library): s_memtime at
    // entry / first loads issued / first block done / loop done / epilogue done, dumped to the workspace
#ifdef DDPM3D_PW_STAMPS
    unsigned long long pw_st[5];
#define PW_STAMP(i) pw_st[i] = __builtin_amdgcn_s_memtime()
#define PW_DUMP() do { if (p.partial != nullptr && p.ksplit == 1 && lane == 0) { \
        const size_t wgl = blockIdx.x + (size_t)gridDim.x * blockIdx.y; \
        unsigned long long* dump = reinterpret_cast<unsigned long long*>(p.partial) + (wgl * 4 + wave) * 8; \
        for (int i = 0; i < 5; ++i) dump[i] = pw_st[i]; \
        dump[5] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define PW_STAMP(i) do { } while (0)
#define PW_DUMP() do { } while (0)
#endif
    PW_STAMP(0);
    const WgId wg = wg_id(p);
    int tile = wg.tile;
    const int tx_i = tile % p.tilesX; tile /= p.tilesX;
    const int ty_i = tile % p.tilesY; tile /= p.tilesY;
    const int tz_i = tile % p.tilesZ; tile /= p.tilesZ;
    const int n = tile;
    const int x0 = tx_i * TX, y0 = ty_i * TY, z0 = tz_i * TZ;

    // this lane's A row: voxel m of the tile (rows beyond the volume read zeros and are never stored)
    const int m = wave * 32 + (lane & 31);
    const int z = z0 + (m >> (TXL + TYL)), y = y0 + ((m >> TXL) & (TY - 1)), x = x0 + (m & (TX - 1));
    const bool inb = halo_inb(p, z, y, x);
    constexpr unsigned ES = S16 ? 2u : 4u;
    const unsigned vox = (unsigned)(((n * p.D + z) * p.H + y) * p.W + x);
    const unsigned voff0 = inb ? vox * (unsigned)p.C0 * ES + g * 16 * ES : DDPM3D_OOB_OFFSET;
    const unsigned voff1 = inb ? vox * (unsigned)p.C1 * ES + g * 16 * ES : DDPM3D_OOB_OFFSET;

    // this thread's share of the weight stage: LDS slots tid + 256 i
    //   slot = (((j * NP + part) * 2 + s) * 2 + gg) * 32 + c   <-   image[chunk 2b + gg][part][cout][k-half s]
    const unsigned wpart = (unsigned)p.CoutPad * 32, wchunk_stride = 2 * wpart;
    const int sc_ = tid & 31, sgg = (tid >> 5) & 1, ss = (tid >> 6) & 1, shi = tid >> 7;
    const unsigned wthread = (unsigned)sgg * wchunk_stride + (LO ? (unsigned)shi * wpart : (unsigned)shi * 1024u) +
                             ((unsigned)(wg.cy * 128 + sc_) * 2 + ss) * 16;
    constexpr unsigned WSTEP = LO ? 1024u : 2048u;   // next unit of the thread: cout tile + 1 (LO) or + 2

    // the range bound is requested first and folded behind the first blocks' loads (most of these launches are
    // a handful of blocks long: a dependent round trip in front of them is a tenth of the kernel).  (The output
    // scales and the bias requested here too, for the epilogue: eight more registers live through the loop,
    // which then spills -- measured slower, r03.)
    float bound_raw = 0.0f;
    if constexpr (PREC == 1 || PREC == 2) bound_raw = act_scale_load(p, n);
    ActScale asc = {1.0f, 1.0f};

    f32x16 acc[4][1];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][0][i] = 0.0f;

    // split-K: blockIdx.z owns a contiguous, even range of the 16-channel chunks
    const int chunk_begin = wg.split * p.chunks_per_split;
    const int chunk_end = min(p.CinPad / DDPM3D_CONV_CK, chunk_begin + p.chunks_per_split);
    const int bb = chunk_begin >> 1, be = chunk_end >> 1;

    // Every issue below is UNCONDITIONAL (a block past the split's range loads with a zero-length descriptor:
    // zeros, no memory traffic): the vmcnt the compiler computes at a join is the most conservative of the
    // paths, so one conditional issue anywhere would turn every wait of the loop into vmcnt(0).
    u32x4 ar[D][NA], br[D][NB];
    auto issue = [&](auto slot, const int b) {
        constexpr int SL = decltype(slot)::value;
        const bool valid = b < be;
        const int c0 = b * 32;
        const bool from0 = c0 < p.C0;
        const __amdgpu_buffer_rsrc_t srsrc =
            make_rsrc(from0 ? p.src0 : p.src1, valid ? (from0 ? p.src0_bytes : p.src1_bytes) : 0u);
        const unsigned voff = from0 ? voff0 : voff1;
        const unsigned soff = valid ? (unsigned)(from0 ? c0 : c0 - p.C0) * ES : 0u;
#pragma unroll
        for (int k = 0; k < NA; ++k) ar[SL][k] = buffer_load16(srsrc, voff, soff + 16 * k);
        const __amdgpu_buffer_rsrc_t wr = make_rsrc(p.w, valid ? p.w_bytes : 0u);
        const unsigned wsoff = valid ? (unsigned)(2 * b) * wchunk_stride : 0u;
#pragma unroll
        for (int i = 0; i < NB; ++i) br[SL][i] = buffer_load16(wr, wthread + WSTEP * i, wsoff);
    };
    auto block = [&](auto slot, const int b) {
        constexpr int SL = decltype(slot)::value;
        unsigned char* stage = lds + SL * STAGE;
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<u32x4*>(stage + (tid + 256 * i) * 16) = br[SL][i];
        h8 ahi[2], alo[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float v[8];
            if constexpr (S16) {
                const f32x4 q0 = half4_expand(u32x2{ar[SL][s][0], ar[SL][s][1]}, (p.io & DDPM3D_IO_HALF_IS_F16) != 0);
                const f32x4 q1 = half4_expand(u32x2{ar[SL][s][2], ar[SL][s][3]}, (p.io & DDPM3D_IO_HALF_IS_F16) != 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) { v[i] = q0[i]; v[4 + i] = q1[i]; }
            } else {
                // (whole-vector casts: __builtin_bit_cast of ONE ext-vector element reads element 0 -- hipcc 7.2)
                const f32x4 q0 = __builtin_bit_cast(f32x4, ar[SL][2 * s]), q1 = __builtin_bit_cast(f32x4, ar[SL][2 * s + 1]);
#pragma unroll
                for (int i = 0; i < 4; ++i) { v[i] = q0[i]; v[4 + i] = q1[i]; }
            }
            pw_operand<PREC>(v, asc.s, ahi[s], alo[s]);
        }
        // the slot's registers are free again -- and the new loads must land in THEM: a load the scheduler
        // hoists above the last read of the old value gets a fresh register and a copy at the loop's back
        // edge, behind an s_waitcnt vmcnt(0) (the whole prefetch drained every iteration)
        // (the empty asm pins the operands HERE: without it the optimiser sinks the conversions to the MFMAs
        // behind the barrier, past the re-issue)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 th = __builtin_bit_cast(u32x4, ahi[s]), tl = __builtin_bit_cast(u32x4, alo[s]);
            if constexpr (LO) asm volatile("" : "+v"(th), "+v"(tl) :: "memory");
            else asm volatile("" : "+v"(th) :: "memory");
            ahi[s] = __builtin_bit_cast(h8, th);
            alo[s] = __builtin_bit_cast(h8, tl);
        }
        issue(slot, b + D);
        __syncthreads();                             // the stage is complete (its last readers: D blocks ago)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const h8 bhi = *reinterpret_cast<const h8*>(stage + ((j * NP) * 2 + s) * 1024 + lane * 16);
                if constexpr (LO) {
                    const h8 blo = *reinterpret_cast<const h8*>(stage + ((j * NP + 1) * 2 + s) * 1024 + lane * 16);
                    acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[s], bhi, acc[j][0], 0, 0, 0);
                    acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[s], blo, acc[j][0], 0, 0, 0);
                    acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[s], bhi, acc[j][0], 0, 0, 0);
                } else {
                    acc[j][0] = mfma16<PREC == 5>(ahi[s], bhi, acc[j][0]);
                }
            }
        }
    };
    // No branch inside the loop: a block past the split's range multiplies zeros (at most D - 1 of them per
    // workgroup; none for the network's channel counts, whose block counts are multiples of 4).
    if (bb < be) {
        pw_static_for<0, D>([&](auto u) { issue(u, bb + decltype(u)::value); });
        if constexpr (PREC == 1 || PREC == 2) asc = act_scale_finish(bound_raw, 1.0f);
        PW_STAMP(1);
        for (int b = bb; b < be; b += D) {
            pw_static_for<0, D>([&](auto u) { block(u, b + decltype(u)::value); });
#ifdef DDPM3D_PW_STAMPS
            if (b == bb) PW_STAMP(2);
#endif
        }
    }
    PW_STAMP(3);

    const int tile_in_n = (tz_i * p.tilesY + ty_i) * p.tilesX + tx_i;
    // The four accumulators are four cout blocks, each with its own output scale and bias: all eight values are
    // requested HERE, in one round trip.  Left to the four epilogue calls they are four dependent round trips in a
    // row (no load moves above the previous call's stores), which was most of the epilogue's 15.7 k cycles -- itself
    // half the life of a wave on the short layers (profiles/r04_pw_stamps_*.txt).  (Requested at kernel START they
    // stay live through the K loop, which then spills: r03.)
    float ws[4], bs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int cout = wg.cy * 128 + j * 32 + (lane & 31);
        const bool cv = cout < p.Cout;
        ws[j] = cv ? p.wscale[cout] : 1.0f;
        bs[j] = (cv && p.ksplit == 1) ? p.bias[(size_t)n * p.bias_stride_n + cout] : 0.0f;
    }
    if constexpr (WIDE) {
        f32x16 accs[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) accs[j] = acc[j][0];
        if (conv_epilogue_lean<PREC, 4, 1, 4, TXL, TYL, false, false>(p, accs, n, z0, y0, x0, tile_in_n, wave,
                                                                     wg.cy * 128 + (lane & 31), g, wg.split, asc.inv, ws, bs)) {
            PW_STAMP(4);
            PW_DUMP();
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        conv_epilogue<PREC, 4, 1, TXL, TYL, WIDE>(p, acc[j], n, z0, y0, x0, tile_in_n, wave,
                                                  wg.cy * 128 + j * 32 + (lane & 31), g, wg.split, asc.inv, true, ws[j], bs[j]);
    PW_STAMP(4);
    PW_DUMP();
}

template <int PREC, int S16, int TXL, int TYL>
hipError_t pw_launch(const ConvK& k, hipStream_t st) {
    constexpr int NP = PREC == 1 ? 2 : 1;
    const int gx = k.N * k.tilesZ * k.tilesY * k.tilesX, gy = k.CoutPad / 128;
    hipLaunchKernelGGL((conv3d_pw_kernel<PREC, S16, TXL, TYL, DDPM3D_PW_WIDE != 0>), dim3(gx, gy, k.ksplit), dim3(256),
                       pw_depth(PREC) * 512 * NP * 16, st, k);
    return hipGetLastError();
}

template <int PREC>
hipError_t pw_launch_prec(const ConvK& k, const ConvCfg& c, hipStream_t st) {
    const bool s16 = (k.io & DDPM3D_IO_SRC0_BF16) != 0;
    if (c.TXL == 3) return s16 ? pw_launch<PREC, 1, 3, 3>(k, st) : pw_launch<PREC, 0, 3, 3>(k, st);
    return s16 ? pw_launch<PREC, 1, 2, 2>(k, st) : pw_launch<PREC, 0, 2, 2>(k, st);
}

}  // namespace

hipError_t ddpm3d_launch_conv_pw(const ConvK& k, const ConvCfg& c, hipStream_t st) {
    switch (c.PREC) {
        case 1: return pw_launch_prec<1>(k, c, st);
        case 2: return pw_launch_prec<2>(k, c, st);
        case 5: return pw_launch_prec<5>(k, c, st);
    }
    return hipErrorInvalidValue;
}
