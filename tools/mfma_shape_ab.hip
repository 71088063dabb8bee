// MFMA-shape A/B on the TAP LOOP of the dominant kernel: the same wave tile
// (64 rows x 32 couts x 4 transformed planes = 128 accumulator registers), the same LDS image
// (4 planes x 10x10 voxels x 80 B), the same per-wave weight stream from L2 (2 KB per tap), two
// 256-thread workgroups per CU, 8 "chunks" per workgroup with the two barriers of the real kernel --
// but no staging arithmetic and no epilogue: what is compared is the operand-fetch + MFMA loop.
//
//   X  v_mfma_f32_32x32x16_f16: 36 taps x (2 row tiles x 3 products) = 216 MFMAs of 32 cycles per chunk,
//      A operands one tap ahead, weight ring of 3 taps, loads-first issue order (conv3d_wz.h, IL = 2)
//   Y  v_mfma_f32_16x16x32_f16: K = 32 has to span TWO taps of the 16-channel chunk (a 32-channel
//      chunk would double the staging registers, which the kernel does not have): per transformed plane
//      the 9 (dy, dx) taps pair as (0,1) (3,4) (6,7) [dx, dx+1], (2,5) [dy, dy+1] and tap 8 alone with
//      half of K multiplying zeros -> 20 K-steps x (4 row frags x 2 cout frags x 3 products) = 480 MFMAs
//      of 16 cycles per chunk = 7680 pipe cycles against X's 6912 (+11 %)
//
// Build + run (GPU box):  hipcc -O3 --offload-arch=gfx950 tools/mfma_shape_ab.hip -o scratch/mfma_shape_ab && scratch/mfma_shape_ab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int VS = 5, RY = 56, RZ = 560, BUF = 4 * RZ * 16;   // LDS geometry of conv3d_stage.h (slots of 16 B)
constexpr int NCH = 8, COUTP = 128;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
template <int AUX = 0>
__device__ __forceinline__ u32x4 bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX);
}

__device__ __forceinline__ void fill_image(unsigned char* lds, const _Float16* src, int tid) {
    for (int i = tid; i < BUF / 2; i += 256) reinterpret_cast<_Float16*>(lds)[i] = src[i & 4095];
}

// ------------------------------------------------------------------ X: 32x32x16
// knobs (what bounds the loop?): R = weight ring depth, AH = taps the A operands are read ahead,
// NOA / NOB = no LDS reads / no weight loads behind the first tap (operands reused: measurement only)
// one MFMA with the accumulator in AGPRs (inline asm: the compiler's own form keeps C / D in VGPRs at this
// register budget) or through the builtin
template <bool AG>
__device__ __forceinline__ void mfma_x(f32x16& c, h8 a, h8 b) {
    if constexpr (AG) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// ORD: 2 = weight loads behind MFMAs 1-2, LDS reads behind 3-6 (the shipped order), 0 = all fetches in
// a clump in front of the tap's MFMAs; AG = accumulators in AGPRs (forces the clumped order: the
// scheduler cannot place instructions around inline asm)
template <int R, int AH, bool NOA, bool NOB, int AUX = 0, int ORD = 2, bool AG = false>
__global__ __launch_bounds__(256, 2) void taps_32x32x16(const _Float16* img, const void* w, unsigned w_bytes, float* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    fill_image(lds, img, tid);
    int arow[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int m = t * 32 + (lane & 31);
        arow[t] = ((m >> 3) * RY + (m & 7) * VS + half) * 16;
    }
    const int cout = wn * 32 + (lane & 31);
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(w, w_bytes);
    const unsigned wlane = ((unsigned)cout * 2 + half) * 16, wpart = COUTP * 32, wchunk = 2 * wpart, wtap = NCH * wchunk;
    f32x16 acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[j][t][i] = 0.f;
    for (int chunk = 0; chunk < NCH; ++chunk) {
        __syncthreads();
        __syncthreads();
        unsigned woff = chunk * wchunk;
        auto bump = [&]() { woff += wtap; asm volatile("" : "+s"(woff)); };
        constexpr int NT = 36, AS = AH + 1;
        u32x4 bq[R][2];
#pragma unroll
        for (int s = 0; s < R - 1; ++s) {
            bq[s][0] = bload<AUX>(wr, wlane, woff);
            bq[s][1] = bload<AUX>(wr, wlane, woff + wpart);
            bump();
        }
        h8 af[AS][2][2];
#pragma unroll
        for (int a = 0; a < AH; ++a) {
            const int off1 = ((a / 9) * RZ + ((a / 3) % 3) * RY + (a % 3) * VS) * 16;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                af[a][t][0] = *reinterpret_cast<const h8*>(lds + arow[t] + off1);
                af[a][t][1] = *reinterpret_cast<const h8*>(lds + arow[t] + off1 + 32);
            }
        }
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
            __builtin_amdgcn_sched_barrier(0);
            if (tap + AH < NT) {
                const int t1 = tap + AH;
                const int off1 = ((t1 / 9) * RZ + ((t1 / 3) % 3) * RY + (t1 % 3) * VS) * 16;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (NOA) {
                        af[t1 % AS][t][0] = af[tap % AS][t][0];
                        af[t1 % AS][t][1] = af[tap % AS][t][1];
                    } else {
                        af[t1 % AS][t][0] = *reinterpret_cast<const h8*>(lds + arow[t] + off1);
                        af[t1 % AS][t][1] = *reinterpret_cast<const h8*>(lds + arow[t] + off1 + 32);
                    }
                }
            }
            if (tap + R - 1 < NT) {
                if (NOB) {
                    bq[(tap + R - 1) % R][0] = bq[tap % R][0];
                    bq[(tap + R - 1) % R][1] = bq[tap % R][1];
                } else {
                    bq[(tap + R - 1) % R][0] = bload<AUX>(wr, wlane, woff);
                    bq[(tap + R - 1) % R][1] = bload<AUX>(wr, wlane, woff + wpart);
                    bump();
                }
            }
            const int j = tap / 9;
            const h8 bhi = __builtin_bit_cast(h8, bq[tap % R][0]), blo = __builtin_bit_cast(h8, bq[tap % R][1]);
            if constexpr (ORD == 0 || AG) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 2; ++t) mfma_x<AG>(acc[j][t], af[tap % AS][t][1], bhi);
#pragma unroll
            for (int t = 0; t < 2; ++t) mfma_x<AG>(acc[j][t], af[tap % AS][t][0], blo);
#pragma unroll
            for (int t = 0; t < 2; ++t) mfma_x<AG>(acc[j][t], af[tap % AS][t][0], bhi);
            if constexpr (ORD == 2 && !AG) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
        }
    }
    if constexpr (AG) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) s += acc[j][t][i];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

// ------------------------------------------------------------------ Y: 16x16x32, K = two taps of the chunk
__global__ __launch_bounds__(256, 2) void taps_16x16x32(const _Float16* img, const void* w, unsigned w_bytes, float* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    fill_image(lds, img, tid);
    const int r = lane & 15, kg = lane >> 4;
    // row r of fragment f = (x parity, y quad): conflict-free ds_read_b128 (rows of lane groups A / B
    // land on distinct even / odd 16-byte slots; derivation in DESIGN.md)
    const int xi = r & 3, yi = ((((r >> 3) ^ (r >> 2)) & 1) << 1) | ((r >> 2) & 1);
    int abase[4][3];   // [fragment][pairing: dx, dy, single]
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int x = 2 * xi + (f & 1), y = 4 * (f >> 1) + yi;
        const int v = (y * RY + x * VS) * 16 + (kg & 1) * 16;
        abase[f][0] = v + (kg >> 1) * VS * 16;
        abase[f][1] = v + (kg >> 1) * RY * 16;
        abase[f][2] = v;
    }
    const __amdgpu_buffer_rsrc_t wr = make_rsrc(w, w_bytes);
    const unsigned wpart = COUTP * 32, wchunk = 2 * wpart, wtap = NCH * wchunk;
    unsigned wl[2][3];   // [cout fragment][pairing]
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const unsigned co = wn * 32 + c * 16 + r;
        const unsigned v = (co * 2 + (kg & 1)) * 16;
        wl[c][0] = v + (kg >> 1) * wtap;
        wl[c][1] = v + (kg >> 1) * 3 * wtap;
        wl[c][2] = (kg >> 1) ? 0xFFFFFFF0u : v;      // the single tap: the upper half of K reads zeros
    }
    f32x4 acc[4][4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][f][c][i] = 0.f;
    // the 5 K-steps of a transformed plane: (first tap, pairing)
    constexpr int ST_TAP[5] = {0, 3, 6, 2, 8}, ST_PAIR[5] = {0, 0, 0, 1, 2};
    for (int chunk = 0; chunk < NCH; ++chunk) {
        __syncthreads();
        __syncthreads();
        const unsigned wc = chunk * wchunk;
        constexpr int NS = 20, R = 3;
        u32x4 bq[R][2][2];   // [ring][cout fragment][hi|lo]
        auto issue_b = [&](int s) {
            const int j = s / 5, k = s % 5;
            const unsigned so = wc + (unsigned)(j * 9 + ST_TAP[k]) * wtap;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                bq[s % R][c][0] = bload(wr, wl[c][ST_PAIR[k]], so);
                bq[s % R][c][1] = bload(wr, wl[c][ST_PAIR[k]], so + wpart);
            }
        };
        h8 af[2][4][2];
        auto issue_a = [&](int s) {
            const int j = s / 5, k = s % 5, t = ST_TAP[k];
            const int off = (j * RZ + (t / 3) * RY + (t % 3) * VS) * 16;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                af[s & 1][f][0] = *reinterpret_cast<const h8*>(lds + abase[f][ST_PAIR[k]] + off);
                af[s & 1][f][1] = *reinterpret_cast<const h8*>(lds + abase[f][ST_PAIR[k]] + off + 32);
            }
        };
#pragma unroll
        for (int s = 0; s < R - 1; ++s) issue_b(s);
        issue_a(0);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < NS) issue_a(s + 1);
            if (s + R - 1 < NS) issue_b(s + R - 1);
            const int j = s / 5;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const h8 bhi = __builtin_bit_cast(h8, bq[s % R][c][0]), blo = __builtin_bit_cast(h8, bq[s % R][c][1]);
#pragma unroll
                for (int f = 0; f < 4; ++f) acc[j][f][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[s & 1][f][1], bhi, acc[j][f][c], 0, 0, 0);
#pragma unroll
                for (int f = 0; f < 4; ++f) acc[j][f][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[s & 1][f][0], blo, acc[j][f][c], 0, 0, 0);
#pragma unroll
                for (int f = 0; f < 4; ++f) acc[j][f][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[s & 1][f][0], bhi, acc[j][f][c], 0, 0, 0);
            }
            // 24 MFMAs: weight loads behind the first four, LDS reads behind the next eight
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) s += acc[j][f][c][i];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <typename K>
static float time_kernel(K kern, int nwg, int ldsb, const _Float16* img, const void* w, unsigned wbytes, float* out,
                         hipEvent_t e0, hipEvent_t e1) {
    for (int rep = 0; rep < 3; ++rep) {   // the third burst is the timed one
        hipEventRecord(e0);
        for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), ldsb, 0, img, w, wbytes, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 8;
}

int main() {
    const int nwg = 2048;
    const size_t wbytes = (size_t)36 * NCH * 2 * COUTP * 32 + 4 * 3 * NCH * 2 * COUTP * 32;   // + run-off of the pairings
    _Float16* img; void* w; float* out;
    hipMalloc(&img, 4096 * 2); hipMalloc(&w, wbytes); hipMalloc(&out, (size_t)nwg * 256 * 4);
    std::vector<_Float16> h(wbytes / 2);
    srand(3);
    for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f);
    hipMemcpy(w, h.data(), wbytes, hipMemcpyHostToDevice);
    hipMemcpy(img, h.data(), 4096 * 2, hipMemcpyHostToDevice);
    // 72 KB of LDS per workgroup pins TWO workgroups per CU for every variant, as in the real kernel
    // (X needs 168 registers here -- no staging -- and would otherwise run three)
    constexpr int LDSB = 72 * 1024;
    typedef void (*kern_t)(const _Float16*, const void*, unsigned, float*);
    struct V { const char* name; kern_t k; double mfma_cycles; };
    const V vs[] = {
        {"32x32x16  ring 3, A 1 tap ahead (the shipped loop)", taps_32x32x16<3, 1, false, false>, 6912},
        {"16x16x32  K = tap pairs (480 MFMAs x 16 cycles: +11 %)", taps_16x16x32, 7680},
        {"32x32x16  ring 4", taps_32x32x16<4, 1, false, false>, 6912},
        {"32x32x16  ring 6", taps_32x32x16<6, 1, false, false>, 6912},
        {"32x32x16  ring 4, A 2 taps ahead", taps_32x32x16<4, 2, false, false>, 6912},
        {"32x32x16  ring 6, A 3 taps ahead", taps_32x32x16<6, 3, false, false>, 6912},
        {"32x32x16  clumped fetches (IL 0 order)", taps_32x32x16<3, 1, false, false, 0, 0, false>, 6912},
        {"32x32x16  clumped fetches, accumulators in AGPRs", taps_32x32x16<3, 1, false, false, 0, 0, true>, 6912},
        {"32x32x16  weight loads nt (aux 2)", taps_32x32x16<3, 1, false, false, 2>, 6912},
        {"32x32x16  weight loads sc0 (aux 1)", taps_32x32x16<3, 1, false, false, 1>, 6912},
        {"32x32x16  weight loads sc1 (aux 16)", taps_32x32x16<3, 1, false, false, 16>, 6912},
        {"32x32x16  weight loads sc0 sc1 (aux 17)", taps_32x32x16<3, 1, false, false, 17>, 6912},
        {"32x32x16  no LDS reads", taps_32x32x16<3, 1, true, false>, 6912},
        {"32x32x16  no weight loads", taps_32x32x16<3, 1, false, true>, 6912},
        {"32x32x16  neither (MFMAs + barriers)", taps_32x32x16<3, 1, true, true>, 6912},
    };
    constexpr int NV = sizeof(vs) / sizeof(vs[0]);
    for (const V& v : vs) hipFuncSetAttribute(reinterpret_cast<const void*>(v.k), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> t[NV];
    for (int round = 0; round < 5; ++round)
        for (int i = 0; i < NV; ++i) t[i].push_back(time_kernel(vs[i].k, nwg, LDSB, img, w, (unsigned)wbytes, out, e0, e1));
    if (hipGetLastError() != hipSuccess) { printf("launch error\n"); return 1; }
    const double useful = 2.0 * 2048 * 4 * 216 * 8 * 32 * 32 * 16;   // issued f16 MFMA FLOPs of the 32x32x16 form
    printf("# tap loop of the dominant kernel (128->128 @ 64^3 equivalent: 2048 workgroups x 8 chunks x 216 MFMAs per wave),\n"
           "# two workgroups per CU, no staging arithmetic, no epilogue; ms per launch: median (min) of 5 interleaved rounds;\n"
           "# TFLOP/s = the 32x32x16 form's MFMA FLOPs / time (the 16x16x32 form issues 11 %% more for the same products)\n");
    for (int i = 0; i < NV; ++i) {
        std::sort(t[i].begin(), t[i].end());
        printf("%-58s %.4f (%.4f) ms   %5.0f TFLOP/s\n", vs[i].name, t[i][2], t[i][0], useful / t[i][2] / 1e9);
    }
    printf("registers: shipped loop 168 here (no staging state; 256 in the real kernel), 16x16x32 form 252 BEFORE any staging state\n");
    return 0;
}
