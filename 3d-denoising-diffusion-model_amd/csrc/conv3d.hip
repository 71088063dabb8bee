// Fused 3-D convolution for gfx950 (MI355X): implicit GEMM on the fp32 matrix
// cores with the GroupNorm-apply / SiLU / FiLM / pool / upsample / concat
// prologue and the bias / residual / GroupNorm-statistics epilogue fused in.
//
//   GEMM view      M = output voxels, N = Cout, K = taps * Cin
//   workgroup      256 threads = 4 waves, tile = 128 voxels x (32*WN) couts
//   wave (wm, wn)  MT 32x32 accumulators: 32*MT voxels x 32 couts
//   A operand      input halo tile [voxel][CK ci] staged ONCE per ci-chunk in
//                  LDS (already normalised+activated), read per tap at a
//                  constant LDS offset (ds_read_b128, 4 k-steps per read)
//   B operand      packed weights streamed L2 -> VGPR (global_load_dwordx4,
//                  one per 4 k-steps), private to the wave's 32 couts
//   MFMA           v_mfma_f32_32x32x2_f32: exact fp32 (bitwise an fmaf chain)
//
// K order inside an 8-channel block is permuted (lane half h supplies channel
// 4h+s at step s) so that both operands are 16-byte loads; the weight packer
// (pack.hip) stores that order.
#include <hip/hip_runtime.h>
#include "conv3d_params.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + expf(-v)); }

template <int ACT>
__device__ __forceinline__ f32x4 affine_act(f32x4 v, f32x4 a, f32x4 b) {
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float y = fmaf(v[i], a[i], b[i]);
        r[i] = ACT ? silu_f(y) : y;
    }
    return r;
}

template <int KS, int CK, int WN, int TXL, int TYL>
__global__ __launch_bounds__(256, 2) void conv3d_f32_kernel(const ConvK p) {
    constexpr int WM = 4 / WN;
    constexpr int MT = 4 / WM;  // 32-row accumulators per wave; tile is always 128 voxels
    constexpr int TX = 1 << TXL, TY = 1 << TYL;
    constexpr int TZ = 128 / (TX * TY);
    constexpr int PAD = KS / 2;
    constexpr int HX = TX + 2 * PAD, HY = TY + 2 * PAD, HZ = TZ + 2 * PAD;
    constexpr int HV = HX * HY * HZ;
    constexpr int RS = CK + 4;  // LDS row stride (floats): conflict-light b128 reads
    constexpr int QPV = CK / 4;
    constexpr int KK = CK / 8;
    constexpr int NT = KS * KS * KS;

    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5;

    // XCD-aware tile order: consecutive tiles (which share halo planes) on one XCD's L2
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    int tile = bid;
    const int tx_i = tile % p.tilesX; tile /= p.tilesX;
    const int ty_i = tile % p.tilesY; tile /= p.tilesY;
    const int tz_i = tile % p.tilesZ; tile /= p.tilesZ;
    const int n = tile;
    const int x0 = tx_i * TX, y0 = ty_i * TY, z0 = tz_i * TZ;

    // per-lane A row offsets (floats) at tap (0,0,0), k-offset of this half
    int arow[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = (wm * MT + t) * 32 + (lane & 31);
        const int tx = m & (TX - 1), ty = (m >> TXL) & (TY - 1), tz = m >> (TXL + TYL);
        arow[t] = ((tz * HY + ty) * HX + tx) * RS + half * 4;
    }

    const int cout = blockIdx.y * (32 * WN) + wn * 32 + (lane & 31);
    const bool wave_active = (blockIdx.y * (32 * WN) + wn * 32) < p.CoutPad;
    const int cout_ld = wave_active ? cout : 0;
    // float4 index of this lane's first weight quad
    const f32x4* wbase = reinterpret_cast<const f32x4*>(p.w) + ((size_t)cout_ld * 2 + half);
    const size_t wtap_stride = (size_t)(p.CinPad / 8) * p.CoutPad * 2;  // float4 per tap
    const size_t wcb_stride = (size_t)p.CoutPad * 2;                     // float4 per 8-ci block

    f32x16 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;

    // source geometry
    const int Hs = p.in_mode == DDPM3D_IN_POOL ? 2 * p.H : (p.in_mode == DDPM3D_IN_UP ? p.H / 2 : p.H);
    const int Ws = p.in_mode == DDPM3D_IN_POOL ? 2 * p.W : (p.in_mode == DDPM3D_IN_UP ? p.W / 2 : p.W);

    const int nchunks = p.CinPad / CK;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int c0 = chunk * CK;
        __syncthreads();  // everyone done reading the previous chunk's tile
        // ------------------------------------------------ stage the halo tile
        {
            const bool from0 = c0 < p.C0;
            const float* __restrict__ src = from0 ? p.src0 : p.src1;
            const int Cs = from0 ? p.C0 : p.C1;
            const int cb = from0 ? c0 : c0 - p.C0;
            const int q = tid % QPV;  // fixed per thread: 256 % QPV == 0
            f32x4 ga = {1.f, 1.f, 1.f, 1.f}, gb = {0.f, 0.f, 0.f, 0.f};
            const bool has_aff = p.affA != nullptr;
            if (has_aff && p.in_mode != DDPM3D_IN_PLANAR2) {
                ga = *reinterpret_cast<const f32x4*>(p.affA + (size_t)n * p.Cin + c0 + q * 4);
                gb = *reinterpret_cast<const f32x4*>(p.affB + (size_t)n * p.Cin + c0 + q * 4);
            }
            for (int idx = tid; idx < HV * QPV; idx += 256) {
                const int hv = idx / QPV;
                const int hz = hv / (HY * HX);
                const int rem = hv - hz * (HY * HX);
                const int hy = rem / HX;
                const int hx = rem - hy * HX;
                const int z = z0 - PAD + hz, y = y0 - PAD + hy, x = x0 - PAD + hx;
                const bool inb = (unsigned)z < (unsigned)p.D && (unsigned)y < (unsigned)p.H &&
                                 (unsigned)x < (unsigned)p.W;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (inb) {
                    if (p.in_mode == DDPM3D_IN_SAME) {
                        const size_t vox = (((size_t)n * p.D + z) * p.H + y) * p.W + x;
                        v = *reinterpret_cast<const f32x4*>(src + vox * Cs + cb + q * 4);
                        if (has_aff) v = p.act ? affine_act<1>(v, ga, gb) : affine_act<0>(v, ga, gb);
                    } else if (p.in_mode == DDPM3D_IN_UP) {
                        const size_t vox = (((size_t)n * p.D + z) * Hs + (y >> 1)) * Ws + (x >> 1);
                        v = *reinterpret_cast<const f32x4*>(src + vox * Cs + cb + q * 4);
                        if (has_aff) v = p.act ? affine_act<1>(v, ga, gb) : affine_act<0>(v, ga, gb);
                    } else if (p.in_mode == DDPM3D_IN_POOL) {
                        // AvgPool3d window order (h, w): ((s00 + s01) + s10) + s11, then / 4
                        const size_t vox = (((size_t)n * p.D + z) * Hs + 2 * y) * Ws + 2 * x;
                        const float* b0 = src + vox * Cs + cb + q * 4;
                        f32x4 s00 = *reinterpret_cast<const f32x4*>(b0);
                        f32x4 s01 = *reinterpret_cast<const f32x4*>(b0 + Cs);
                        f32x4 s10 = *reinterpret_cast<const f32x4*>(b0 + (size_t)Ws * Cs);
                        f32x4 s11 = *reinterpret_cast<const f32x4*>(b0 + (size_t)Ws * Cs + Cs);
                        if (has_aff) {
                            if (p.act) {
                                s00 = affine_act<1>(s00, ga, gb); s01 = affine_act<1>(s01, ga, gb);
                                s10 = affine_act<1>(s10, ga, gb); s11 = affine_act<1>(s11, ga, gb);
                            } else {
                                s00 = affine_act<0>(s00, ga, gb); s01 = affine_act<0>(s01, ga, gb);
                                s10 = affine_act<0>(s10, ga, gb); s11 = affine_act<0>(s11, ga, gb);
                            }
                        }
                        v = (((s00 + s01) + s10) + s11) * 0.25f;
                    } else {  // PLANAR2: two single-channel volumes, channels 0 and 1 of the chunk
                        if (q == 0 && chunk == 0) {
                            const size_t vox = (((size_t)n * p.D + z) * p.H + y) * p.W + x;
                            v[0] = p.src0[vox];
                            v[1] = p.src1[vox];
                        }
                    }
                }
                *reinterpret_cast<f32x4*>(lds + hv * RS + q * 4) = v;
            }
        }
        __syncthreads();
        if (!wave_active) continue;

        // ------------------------------------------------ taps x k-steps
        const f32x4* wchunk = wbase + (size_t)chunk * KK * wcb_stride;
        f32x4 bcur[KK], bnxt[KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) bcur[kk] = wchunk[kk * wcb_stride];
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
            if (tap + 1 < NT) {
#pragma unroll
                for (int kk = 0; kk < KK; ++kk)
                    bnxt[kk] = wchunk[(size_t)(tap + 1) * wtap_stride + kk * wcb_stride];
            }
            const int dz = tap / (KS * KS), dy = (tap / KS) % KS, dx = tap % KS;
            const int tapoff = ((dz * HY + dy) * HX + dx) * RS;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                f32x4 a[MT];
#pragma unroll
                for (int t = 0; t < MT; ++t)
                    a[t] = *reinterpret_cast<const f32x4*>(lds + arow[t] + tapoff + kk * 8);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int t = 0; t < MT; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][s], bcur[kk][s], acc[t], 0, 0, 0);
            }
            if (tap + 1 < NT) {
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) bcur[kk] = bnxt[kk];
            }
        }
    }

    if (!wave_active) return;

    // ---------------------------------------------------------- epilogue
    // C/D map of 32x32 MFMA: col = lane&31 (cout), row = (reg&3) + 8*(reg>>2) + 4*half
    const bool cvalid = cout < p.Cout;
    const float bias = cvalid ? p.bias[(size_t)n * p.bias_stride_n + cout] : 0.0f;
    float s1 = 0.0f, s2 = 0.0f;
    const size_t DHW = (size_t)p.D * p.H * p.W;
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
                float val = acc[t][reg] + bias;
                const size_t vox = ((size_t)z * p.H + y) * p.W + x;
                if (p.res_mode == DDPM3D_RES_SAME) {
                    val += p.res[((size_t)n * DHW + vox) * p.Cout + cout];
                } else if (p.res_mode == DDPM3D_RES_UP) {
                    const int Hr = p.H / 2, Wr = p.W / 2;
                    const size_t rv = (((size_t)n * p.D + z) * Hr + (y >> 1)) * Wr + (x >> 1);
                    val += p.res[rv * p.Cout + cout];
                } else if (p.res_mode == DDPM3D_RES_POOL) {
                    const int Hr = p.H * 2, Wr = p.W * 2;
                    const size_t rv = (((size_t)n * p.D + z) * Hr + 2 * y) * Wr + 2 * x;
                    const float* r0 = p.res + rv * p.Cout + cout;
                    const float r = ((r0[0] + r0[p.Cout]) + r0[(size_t)Wr * p.Cout]) +
                                    r0[(size_t)Wr * p.Cout + p.Cout];
                    val += r * 0.25f;
                }
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
            const int tile_in_n = (tz_i * p.tilesY + ty_i) * p.tilesX + tx_i;
            const size_t row = (size_t)n * p.stats_rows + (size_t)tile_in_n * WM + wm;
            float2 v2 = make_float2(s1, s2);
            *reinterpret_cast<float2*>(p.stats + (row * p.Cout + cout) * 2) = v2;
        }
    }
}

// ---------------------------------------------------------------- dispatch
template <int KS, int WN, int TXL, int TYL>
static hipError_t launch_cfg(const ConvK& k, int grid_x, int grid_y, hipStream_t st) {
    constexpr int CK = DDPM3D_CONV_CK;
    constexpr int WM = 4 / WN;
    constexpr int TX = 1 << TXL, TY = 1 << TYL, TZ = 128 / (TX * TY);
    constexpr int PAD = KS / 2;
    constexpr int HV = (TX + 2 * PAD) * (TY + 2 * PAD) * (TZ + 2 * PAD);
    constexpr size_t lds_bytes = (size_t)HV * (CK + 4) * sizeof(float);
    (void)WM;
    hipLaunchKernelGGL((conv3d_f32_kernel<KS, CK, WN, TXL, TYL>), dim3(grid_x, grid_y, 1), dim3(256),
                       lds_bytes, st, k);
    return hipGetLastError();
}

hipError_t ddpm3d_launch_conv_f32(const ConvK& k, const ConvCfg& c, hipStream_t st) {
    const int gx = k.N * k.tilesZ * k.tilesY * k.tilesX;
    const int gy = (k.CoutPad + 32 * c.WN - 1) / (32 * c.WN);
#define CASE(KS_, WN_, TXL_, TYL_) \
    if (c.KS == KS_ && c.WN == WN_ && c.TXL == TXL_ && c.TYL == TYL_) return launch_cfg<KS_, WN_, TXL_, TYL_>(k, gx, gy, st);
    CASE(3, 4, 3, 3) CASE(3, 2, 3, 3) CASE(3, 1, 3, 3)
    CASE(3, 4, 2, 2) CASE(3, 2, 2, 2) CASE(3, 1, 2, 2)
    CASE(1, 4, 3, 3) CASE(1, 2, 3, 3) CASE(1, 1, 3, 3)
    CASE(1, 4, 2, 2) CASE(1, 2, 2, 2) CASE(1, 1, 2, 2)
#undef CASE
    return hipErrorInvalidValue;
}
