// Device calibration probe (ddpm3d_mfma_probe): a register-only MFMA loop on pseudo-random
// operands, one or two waves per SIMD on every CU.  What it is for: MI355X boards of one pool
// sustain different clocks under matrix load (+-6 % measured between boxes), so a roofline
// fraction against the NOMINAL peak cannot be compared across runs; bench.py times this loop
// beside the workload and reports the dominant kernel against what THIS device sustains.
// No memory traffic inside the loop; the only stores are the accumulator checksum (keeps the
// loop alive) and two clock stamps per workgroup (s_memtime / s_memrealtime -> in-kernel GHz).
#include <hip/hip_runtime.h>
#include "ops.h"
#include "ddpm3d.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

// cheap per-lane pseudo-random value in [-1, 1)
__device__ __forceinline__ float probe_rand(unsigned s) {
    s = s * 2654435761u + 0x9E3779B9u;
    s ^= s >> 15;
    s *= 2246822519u;
    s ^= s >> 13;
    return (float)(s & 0xFFFFu) * (1.0f / 32768.0f) - 1.0f;
}

// KIND: DDPM3D_PROBE_* (ddpm3d.h).  Every kind keeps 64 rows x 32 couts x 4 "planes" of fp32
// accumulators per wave, the wave tile of the dominant conv kernel.
template <int KIND>
__global__ __launch_bounds__(256, 2) void mfma_probe_kernel(int iters, float* out, unsigned long long* clk) {
    const unsigned lane = threadIdx.x + blockIdx.x * 977u;
    unsigned long long t0, r0, t1, r1;
    float s = 0.f;
    if constexpr (KIND == DDPM3D_PROBE_F16_32X32X16 || KIND == DDPM3D_PROBE_BF16_32X32X16) {
        f32x16 acc[8];
        h8 a[4], b[2];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) a[t][j] = (_Float16)probe_rand(lane * 64 + t * 8 + j);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) b[t][j] = (_Float16)probe_rand(lane * 64 + 32 + t * 8 + j);
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    if constexpr (KIND == DDPM3D_PROBE_F16_32X32X16)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(t + u) & 3], b[t & 1], acc[t], 0, 0, 0);
                    else
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a[(t + u) & 3]),
                                                                         __builtin_bit_cast(bf8, b[t & 1]), acc[t], 0, 0, 0);
                }
        }
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) s += acc[t][i];
    } else if constexpr (KIND == DDPM3D_PROBE_F16_16X16X32 || KIND == DDPM3D_PROBE_BF16_16X16X32) {
        f32x4 acc[32];
        h8 a[4], b[2];
#pragma unroll
        for (int t = 0; t < 32; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[t][i] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) a[t][j] = (_Float16)probe_rand(lane * 64 + t * 8 + j);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) b[t][j] = (_Float16)probe_rand(lane * 64 + 32 + t * 8 + j);
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 32; ++t) {
                if constexpr (KIND == DDPM3D_PROBE_F16_16X16X32)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t & 3], b[(t >> 2) & 1], acc[t], 0, 0, 0);
                else
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a[t & 3]),
                                                                     __builtin_bit_cast(bf8, b[(t >> 2) & 1]), acc[t], 0, 0, 0);
            }
        }
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
        for (int t = 0; t < 32; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) s += acc[t][i];
    } else {   // DDPM3D_PROBE_F32_32X32X2
        f32x16 acc[4];
        float a[4], b[2];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t] = probe_rand(lane * 8 + t);
#pragma unroll
        for (int t = 0; t < 2; ++t) b[t] = probe_rand(lane * 8 + 4 + t);
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(t + u) & 3], b[u & 1], acc[t], 0, 0, 0);
        }
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) s += acc[t][i];
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        clk[blockIdx.x * 2] = t1 - t0;       // shader cycles
        clk[blockIdx.x * 2 + 1] = r1 - r0;   // 100 MHz ticks
    }
}

// FLOPs one workgroup (4 waves) issues per loop iteration
double ddpm3d_probe_flops_per_iter(int kind) {
    switch (kind) {
        case DDPM3D_PROBE_F16_32X32X16:
        case DDPM3D_PROBE_BF16_32X32X16: return 4.0 * 16 * (2.0 * 32 * 32 * 16);
        case DDPM3D_PROBE_F16_16X16X32:
        case DDPM3D_PROBE_BF16_16X16X32: return 4.0 * 32 * (2.0 * 16 * 16 * 32);
        case DDPM3D_PROBE_F32_32X32X2: return 4.0 * 16 * (2.0 * 32 * 32 * 2);
    }
    return 0.0;
}

hipError_t ddpm3d_launch_mfma_probe(int kind, int iters, int blocks, float* out, unsigned long long* clk,
                                    hipStream_t st) {
    switch (kind) {
#define PROBE(K_) case K_: hipLaunchKernelGGL(mfma_probe_kernel<K_>, dim3(blocks), dim3(256), 0, st, iters, out, clk); break;
        PROBE(DDPM3D_PROBE_F16_32X32X16) PROBE(DDPM3D_PROBE_F16_16X16X32) PROBE(DDPM3D_PROBE_F32_32X32X2)
        PROBE(DDPM3D_PROBE_BF16_32X32X16) PROBE(DDPM3D_PROBE_BF16_16X16X32)
#undef PROBE
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
