// Sustained MFMA rate and in-kernel clock of this device (register-only loop,
// random operands, one or two waves per SIMD on every CU).  Calibration tool:
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o scratch/mfma_peak && scratch/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int MODE>  // 0: f16 32x32x16, 1: f32 32x32x2
__global__ __launch_bounds__(256) void mfma_loop(const float* in, float* out, int iters,
                                                 unsigned long long* clk) {
    const int lane = threadIdx.x;
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    h8 a[4], b;
    float af[4], bf;
    for (int t = 0; t < 4; ++t) {
        for (int j = 0; j < 8; ++j) a[t][j] = (_Float16)in[(lane * 37 + t * 11 + j) & 1023];
        af[t] = in[(lane * 13 + t) & 1023];
    }
    for (int j = 0; j < 8; ++j) b[j] = (_Float16)in[(lane * 7 + j) & 1023];
    bf = in[(lane * 5) & 1023];
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (MODE == 0)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t], b, acc[t], 0, 0, 0);
                else
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t], bf, acc[t], 0, 0, 0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 16; ++i) s += acc[t][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        clk[blockIdx.x * 2] = t1 - t0;
        clk[blockIdx.x * 2 + 1] = r1 - r0;
    }
}

int main() {
    const int blocks_per_cu[2] = {1, 2};
    float* in; float* out; unsigned long long* clk;
    hipMalloc(&in, 1024 * 4); hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&clk, 256 * 8 * 16);
    std::vector<float> h(1024);
    srand(1);
    for (auto& v : h) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        for (int b = 0; b < 2; ++b) {
            const int grid = 256 * blocks_per_cu[b];
            const int iters = mode == 0 ? 40000 : 20000;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(grid), dim3(256), 0, 0, in, out, iters, clk);
                else hipLaunchKernelGGL(mfma_loop<1>, dim3(grid), dim3(256), 0, 0, in, out, iters, clk);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> c(grid * 2);
            hipMemcpy(c.data(), clk, grid * 16, hipMemcpyDeviceToHost);
            double ghz = 0;
            for (int i = 0; i < grid; ++i) ghz += (double)c[2 * i] / (double)c[2 * i + 1] * 0.1;
            ghz /= grid;
            const double flops = (mode == 0 ? 32.0 * 32 * 16 * 2 : 32.0 * 32 * 2 * 2) * 16.0 * iters * 4 * grid;
            printf("%s  %d block/CU (%d waves/SIMD): %.2f ms  %.1f TFLOP/s  in-kernel clock %.2f GHz\n",
                   mode == 0 ? "f16 32x32x16" : "f32 32x32x2 ", blocks_per_cu[b], blocks_per_cu[b], ms,
                   flops / ms / 1e9, ghz);
        }
    }
    return 0;
}
