// Internal launcher prototypes (definitions in ops.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

hipError_t ddpm3d_launch_pack(const float* w, int Cout, int Cin, int ks, int prec, void* out, hipStream_t st);
hipError_t ddpm3d_launch_gn_finalize(const double* st0, int C0, int rows0, const double* st1, int C1,
                                     int rows1, int N, int groups, double count, float eps,
                                     const float* gamma, const float* beta, const float* film,
                                     int film_stride, int film_off, float* A, float* B, float* bound,
                                     hipStream_t st);
hipError_t ddpm3d_launch_absmax(const float* x0, const float* x1, int N, size_t per_sample, float* bound,
                                hipStream_t st);
hipError_t ddpm3d_launch_gn_stats(const float* x, int N, int voxels, int C, double* stats, hipStream_t st);
int ddpm3d_gn_stats_rows_impl(int voxels);
hipError_t ddpm3d_launch_timestep_embedding(const float* t, int rows, int dim, const float* freqs,
                                            float* out, hipStream_t st);
hipError_t ddpm3d_launch_linear(const float* in, int rows, int K, const float* w, const float* bias, int O,
                                int silu_in, float* out, int out_stride, hipStream_t st);
hipError_t ddpm3d_launch_transpose(const float* in, int N, int R, int S, float* out, hipStream_t st);
hipError_t ddpm3d_launch_to_ndhwc_pad(const float* in, int N, int C, int voxels, int Cpad, float* out, hipStream_t st);
hipError_t ddpm3d_launch_subsample_hw2(const float* in, int N, int D, int H, int W, int C, float* out,
                                       hipStream_t st);
hipError_t ddpm3d_launch_sample_step(bool ddim, const float* mo, const float* x, const float* noise,
                                     const float* coef, const int64_t* t_idx, int N, int voxels, int flags,
                                     float eta, float* sample, float* pred_xstart, hipStream_t st);
hipError_t ddpm3d_launch_attention(const float* qkv, int N, int T, int heads, int ch, int precision,
                                   const float* bound, int bound_count, int bound_stride, float* out,
                                   hipStream_t st);
hipError_t ddpm3d_launch_add_embedding(float* emb, const float* table, const int64_t* idx, int rows, int dim,
                                       int num_classes, hipStream_t st);
hipError_t ddpm3d_launch_pool_act(const float* src, const float* A, const float* B, int act, int fast, int N, int D,
                                  int H, int W, int C, float* out, int src16, int out16, int f16, hipStream_t st);
// probe.hip
double ddpm3d_probe_flops_per_iter(int kind);
hipError_t ddpm3d_launch_mfma_probe(int kind, int iters, int blocks, float* out, unsigned long long* clk,
                                    hipStream_t st);
