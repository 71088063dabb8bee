// Small gfx950 kernels around the conv: weight packing, GroupNorm statistics
// folding, timestep embedding, Linear, layout changes and the per-step
// sampler update.  All HBM- or latency-bound; wave64 throughout.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "conv3d_params.h"
#include "conv3d_load.h"   // affine_act, act_quad, half_pack
#include "ops.h"

// ------------------------------------------------------------------ packing
// OIDHW -> [tap][ci/8][CoutPad][8]  (zero padded in ci and cout).  The inner 8
// is the channel within the 8-block; lane half h of the conv reads [4h, 4h+4).
__global__ void pack_weight_kernel(const float* __restrict__ w, int Cout, int Cin, int taps,
                                   int CoutPad, int CinPad, float* __restrict__ out) {
    const size_t total = (size_t)taps * CinPad * CoutPad;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7);
        size_t r = i >> 3;
        const int co = (int)(r % CoutPad); r /= CoutPad;
        const int cb = (int)(r % (CinPad / 8));
        const int tap = (int)(r / (CinPad / 8));
        const int ci = cb * 8 + j;
        float v = 0.0f;
        if (co < Cout && ci < Cin) v = w[((size_t)co * Cin + ci) * taps + tap];
        out[i] = v;
    }
}

// Split-f16 image.  Pass 1: per output channel, the power-of-two scale
// s = 2^floor(log2(W_TARGET / max|w|)) and the epilogue factor 1 / s.
__global__ __launch_bounds__(256) void pack_x3_scale_kernel(const float* __restrict__ w, int Cout, int per_cout,
                                                            float* __restrict__ wscale) {
    const int co = blockIdx.x;
    float m = 0.0f;
    if (co < Cout)
        for (int i = threadIdx.x; i < per_cout; i += 256) m = fmaxf(m, fabsf(w[(size_t)co * per_cout + i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        float s = 1.0f;
        if (m > 0.0f && m < 3.0e38f) s = exp2f(floorf(log2f(DDPM3D_X3_W_TARGET / m)));
        s = fminf(fmaxf(s, 1.0f / 16777216.0f), 16777216.0f);
        wscale[co] = 1.0f / s;  // exact: powers of two
    }
}

// Pass 2: OIDHW -> [tap][ci/16][hi|lo][CoutPad][16 f16] of w * s[cout] (zero padded),
// with s = 1 / wscale[cout] recovered exactly from pass 1's output.
// bf16 = true: the bf16 mode's image in the same layout -- hi = bf16(w) (no scale, wscale = 1), lo = 0
__device__ __forceinline__ _Float16 bf16_bits_as_h(float v) {
    return __builtin_bit_cast(_Float16, __builtin_bit_cast(unsigned short, (__bf16)v));
}
__global__ void pack_ones_kernel(float* __restrict__ wscale, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) wscale[i] = 1.0f;
}

__global__ void pack_x3_kernel(const float* __restrict__ w, const float* __restrict__ wscale, int Cout,
                               int Cin, int taps, int CoutPad, int CinPad, _Float16* __restrict__ out, bool bf16) {
    const size_t total = (size_t)taps * CinPad * CoutPad;  // one (hi, lo) pair per element
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i & 15);
        size_t r = i >> 4;
        const int co = (int)(r % CoutPad); r /= CoutPad;
        const int cb = (int)(r % (CinPad / 16));
        const int tap = (int)(r / (CinPad / 16));
        const int ci = cb * 16 + j;
        float v = 0.0f;
        if (co < Cout && ci < Cin)
            v = w[((size_t)co * Cin + ci) * taps + tap] * (1.0f / wscale[co]);
        const _Float16 hi = bf16 ? bf16_bits_as_h(v) : (_Float16)v;
        const _Float16 lo = bf16 ? (_Float16)0.0f : (_Float16)(v - (float)hi);
        const size_t base = (((size_t)tap * (CinPad / 16) + cb) * 2) * CoutPad * 16;
        out[base + (size_t)co * 16 + j] = hi;
        out[base + (size_t)CoutPad * 16 + (size_t)co * 16 + j] = lo;
    }
}

// Winograd F(2,3) along depth (conv3d_wz.h): weight transform at pack time.  For every
// (cout, ci, dy, dx): U0 = g0, U1 = (g0+g1+g2)/2, U2 = (g0-g1+g2)/2, U3 = g2 with g_k the
// weight at dz = k; 36 "taps" ordered (j, dy, dx).
__device__ __forceinline__ float wz_weight(const float* __restrict__ w, size_t co_ci, int j, int dydx) {
    const float g0 = w[co_ci * 27 + dydx], g1 = w[co_ci * 27 + 9 + dydx], g2 = w[co_ci * 27 + 18 + dydx];
    if (j == 0) return g0;
    if (j == 3) return g2;
    return j == 1 ? 0.5f * ((g0 + g2) + g1) : 0.5f * ((g0 + g2) - g1);
}

__global__ __launch_bounds__(256) void pack_wz_scale_kernel(const float* __restrict__ w, int Cout, int Cin,
                                                            float* __restrict__ wscale) {
    const int co = blockIdx.x;
    float m = 0.0f;
    if (co < Cout)
        for (int i = threadIdx.x; i < Cin * 36; i += 256) {
            const int ci = i / 36, tap = i % 36;
            m = fmaxf(m, fabsf(wz_weight(w, (size_t)co * Cin + ci, tap / 9, tap % 9)));
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        float s = 1.0f;
        if (m > 0.0f && m < 3.0e38f) s = exp2f(floorf(log2f(DDPM3D_X3_W_TARGET / m)));
        s = fminf(fmaxf(s, 1.0f / 16777216.0f), 16777216.0f);
        wscale[co] = 1.0f / s;
    }
}

__global__ void pack_wz_kernel(const float* __restrict__ w, const float* __restrict__ wscale, int Cout, int Cin,
                               int CoutPad, int CinPad, _Float16* __restrict__ out, bool bf16) {
    const size_t total = (size_t)36 * CinPad * CoutPad;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int jj = (int)(i & 15);
        size_t r = i >> 4;
        const int co = (int)(r % CoutPad); r /= CoutPad;
        const int cb = (int)(r % (CinPad / 16));
        const int tap = (int)(r / (CinPad / 16));
        const int ci = cb * 16 + jj;
        float v = 0.0f;
        if (co < Cout && ci < Cin)
            v = wz_weight(w, (size_t)co * Cin + ci, tap / 9, tap % 9) * (1.0f / wscale[co]);
        const _Float16 hi = bf16 ? bf16_bits_as_h(v) : (_Float16)v;
        const _Float16 lo = bf16 ? (_Float16)0.0f : (_Float16)(v - (float)hi);
        const size_t base = (((size_t)tap * (CinPad / 16) + cb) * 2) * CoutPad * 16;
        out[base + (size_t)co * 16 + jj] = hi;
        out[base + (size_t)CoutPad * 16 + (size_t)co * 16 + jj] = lo;
    }
}

hipError_t ddpm3d_launch_pack(const float* w, int Cout, int Cin, int ks, int prec, void* out, hipStream_t st) {
    const int CoutPad = ddpm3d_cout_pad(Cout), CinPad = ddpm3d_cin_pad(Cin);
    const bool bf16 = prec == DDPM3D_PREC_BF16 || prec == DDPM3D_PREC_BF16_WZ;
    if (prec == DDPM3D_PREC_F16X3_WZ || prec == DDPM3D_PREC_F16_WZ || prec == DDPM3D_PREC_BF16_WZ) {
        // [16-bit image of 36 transformed taps][CoutPad fp32 wscale]
        const size_t total = (size_t)36 * CinPad * CoutPad;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        float* wscale = reinterpret_cast<float*>(reinterpret_cast<char*>(out) + total * 4);
        if (bf16)
            hipLaunchKernelGGL(pack_ones_kernel, dim3((CoutPad + 255) / 256), dim3(256), 0, st, wscale, CoutPad);
        else
            hipLaunchKernelGGL(pack_wz_scale_kernel, dim3(CoutPad), dim3(256), 0, st, w, Cout, Cin, wscale);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(pack_wz_kernel, dim3(blocks), dim3(256), 0, st, w, wscale, Cout, Cin, CoutPad, CinPad,
                           (_Float16*)out, bf16);
        return hipGetLastError();
    }
    const int taps = ks * ks * ks;
    const size_t total = (size_t)taps * CinPad * CoutPad;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (prec == 0) {
        hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, st, w, Cout, Cin, taps, CoutPad,
                           CinPad, (float*)out);
        return hipGetLastError();
    }
    // PREC 1: [f16 image (total hi/lo pairs = total*4 bytes)][CoutPad fp32 wscale]
    float* wscale = reinterpret_cast<float*>(reinterpret_cast<char*>(out) + total * 4);
    if (bf16)
        hipLaunchKernelGGL(pack_ones_kernel, dim3((CoutPad + 255) / 256), dim3(256), 0, st, wscale, CoutPad);
    else
        hipLaunchKernelGGL(pack_x3_scale_kernel, dim3(CoutPad), dim3(256), 0, st, w, Cout, Cin * taps, wscale);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(pack_x3_kernel, dim3(blocks), dim3(256), 0, st, w, wscale, Cout, Cin, taps, CoutPad,
                       CinPad, (_Float16*)out, bf16);
    return hipGetLastError();
}

// ------------------------------------------------------------- wave helpers
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ------------------------------------------------------------- GN finalize
// One workgroup per (n, group): fold partial (sum, sumsq) rows in fp64, then
// write A, B for the group's channels.  256 threads, or 1024 where a group's run of partial sums
// is long (the launcher decides from the shape alone, so results stay repeatable): the kernel is a
// latency chain -- N x 32 workgroups on 256 CUs, each waiting for its own loads -- and four times
// the loads in flight cut the long ones (2048 rows at the 64^3 level: 128 KB per group) from
// 17-33 us to the launch floor (r03: the fp64 statistics had doubled this kernel's bytes).
__global__ __launch_bounds__(1024) void gn_finalize_kernel(
    const double* __restrict__ st0, int C0, int rows0, const double* __restrict__ st1, int C1, int rows1,
    int groups, double count, float eps, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ film, int film_stride, int film_off, float* __restrict__ A,
    float* __restrict__ B, float* __restrict__ bound) {
    const int C = C0 + C1;
    const int cg = C / groups;
    const int n = blockIdx.x / groups, g = blockIdx.x % groups;
    const int cbeg = g * cg;
    const bool from0 = cbeg < C0;
    const double* st = from0 ? st0 : st1;
    const int Cs = from0 ? C0 : C1;
    const int rows = from0 ? rows0 : rows1;
    const int cs = from0 ? cbeg : cbeg - C0;
    // statistics are channel-major [N][C][rows][2] fp64: the group's cg channels x rows partial sums
    // are ONE contiguous, 16-byte aligned run of cg*rows (sum, sum of squares) pairs
    // the affine's inputs do not depend on the statistics: requested FIRST, so that their latency runs
    // beside the partial-sum loads' instead of behind the reduction (one dependent round trip less)
    const int c_own = cbeg + (int)(threadIdx.x < (unsigned)cg ? threadIdx.x : 0);
    float g_own = 0.0f, b_own = 0.0f, sc_own = 1.0f, sh_own = 0.0f;
    if (gamma != nullptr) {
        g_own = gamma[c_own];
        b_own = beta[c_own];
        if (film != nullptr) {
            sc_own = 1.0f + film[(size_t)n * film_stride + film_off + c_own];
            sh_own = film[(size_t)n * film_stride + film_off + C + c_own];
        }
    }
    double s1 = 0.0, s2 = 0.0;
    float m2 = 0.0f;                                  // largest sum of squares of any row of the group
    const size_t items = (size_t)rows * cg;          // double2 count
    const double2* run = reinterpret_cast<const double2*>(st + ((size_t)n * Cs + cs) * rows * 2);
    // Eight 16-byte loads in flight per thread, UNCONDITIONALLY: buffer loads past the run return zeros, which change
    // neither sum nor maximum.  (r01-r04 had `#pragma unroll 8` on a loop with a run-time trip count: what came out
    // was eight guarded loads, each behind an s_waitcnt vmcnt(0) -- eight dependent round trips for the 64^3 level's
    // eight items per thread, 12-15 us of the kernel's 5 us floor.  Same additions in the same order.)
    {
        const unsigned bytes = items * 16 < 0xFFFFFFF0ull ? (unsigned)(items * 16) : 0xFFFFFFF0u;
        const __amdgpu_buffer_rsrc_t rr = make_rsrc(run, bytes);
        const unsigned step = blockDim.x * 16u;
        for (size_t i0 = 0; i0 < items; i0 += (size_t)blockDim.x * 8) {
            u32x4 t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = buffer_load16(rr, (unsigned)(i0 + threadIdx.x) * 16u + k * step, 0);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const double2 v = __builtin_bit_cast(double2, t[k]);
                s1 += v.x;
                s2 += v.y;
                m2 = fmaxf(m2, (float)v.y);
            }
        }
    }
    __shared__ double red[2][16];
    __shared__ float redm[16], redab[2][16];
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m2 = fmaxf(m2, __shfl_xor(m2, o));
    const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wave] = s1; red[1][wave] = s2; redm[wave] = m2; }
    __syncthreads();
    s1 = 0.0; s2 = 0.0; m2 = 0.0f;
    for (int w = 0; w < nwaves; ++w) {               // fixed order: the same sums on every run
        s1 += red[0][w];
        s2 += red[1][w];
        m2 = fmaxf(m2, redm[w]);
    }
    // every |x| of the group is at most the square root of the largest row's sum of squares
    const float xmax = sqrtf(m2);
    if (gamma == nullptr) {                          // bounds only (tensor consumed without a GroupNorm)
        if (threadIdx.x == 0 && bound != nullptr) {
            bound[(size_t)blockIdx.x * 2] = xmax;
            bound[(size_t)blockIdx.x * 2 + 1] = xmax;
        }
        return;
    }
    const double cnt = count * cg;
    const double mean = s1 / cnt;
    double var = s2 / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float meanf = (float)mean;
    float amax = 0.0f, bmax = 0.0f;
    for (int c = threadIdx.x; c < cg; c += blockDim.x) {
        const int ch = cbeg + c;
        const bool own = c == (int)threadIdx.x;       // (cg <= blockDim.x: every channel is some thread's own)
        float a = rstd * (own ? g_own : gamma[ch]);
        float b = fmaf(-meanf, a, own ? b_own : beta[ch]);
        if (film != nullptr) {
            const float sc = own ? sc_own : 1.0f + film[(size_t)n * film_stride + film_off + ch];
            const float sh = own ? sh_own : film[(size_t)n * film_stride + film_off + C + ch];
            a = a * sc;
            b = fmaf(b, sc, sh);
        }
        A[(size_t)n * C + ch] = a;
        B[(size_t)n * C + ch] = b;
        amax = fmaxf(amax, fabsf(a));
        bmax = fmaxf(bmax, fabsf(b));
    }
    if (bound != nullptr) {
        // |act(a x + b)| <= |a| xmax + |b| for act = identity or SiLU
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            amax = fmaxf(amax, __shfl_xor(amax, o));
            bmax = fmaxf(bmax, __shfl_xor(bmax, o));
        }
        if ((threadIdx.x & 63) == 0) { redab[0][wave] = amax; redab[1][wave] = bmax; }
        __syncthreads();
        if (threadIdx.x == 0) {
            amax = 0.0f; bmax = 0.0f;
            for (int w = 0; w < nwaves; ++w) {
                amax = fmaxf(amax, redab[0][w]);
                bmax = fmaxf(bmax, redab[1][w]);
            }
            bound[(size_t)blockIdx.x * 2] = fmaf(amax, xmax, bmax);
            bound[(size_t)blockIdx.x * 2 + 1] = xmax;
        }
    }
}

// max |x| per sample of up to two tensors: bound[n * count + t].  ABSMAX_PARTS workgroups per (sample, tensor)
// fold their slices with an atomic max on the bit patterns (non-negative floats order like unsigned integers:
// exact, so the result does not depend on arrival order); the launcher zeroes `bound` first.  (r03: ONE
// 1024-thread workgroup per tensor walked its 1 MB in a 64-deep dependent loop: 36 us per DDPM step.)
#define ABSMAX_PARTS 32
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x0, const float* __restrict__ x1,
                                                     size_t per_sample, int count, float* __restrict__ bound) {
    const int part = blockIdx.x % ABSMAX_PARTS, nt = blockIdx.x / ABSMAX_PARTS;
    const int n = nt / count, t = nt % count;
    const float* x = (t == 0 ? x0 : x1) + (size_t)n * per_sample;
    float m = 0.0f;
    const size_t quads = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) ? per_sample / 4 : 0;
    const size_t qper = (quads + ABSMAX_PARTS - 1) / ABSMAX_PARTS;
    const size_t q0 = (size_t)part * qper, q1 = q0 + qper < quads ? q0 + qper : quads;
#pragma unroll 4
    for (size_t i = q0 + threadIdx.x; i < q1; i += 256) {
        const float4 v = *reinterpret_cast<const float4*>(x + i * 4);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (part == 0)   // the unaligned / remainder elements
        for (size_t i = quads * 4 + threadIdx.x; i < per_sample; i += 256) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) m = fmaxf(m, red[w]);
        // m is a non-negative, non-NaN float (fmaxf drops NaNs): its bits order like the value
        atomicMax(reinterpret_cast<unsigned*>(bound) + nt, __float_as_uint(m));
    }
}

hipError_t ddpm3d_launch_absmax(const float* x0, const float* x1, int N, size_t per_sample, float* bound,
                                hipStream_t st) {
    const int count = x1 ? 2 : 1;
    hipError_t e = hipMemsetAsync(bound, 0, sizeof(float) * (size_t)N * count, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(absmax_kernel, dim3(N * count * ABSMAX_PARTS), dim3(256), 0, st, x0, x1, per_sample, count, bound);
    return hipGetLastError();
}

hipError_t ddpm3d_launch_gn_finalize(const double* st0, int C0, int rows0, const double* st1, int C1,
                                     int rows1, int N, int groups, double count, float eps,
                                     const float* gamma, const float* beta, const float* film,
                                     int film_stride, int film_off, float* A, float* B, float* bound,
                                     hipStream_t st) {
    const long long items = (long long)(rows0 > rows1 ? rows0 : rows1) * ((C0 + C1) / groups);   // double2 per group
    const int threads = items >= 2048 ? 1024 : 256;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(N * groups), dim3(threads), 0, st, st0, C0, rows0, st1, C1,
                       rows1, groups, count, eps, gamma, beta, film, film_stride, film_off, A, B, bound);
    return hipGetLastError();
}

// --------------------------------------------------------------- GN stats
// Stand-alone statistics pass for NDHWC tensors no conv epilogue produced.
// One workgroup per (n, row of GN_STATS_VOX voxels): thread c-strided over
// channels, 16-byte loads along C.
#define GN_STATS_VOX 256
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, int voxels, int C,
                                                       int rows, double* __restrict__ stats) {
    const int n = blockIdx.x / rows, r = blockIdx.x % rows;
    const int v0 = r * GN_STATS_VOX;
    const int v1 = min(v0 + GN_STATS_VOX, voxels);
    const int C4 = C / 4;
    // thread -> (channel quad, voxel lane); quads fastest so a wave reads contiguous bytes
    const int qpt = min(C4, 256);
    const int vlanes = 256 / qpt;
    __shared__ double sh[2][256 * 4];
    for (int q0 = 0; q0 < C4; q0 += qpt) {
        const int q = q0 + (threadIdx.x % qpt);
        const int vl = threadIdx.x / qpt;
        double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};   // fp64 sums (cf. conv3d_params.h: GnAcc, the conv epilogues' pivoted fp32 form)
        if (q < C4 && vl < vlanes) {
            for (int v = v0 + vl; v < v1; v += vlanes) {
                const float4 t = *reinterpret_cast<const float4*>(x + ((size_t)n * voxels + v) * C + q * 4);
                const double tx = t.x, ty = t.y, tz = t.z, tw = t.w;
                s1[0] += tx; s1[1] += ty; s1[2] += tz; s1[3] += tw;
                s2[0] = fma(tx, tx, s2[0]); s2[1] = fma(ty, ty, s2[1]);
                s2[2] = fma(tz, tz, s2[2]); s2[3] = fma(tw, tw, s2[3]);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { sh[0][threadIdx.x * 4 + i] = s1[i]; sh[1][threadIdx.x * 4 + i] = s2[i]; }
        __syncthreads();
        if (threadIdx.x < qpt && q < C4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                double a = 0.0, b = 0.0;
                for (int l = 0; l < vlanes; ++l) {
                    a += sh[0][(l * qpt + threadIdx.x) * 4 + i];
                    b += sh[1][(l * qpt + threadIdx.x) * 4 + i];
                }
                double* o = stats + (((size_t)n * C + q * 4 + i) * rows + r) * 2;
                o[0] = a;
                o[1] = b;
            }
        }
        __syncthreads();
    }
}

hipError_t ddpm3d_launch_gn_stats(const float* x, int N, int voxels, int C, double* stats, hipStream_t st) {
    const int rows = (voxels + GN_STATS_VOX - 1) / GN_STATS_VOX;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(N * rows), dim3(256), 0, st, x, voxels, C, rows, stats);
    return hipGetLastError();
}
int ddpm3d_gn_stats_rows_impl(int voxels) { return (voxels + GN_STATS_VOX - 1) / GN_STATS_VOX; }

// ------------------------------------------------------ timestep embedding
__global__ void timestep_embedding_kernel(const float* __restrict__ t, int rows, int dim,
                                          const float* __restrict__ freqs, float* __restrict__ out) {
    const int half = dim / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * dim) return;
    const int r = i / dim, j = i - r * dim;
    float v = 0.0f;
    if (j < 2 * half) {
        const int f = j < half ? j : j - half;
        // args = t * freqs in fp32 (nn.py:116); cos / sin evaluated in fp64 and rounded
        // once, i.e. the correctly rounded fp32 value a <=1-ulp fp32 libm returns almost always
        const float arg = t[r] * freqs[f];
        v = j < half ? (float)cos((double)arg) : (float)sin((double)arg);
    }
    out[i] = v;
}

hipError_t ddpm3d_launch_timestep_embedding(const float* t, int rows, int dim, const float* freqs,
                                            float* out, hipStream_t st) {
    const int total = rows * dim;
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((total + 255) / 256), dim3(256), 0, st, t, rows, dim,
                       freqs, out);
    return hipGetLastError();
}

// ------------------------------------------------- pooled activation (pre-pass)
// out[n][z][y][x][c] = mean over the (1,2,2) window of act(A[n][c] * src + B[n][c]): the IN_POOL prologue of
// ddpm3d_conv3d (conv3d_load.h halo_fetch: same window order, same SiLU) as a pass of its own, so that the
// down ResBlocks' first conv (unet.py:238-242: h = in_conv(h_upd(in_rest(x)))) can run the Winograd-D form on
// a plain tensor.  One thread = one output voxel x 4 channels: four 16-byte loads, one 16-byte store.
template <bool FAST>
__global__ __launch_bounds__(256) void pool_act_kernel(const float* __restrict__ src, const float* __restrict__ A,
                                                       const float* __restrict__ B, int act, int N, int D, int H,
                                                       int W, int C, float* __restrict__ out, int src16,
                                                       int out16, int f16) {
    const int quads = C >> 2;
    const size_t total = (size_t)N * D * H * W * quads;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int cq = (int)(i % quads);
    size_t v = i / quads;
    const int x = (int)(v % W); v /= W;
    const int y = (int)(v % H); v /= H;
    const int z = (int)(v % D);
    const int n = (int)(v / D);
    const size_t Ws = 2 * (size_t)W, Hs = 2 * (size_t)H;
    const size_t e0 = ((((size_t)n * D + z) * Hs + 2 * y) * Ws + 2 * x) * C + cq * 4;
    f32x4 s00 = act_quad(src, e0, src16 != 0, f16 != 0);
    f32x4 s01 = act_quad(src, e0 + C, src16 != 0, f16 != 0);
    f32x4 s10 = act_quad(src, e0 + Ws * C, src16 != 0, f16 != 0);
    f32x4 s11 = act_quad(src, e0 + Ws * C + C, src16 != 0, f16 != 0);
    if (A != nullptr) {
        const f32x4 ga = *reinterpret_cast<const f32x4*>(A + (size_t)n * C + cq * 4);
        const f32x4 gb = *reinterpret_cast<const f32x4*>(B + (size_t)n * C + cq * 4);
        if (act) {
            s00 = affine_act<1, FAST>(s00, ga, gb); s01 = affine_act<1, FAST>(s01, ga, gb);
            s10 = affine_act<1, FAST>(s10, ga, gb); s11 = affine_act<1, FAST>(s11, ga, gb);
        } else {
            s00 = affine_act<0, FAST>(s00, ga, gb); s01 = affine_act<0, FAST>(s01, ga, gb);
            s10 = affine_act<0, FAST>(s10, ga, gb); s11 = affine_act<0, FAST>(s11, ga, gb);
        }
    }
    // AvgPool3d window order (h, w): ((s00 + s01) + s10) + s11, then * 1/4
    const f32x4 r = (((s00 + s01) + s10) + s11) * 0.25f;
    const size_t eo = i * 4;
    if (out16)
        *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(out) + eo) =
            u32x2{half_pack(r[0], r[1], f16 != 0), half_pack(r[2], r[3], f16 != 0)};
    else
        *reinterpret_cast<f32x4*>(out + eo) = r;
}

hipError_t ddpm3d_launch_pool_act(const float* src, const float* A, const float* B, int act, int fast, int N, int D,
                                  int H, int W, int C, float* out, int src16, int out16, int f16, hipStream_t st) {
    const size_t total = (size_t)N * D * H * W * (C / 4);
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (fast)
        hipLaunchKernelGGL(pool_act_kernel<true>, dim3(blocks), dim3(256), 0, st, src, A, B, act, N, D, H, W, C, out,
                           src16, out16, f16);
    else
        hipLaunchKernelGGL(pool_act_kernel<false>, dim3(blocks), dim3(256), 0, st, src, A, B, act, N, D, H, W, C, out,
                           src16, out16, f16);
    return hipGetLastError();
}

// ------------------------------------------------------------ label embedding
// emb[r][:] += table[idx[r]][:]  (unet.py:703-705: emb = emb + self.label_emb(y))
// A label outside [0, num_classes) reads nothing (its row stays as it was): the C ABI is safe by itself,
// the Python host still refuses such labels where nn.Embedding would.
__global__ void add_embedding_kernel(float* __restrict__ emb, const float* __restrict__ table,
                                     const int64_t* __restrict__ idx, int rows, int dim, int num_classes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * dim) return;
    const int r = i / dim, j = i - r * dim;
    const int64_t c = idx[r];
    if ((uint64_t)c < (uint64_t)num_classes) emb[i] += table[(size_t)c * dim + j];
}

hipError_t ddpm3d_launch_add_embedding(float* emb, const float* table, const int64_t* idx, int rows, int dim,
                                       int num_classes, hipStream_t st) {
    const int total = rows * dim;
    hipLaunchKernelGGL(add_embedding_kernel, dim3((total + 255) / 256), dim3(256), 0, st, emb, table, idx, rows, dim,
                       num_classes);
    return hipGetLastError();
}

// ------------------------------------------------------------------ linear
// One wave per output feature o, LIN_ROWS rows of the batch at a time: the
// weight row is read once (coalesced 256-B segments) and reused across rows.
#define LIN_ROWS 8
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ in, int rows, int K,
                                                     const float* __restrict__ w,
                                                     const float* __restrict__ bias, int O, int silu_in,
                                                     float* __restrict__ out, int out_stride) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int r0 = blockIdx.y * LIN_ROWS;
    if (o >= O) return;
    float acc[LIN_ROWS];
#pragma unroll
    for (int r = 0; r < LIN_ROWS; ++r) acc[r] = 0.0f;
    for (int k = lane; k < K; k += 64) {
        const float wv = w[(size_t)o * K + k];
#pragma unroll
        for (int r = 0; r < LIN_ROWS; ++r) {
            if (r0 + r < rows) {
                float xv = in[(size_t)(r0 + r) * K + k];
                if (silu_in) xv = xv / (1.0f + expf(-xv));
                acc[r] = fmaf(xv, wv, acc[r]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < LIN_ROWS; ++r) {
        const float s = wave_sum(acc[r]);
        if (lane == 0 && r0 + r < rows) out[(size_t)(r0 + r) * out_stride + o] = s + bias[o];
    }
}

hipError_t ddpm3d_launch_linear(const float* in, int rows, int K, const float* w, const float* bias, int O,
                                int silu_in, float* out, int out_stride, hipStream_t st) {
    dim3 grid((O + 3) / 4, (rows + LIN_ROWS - 1) / LIN_ROWS);
    hipLaunchKernelGGL(linear_kernel, grid, dim3(256), 0, st, in, rows, K, w, bias, O, silu_in, out,
                       out_stride);
    return hipGetLastError();
}

// ----------------------------------------------------------------- layouts
// LDS-tiled transpose between [C][V] and [V][C] per sample (32x32 tiles, +1 pad).
__global__ __launch_bounds__(256) void transpose_cv_kernel(const float* __restrict__ in, int R, int S,
                                                           float* __restrict__ out) {
    // in: [n][R][S] -> out: [n][S][R]
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int s0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const float* src = in + (size_t)n * R * S;
    float* dst = out + (size_t)n * R * S;
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int r = r0 + ty + j, s = s0 + tx;
        tile[ty + j][tx] = (r < R && s < S) ? src[(size_t)r * S + s] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int s = s0 + ty + j, r = r0 + tx;
        if (r < R && s < S) dst[(size_t)s * R + r] = tile[tx][ty + j];
    }
}

__global__ __launch_bounds__(256) void to_ndhwc_pad_kernel(const float* __restrict__ in, int C, int voxels, int Cpad,
                                                           float* __restrict__ out) {
    const int n = blockIdx.y;
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= voxels) return;
    float* o = out + ((size_t)n * voxels + v) * Cpad;
    for (int c = 0; c < Cpad; ++c) o[c] = c < C ? in[((size_t)n * C + c) * voxels + v] : 0.0f;
}
hipError_t ddpm3d_launch_to_ndhwc_pad(const float* in, int N, int C, int voxels, int Cpad, float* out, hipStream_t st) {
    hipLaunchKernelGGL(to_ndhwc_pad_kernel, dim3((voxels + 255) / 256, N), dim3(256), 0, st, in, C, voxels, Cpad, out);
    return hipGetLastError();
}

hipError_t ddpm3d_launch_transpose(const float* in, int N, int R, int S, float* out, hipStream_t st) {
    dim3 grid((S + 31) / 32, (R + 31) / 32, N);
    hipLaunchKernelGGL(transpose_cv_kernel, grid, dim3(256), 0, st, in, R, S, out);
    return hipGetLastError();
}

// out[n][z][y][x][:] = in[n][z][2y][2x][:]  (NDHWC, C % 4 == 0): the even positions of a
// stride-1 conv output = the stride-(1,2,2) conv of Downsample(use_conv=True), unet.py:129-133.
__global__ __launch_bounds__(256) void subsample_hw2_kernel(const float4* __restrict__ in, int D, int H, int W,
                                                            int C4, float4* __restrict__ out, size_t total) {
    const int Ho = H / 2, Wo = W / 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4);
        size_t v = i / C4;
        const int x = (int)(v % Wo); v /= Wo;
        const int y = (int)(v % Ho); v /= Ho;   // v = n * D + z
        out[i] = in[((v * H + 2 * y) * W + 2 * x) * C4 + c];
    }
}

hipError_t ddpm3d_launch_subsample_hw2(const float* in, int N, int D, int H, int W, int C, float* out,
                                       hipStream_t st) {
    const size_t total = (size_t)N * D * (H / 2) * (W / 2) * (C / 4);
    const size_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(subsample_hw2_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, st,
                       reinterpret_cast<const float4*>(in), D, H, W, C / 4, reinterpret_cast<float4*>(out), total);
    return hipGetLastError();
}

// ---------------------------------------------------------- sampler update
// The arithmetic order mirrors the reference's torch expressions one rounding
// at a time, so contraction into FMAs is switched off here.
#pragma clang fp contract(off)

template <bool DDIM>
__global__ __launch_bounds__(256) void sample_step_kernel(
    const float* __restrict__ mo, const float* __restrict__ x, const float* __restrict__ noise,
    const float* __restrict__ coef, const int64_t* __restrict__ t_idx, int voxels, int flags, float eta,
    float* __restrict__ sample, float* __restrict__ pred_xstart) {
    const int n = blockIdx.y;
    const int64_t ti = t_idx[n];
    const float* c = coef + (size_t)ti * DDPM3D_NCOEF;
    const float c_recip = c[DDPM3D_C_SQRT_RECIP_ACP], c_recipm1 = c[DDPM3D_C_SQRT_RECIPM1_ACP];
    const float c1 = c[DDPM3D_C_POST_MEAN_COEF1], c2 = c[DDPM3D_C_POST_MEAN_COEF2];
    const float min_log = c[DDPM3D_C_MIN_LOG], max_log = c[DDPM3D_C_MAX_LOG];
    const float ab = c[DDPM3D_C_ACP], ab_prev = c[DDPM3D_C_ACP_PREV];
    const bool learn = flags & DDPM3D_F_LEARN_SIGMA;
    const int ch = learn ? 2 : 1;
    const float mask = ti != 0 ? 1.0f : 0.0f;
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < voxels; v += gridDim.x * blockDim.x) {
        const size_t i = (size_t)n * voxels + v;
        const float xv = x[i];
        const float e = mo[((size_t)n * ch) * voxels + v];
        float x0;
        if (flags & DDPM3D_F_PREDICT_XSTART) {
            x0 = e;
        } else {
            x0 = c_recip * xv - c_recipm1 * e;  // gaussian_diffusion.py:330-333
        }
        if (flags & DDPM3D_F_CLIP) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
        float out;
        if (!DDIM) {
            float logvar;
            if (learn) {
                const float vv = mo[((size_t)n * ch + 1) * voxels + v];
                const float frac = (vv + 1.0f) / 2.0f;                      // :274
                logvar = frac * max_log + (1.0f - frac) * min_log;          // :275
            } else {
                logvar = min_log;
            }
            const float mean = c1 * x0 + c2 * xv;                           // :216-219
            out = mean + mask * expf(0.5f * logvar) * noise[i];             // :438
        } else {
            const float eps = (c_recip * xv - x0) / c_recipm1;              // :345-349
            const float sigma = eta * sqrtf((1.0f - ab_prev) / (1.0f - ab)) * sqrtf(1.0f - ab / ab_prev);
            const float mean_pred = x0 * sqrtf(ab_prev) + sqrtf(1.0f - ab_prev - sigma * sigma) * eps;
            out = mean_pred + mask * sigma * noise[i];                      // :584
        }
        sample[i] = out;
        if (pred_xstart != nullptr) pred_xstart[i] = x0;
    }
}

hipError_t ddpm3d_launch_sample_step(bool ddim, const float* mo, const float* x, const float* noise,
                                     const float* coef, const int64_t* t_idx, int N, int voxels, int flags,
                                     float eta, float* sample, float* pred_xstart, hipStream_t st) {
    int bx = (voxels + 255) / 256;
    if (bx > 1024) bx = 1024;
    dim3 grid(bx, N);
    if (ddim)
        hipLaunchKernelGGL(sample_step_kernel<true>, grid, dim3(256), 0, st, mo, x, noise, coef, t_idx, voxels,
                           flags, eta, sample, pred_xstart);
    else
        hipLaunchKernelGGL(sample_step_kernel<false>, grid, dim3(256), 0, st, mo, x, noise, coef, t_idx,
                           voxels, flags, eta, sample, pred_xstart);
    return hipGetLastError();
}
