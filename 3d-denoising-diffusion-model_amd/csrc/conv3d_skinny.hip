// 3x3x3 convolution with ONE or TWO output channels: the network's last layer (unet.py:995-997,
// out = conv(silu(GroupNorm(h))): 128 -> 2 at full resolution).
//
// The general kernels pad Cout to a 32-wide MFMA tile: 128 -> 2 @ 64^3 ran 16x the needed MFMAs
// (0.22 ms, 1.6 % of the forward for 0.06 % of its FLOPs).  Here the roles are turned
// around: per INPUT voxel the 27 x Cout products with the weight taps are one small GEMM
//     P[v][tap, co] = sum_ci a[v][ci] * w[co][ci][tap]          (K = Cin, N = 27 * Cout <= 64)
// and an output voxel gathers its 27 contributions  out[v][co] = sum_tap P[v + tap][tap, co].
// A workgroup owns an 8x8 column of ZSEG output planes and marches along depth: per input plane
// its four waves compute P for the 10x10 halo (A fragments straight from global memory through the
// GroupNorm affine + SiLU + hi/lo split: a lane's 8 channels of a voxel are 32 contiguous bytes; B
// fragments = the packed weight image of the general kernels, regrouped once into LDS), write it to
// LDS, and 64 x Cout threads add its three depth taps into a register ring of output planes.
// K order: the MFMA's k-step s pairs lane half h with channels h * Cin/2 + 8 s .. + 7, so a lane walks
// 4 * Cin/2 CONTIGUOUS bytes of its voxel over the k-steps (consecutive loads share 64-byte sectors; with
// the natural order -- 8 channels of chunk s -- four loads far apart shared one and L1 thrashed); the
// weights are regrouped into the same order when they are copied to LDS.
// MFMAs per output voxel: 8.6x fewer than the padded form; the input is read once per (8x8, ZSEG)
// column with a 10/8 x 10/8 x (ZSEG+2)/ZSEG halo, from L2 mostly.  Bound: HBM (the input tensor,
// Cin * 4 bytes per voxel; the output is Cout * 4).
#include "conv3d_stage.h"

namespace {

constexpr int SK_TX = 8, SK_HX = 10, SK_HALO = SK_HX * SK_HX;   // 8x8 outputs, 10x10 halo voxels
constexpr int SK_COLS = 64;                                      // 27 * Cout padded to two MFMA column tiles
constexpr int SK_PSTRIDE = SK_COLS + 1;                          // P row stride in floats
#ifndef SK_MIN_WGS
#define SK_MIN_WGS 512
#endif

// one plane of a lane's halo voxel: per k-step its 8 channels (two 16-byte quads of fp32, or one of bf16)
template <int NCH, int B16>
__device__ __forceinline__ void sk_issue(__amdgpu_buffer_rsrc_t rs, unsigned vo, unsigned soff_, u32x4 (&raw)[NCH][2]) {
    // (forced uniform: the compiler keeps it in a VGPR otherwise and wraps every load in a waterfall loop)
    const unsigned soff = __builtin_amdgcn_readfirstlane(soff_);
#pragma unroll
    for (int s = 0; s < NCH; ++s) {
        const unsigned o = soff + (unsigned)s * 8u * (B16 ? 2u : 4u);
        raw[s][0] = buffer_load16(rs, vo, o);
        raw[s][1] = B16 ? raw[s][0] : buffer_load16(rs, vo, o + 16);
    }
}

// NCH = Cin / 16 and B16 (bf16 source tensor) are compile-time: with run-time tests per k-step the
// compiler split the plane loop into blocks joined by 64 accumulator copies each and selected the
// element type per value -- as many instructions again as the arithmetic.
template <int MODE, int NCH, int B16>
__global__ __launch_bounds__(256, 2) void conv3d_skinny_kernel(const ConvK p, int zseg, int zsegs) {
    constexpr bool X3 = MODE == WZ_F16X3;
    constexpr int CK = DDPM3D_CONV_CK, CINP = NCH * CK;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    constexpr int nchunks = NCH;
    const int ncols = 27 * p.Cout;

    // LDS: weights [chunk][hi|lo][64 cols][16] f16 | affine table [Cin][2] | P [128 rows][65]
    unsigned char* wl = lds;
    float* aff = reinterpret_cast<float*>(lds + (size_t)nchunks * 2 * SK_COLS * 32);
    float* P = aff + 2 * CINP;

    int tile = blockIdx.x;
    const int tilesX = (p.W + SK_TX - 1) / SK_TX, tilesY = (p.H + SK_TX - 1) / SK_TX;
    const int tx_i = tile % tilesX; tile /= tilesX;
    const int ty_i = tile % tilesY; tile /= tilesY;
    const int zs_i = tile % zsegs; tile /= zsegs;
    const int n = tile;
    const int x0 = tx_i * SK_TX, y0 = ty_i * SK_TX, z0 = zs_i * zseg;
    const int zend = min(z0 + zseg, p.D);

    ActScale asc = {1.0f, 1.0f};
    if constexpr (MODE != WZ_BF16) asc = act_scale(p, n, 1.0f);

    // ---- once per workgroup: weights and the (scaled) affine into LDS
    {
        const unsigned wpart = (unsigned)p.CoutPad * 32;                  // bytes of one hi (or lo) block
        const unsigned wtap = (unsigned)nchunks * 2 * wpart;              // bytes per tap
        const int items = nchunks * 2 * SK_COLS * 2;                      // 16-byte pieces
        for (int i = tid; i < items; i += 256) {
            const int kh = i & 1, col = (i >> 1) & (SK_COLS - 1), hl = (i >> 7) & 1, s = i >> 8;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (col < ncols) {
                const int tap = col / p.Cout, co = col - tap * p.Cout;
                const int c0 = kh * (CINP / 2) + s * 8;           // first channel of (k-step s, lane half kh)
                v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(p.w) + (size_t)tap * wtap +
                                                    (size_t)((c0 >> 4) * 2 + hl) * wpart + co * 32 + ((c0 >> 3) & 1) * 16);
            }
            *reinterpret_cast<u32x4*>(wl + ((size_t)(s * 2 + hl) * SK_COLS + col) * 32 + kh * 16) = v;
        }
        for (int c = tid; c < CINP; c += 256) {
            float a = 1.0f, b = 0.0f;
            if (p.affA != nullptr && c < p.Cin) {
                a = p.affA[(size_t)n * p.Cin + c];
                b = p.affB[(size_t)n * p.Cin + c];
            }
            aff[2 * c] = a * asc.s;
            aff[2 * c + 1] = b * asc.s;
        }
    }
    // SiLU or identity without a select (conv3d_stage.h): e = exp2(fma(yS, km, ka))
    const float km = p.act ? -1.44269504088896341f / asc.s : 0.0f;
    const float ka = p.act ? 0.0f : -126.0f;

    // ---- this lane's halo voxel (GEMM row) and its byte offset inside a depth plane
    const int r = wave * 32 + (lane & 31);
    const int hy = r / SK_HX, hx = r - hy * SK_HX;
    const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
    const bool rok = r < SK_HALO && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
    constexpr unsigned es = B16 ? 2u : 4u;
    const unsigned vo = rok ? ((unsigned)(yy * p.W + xx) * (unsigned)p.Cin + half * (unsigned)(CINP / 2)) * es : DDPM3D_OOB_OFFSET;
    const unsigned plane_bytes = (unsigned)(p.H * p.W) * (unsigned)p.Cin * es;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.src0, p.src0_bytes);

    // ---- the gathering threads: (output position, cout) and their ring of three output planes
    const int gpos = tid / p.Cout, gco = tid - gpos * p.Cout;
    const bool gat = tid < 64 * p.Cout;
    const int gy = gpos >> 3, gx = gpos & 7;
    float s_prev = 0.0f, s_cur = 0.0f, s_next = 0.0f;
    // Conv3d zero-pads the ACTIVATED tensor: halo voxels outside H x W contribute nothing.  Their P rows
    // hold act(B) * w (the loads return 0), so the gather masks them (cheaper than zeroing A: 9 values
    // per gathering thread instead of one multiply per element)
    float gm[3][3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
            gm[dy][dx] = ((unsigned)(y0 + gy + dy - 1) < (unsigned)p.H && (unsigned)(x0 + gx + dx - 1) < (unsigned)p.W) ? 1.0f : 0.0f;
    const float oscale = gat ? p.wscale[gco] * asc.inv : 0.0f;
    const float bias = gat ? p.bias[(size_t)n * p.bias_stride_n + gco] : 0.0f;
    const size_t DHW = (size_t)p.D * p.H * p.W;

    // software pipeline over the planes: the loads of plane zi + 1 are in flight while plane zi is
    // activated, split and multiplied (64 more registers; the LDS footprint allows two workgroups
    // per CU either way)
    u32x4 cur[NCH][2], nxt[NCH][2];
    sk_issue<NCH, B16>(rs, vo, (unsigned)(n * p.D + max(z0 - 1, 0)) * plane_bytes, cur);
    __syncthreads();
    for (int zi = z0 - 1; zi <= zend; ++zi) {
        const bool inplane = (unsigned)zi < (unsigned)p.D;
        if (inplane) {
            if (zi + 1 <= zend && zi + 1 < p.D)
                sk_issue<NCH, B16>(rs, vo, (unsigned)(n * p.D + zi + 1) * plane_bytes, nxt);
            // P[row][col] of this plane: per 16-channel chunk one A fragment (8 channels per lane) x 2 column tiles
            f32x16 acc[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[ct][i] = 0.0f;
#pragma unroll
            for (int s = 0; s < NCH; ++s) {
                {
                    const f32x4 q0 = B16 ? half4_expand(u32x2{cur[s][0][0], cur[s][0][1]}, B16 == 2) : __builtin_bit_cast(f32x4, cur[s][0]);
                    const f32x4 q1 = B16 ? half4_expand(u32x2{cur[s][0][2], cur[s][0][3]}, B16 == 2) : __builtin_bit_cast(f32x4, cur[s][1]);
                    const float* at = aff + 2 * (half * (CINP / 2) + s * 8);
                    float v[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const float xc = c < 4 ? q0[c & 3] : q1[c & 3];
                        const float ys = __builtin_fmaf(xc, at[2 * c], at[2 * c + 1]);
                        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(ys, km, ka));
                        v[c] = ys * __builtin_amdgcn_rcpf(1.0f + e);
                    }
                    unsigned hi[4], lo[4];
                    if constexpr (MODE == WZ_BF16) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) hi[c] = bf16_pack(v[2 * c], v[2 * c + 1]);
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) split_pair(v[2 * c], v[2 * c + 1], hi[c], lo[c]);
                    }
                    const h8 ahi = __builtin_bit_cast(h8, u32x4{hi[0], hi[1], hi[2], hi[3]});
                    const unsigned char* wb = wl + (size_t)s * 2 * SK_COLS * 32 + (lane & 31) * 32 + half * 16;
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        const h8 bhi = *reinterpret_cast<const h8*>(wb + ct * 32 * 32);
                        if constexpr (X3) {
                            const h8 alo = __builtin_bit_cast(h8, u32x4{lo[0], lo[1], lo[2], lo[3]});
                            const h8 blo = *reinterpret_cast<const h8*>(wb + SK_COLS * 32 + ct * 32 * 32);
                            acc[ct] = mfma16<false>(alo, bhi, acc[ct]);
                            acc[ct] = mfma16<false>(ahi, blo, acc[ct]);
                        }
                        acc[ct] = mfma16<MODE == WZ_BF16>(ahi, bhi, acc[ct]);
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < NCH; ++s) {
                cur[s][0] = nxt[s][0];
                cur[s][1] = nxt[s][1];
            }
            // C layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * half
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int row = wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                    P[row * SK_PSTRIDE + ct * 32 + (lane & 31)] = acc[ct][reg];
                }
        }
        __syncthreads();
        if (gat) {
            if (inplane) {
                // input plane zi is depth tap dz of output plane zi + 1 - dz
                float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float* pr = P + ((gy + dy) * SK_HX + gx + dx) * SK_PSTRIDE + gco;
                        t0 = __builtin_fmaf(pr[((0 * 3 + dy) * 3 + dx) * p.Cout], gm[dy][dx], t0);
                        t1 = __builtin_fmaf(pr[((1 * 3 + dy) * 3 + dx) * p.Cout], gm[dy][dx], t1);
                        t2 = __builtin_fmaf(pr[((2 * 3 + dy) * 3 + dx) * p.Cout], gm[dy][dx], t2);
                    }
                s_next = t0;
                s_cur += t1;
                s_prev += t2;
            } else {
                s_next = 0.0f;
            }
            // output plane zi - 1 has all three depth taps now
            const int zo = zi - 1, y = y0 + gy, x = x0 + gx;
            if (zo >= z0 && zo < zend && y < p.H && x < p.W) {
                const float val = s_prev * oscale + bias;
                const size_t vox = ((size_t)zo * p.H + y) * p.W + x;
                if (p.out_layout == DDPM3D_OUT_NDHWC)
                    ddpm3d_act_store(p.out, ((size_t)n * DHW + vox) * p.Cout + gco, val, (p.io & DDPM3D_IO_OUT_BF16) != 0,
                                     (p.io & DDPM3D_IO_HALF_IS_F16) != 0);
                else
                    p.out[((size_t)n * p.Cout + gco) * DHW + vox] = val;
            }
            s_prev = s_cur;
            s_cur = s_next;
        }
        __syncthreads();
    }
}

}  // namespace

static size_t skinny_lds_bytes(int CinPad) {
    return (size_t)(CinPad / DDPM3D_CONV_CK) * 2 * SK_COLS * 32 + (size_t)2 * CinPad * 4 + (size_t)128 * SK_PSTRIDE * 4;
}

template <int MODE, int NCH, int B16>
static hipError_t sk_launch(const ConvK& k, dim3 grid, int zseg, int zsegs, hipStream_t st) {
    const size_t lds = skinny_lds_bytes(NCH * DDPM3D_CONV_CK);
    if (lds > 65536) {
        static DynLdsOnce once = {};
        const hipError_t a = ddpm3d_allow_dynamic_lds(once, reinterpret_cast<const void*>(&conv3d_skinny_kernel<MODE, NCH, B16>),
                                                      (int)lds);
        if (a != hipSuccess) return a;
    }
    hipLaunchKernelGGL((conv3d_skinny_kernel<MODE, NCH, B16>), grid, dim3(256), lds, st, k, zseg, zsegs);
    return hipGetLastError();
}

// Instantiated: Cin = 32, 64, 128 channels (base widths); fp32 sources in the f16x3 / f16 arithmetic, bf16
// sources in the bf16 mode and f16 sources in the f16 mode (the modes whose residual stream is 16-bit).
// Anything else stays on the general kernel (api.hip asks here).  src16: 0 fp32, 1 bf16, 2 f16.
bool ddpm3d_skinny_ok(int CinPad, int prec, int src16) {
    if (CinPad != 32 && CinPad != 64 && CinPad != 128) return false;
    if (prec == DDPM3D_PREC_BF16) return src16 == 1;
    if (prec == DDPM3D_PREC_F16) return src16 == 0 || src16 == 2;
    return prec == DDPM3D_PREC_F16X3 && src16 == 0;
}

template <int MODE, int B16>
static hipError_t sk_launch_mode(const ConvK& k, dim3 grid, int zseg, int zsegs, hipStream_t st) {
    if (k.CinPad == 32) return sk_launch<MODE, 2, B16>(k, grid, zseg, zsegs, st);
    if (k.CinPad == 64) return sk_launch<MODE, 4, B16>(k, grid, zseg, zsegs, st);
    return sk_launch<MODE, 8, B16>(k, grid, zseg, zsegs, st);
}

hipError_t ddpm3d_launch_conv_skinny(const ConvK& k, int prec, hipStream_t st) {
    // depth segments: as long as possible (less halo) while the grid still fills the chip twice over
    const int tiles = k.N * ((k.H + 7) / 8) * ((k.W + 7) / 8);
    int zseg = k.D;
    while (zseg > 4 && (long long)tiles * ((k.D + zseg - 1) / zseg) < SK_MIN_WGS) zseg = (zseg + 1) / 2;
    const int zsegs = (k.D + zseg - 1) / zseg;
    const dim3 grid(tiles * zsegs);
    if (prec == DDPM3D_PREC_F16) {
        if (k.io & DDPM3D_IO_SRC0_BF16) return sk_launch_mode<WZ_F16, 2>(k, grid, zseg, zsegs, st);   // f16 source
        return sk_launch_mode<WZ_F16, 0>(k, grid, zseg, zsegs, st);
    }
    if (prec == DDPM3D_PREC_BF16) return sk_launch_mode<WZ_BF16, 1>(k, grid, zseg, zsegs, st);
    return sk_launch_mode<WZ_F16X3, 0>(k, grid, zseg, zsegs, st);
}
