// Shared prologue of the conv kernels: fetch one (halo voxel, 4-channel quad)
// of the conv's INPUT, i.e. the source tensor(s) seen through the virtual
// concat, the pool / upsample / planar mode, the GroupNorm(+FiLM) affine and
// SiLU.  Out-of-bounds voxels are exact zeros (Conv3d zero-pads the tensor
// AFTER norm+activation).
#pragma once
#include "conv3d_params.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// FAST = false: expf + IEEE divide (<= 1 ulp each), for the exact-fp32 path.
// FAST = true : v_exp_f32 / v_rcp_f32 based (~2 ulp), for the split-f16 path
//               whose operand representation error is of the same order.
template <bool FAST>
__device__ __forceinline__ float silu_f(float v) {
    // (not __fdividef: hipcc expands that to the full div_scale/div_fmas/div_fixup sequence,
    // ~10 instructions -- it was a third of the staging VALU of the f16x3 kernels)
    if (FAST) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
    return v / (1.0f + expf(-v));
}

template <int ACT, bool FAST>
__device__ __forceinline__ f32x4 affine_act(f32x4 v, f32x4 a, f32x4 b) {
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float y = fmaf(v[i], a[i], b[i]);
        r[i] = ACT ? silu_f<FAST>(y) : y;
    }
    return r;
}

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// The 16-bit storage of an activation tensor (ddpm3d_conv_desc.io_dtype): bf16, or -- with
// DDPM3D_IO_HALF_IS_F16, the reference's own --use_fp16 storage (unet.py:1035, fp16_util.py:15-22) --
// IEEE f16.  `f16` is launch-uniform everywhere below.
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
// four bf16 (two dwords) -> four fp32
__device__ __forceinline__ f32x4 bf16x4_expand(u32x2 v) {
    f32x4 r;
    r[0] = __builtin_bit_cast(float, v[0] << 16);
    r[1] = __builtin_bit_cast(float, v[0] & 0xffff0000u);
    r[2] = __builtin_bit_cast(float, v[1] << 16);
    r[3] = __builtin_bit_cast(float, v[1] & 0xffff0000u);
    return r;
}
// two fp32 -> packed bf16 pair (round to nearest even, NaN stays NaN)
__device__ __forceinline__ unsigned bf16_pack(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// four f16 (two dwords) -> four fp32 (exact)
__device__ __forceinline__ f32x4 f16x4_expand(u32x2 v) {
    return __builtin_convertvector(__builtin_bit_cast(h4v, v), f32x4);
}
// two fp32 -> packed f16 pair (round to nearest even; beyond 65504 -> inf, as torch's .half())
__device__ __forceinline__ unsigned f16_pack(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ f32x4 half4_expand(u32x2 v, bool f16) { return f16 ? f16x4_expand(v) : bf16x4_expand(v); }
__device__ __forceinline__ unsigned half_pack(float lo, float hi, bool f16) { return f16 ? f16_pack(lo, hi) : bf16_pack(lo, hi); }
// quad q4 (4 channels starting at element offset e) of an fp32 or 16-bit activation tensor
__device__ __forceinline__ f32x4 act_quad(const float* base, size_t e, bool b16, bool f16) {
    if (b16) return half4_expand(*reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned short*>(base) + e), f16);
    return *reinterpret_cast<const f32x4*>(base + e);
}

struct HaloSrc {
    const float* src;  // source tensor of this chunk (after the concat split)
    bool b16;          // it holds 16-bit elements (ddpm3d_conv_desc.io_dtype) ...
    bool f16;          // ... which are IEEE f16 rather than bf16
    unsigned src_bytes;  // its extent (buffer descriptor num_records)
    int Cs;            // its channel count
    int cb;            // first channel of the chunk inside it
    int Hs, Ws;        // its H, W
    bool has_aff;
    f32x4 ga, gb;      // affine of this thread's quad
};

template <int CK>
__device__ __forceinline__ HaloSrc halo_src(const ConvK& p, int n, int chunk, int q) {
    HaloSrc h;
    const int c0 = chunk * CK;
    const bool from0 = c0 < p.C0;
    h.src = from0 ? p.src0 : p.src1;
    h.b16 = (p.io & (from0 ? DDPM3D_IO_SRC0_BF16 : DDPM3D_IO_SRC1_BF16)) != 0;
    h.f16 = (p.io & DDPM3D_IO_HALF_IS_F16) != 0;
    h.src_bytes = from0 ? p.src0_bytes : p.src1_bytes;
    h.Cs = from0 ? p.C0 : p.C1;
    h.cb = from0 ? c0 : c0 - p.C0;
    const bool dbl = p.in_mode == DDPM3D_IN_POOL || p.in_mode == DDPM3D_IN_STRIDE2;
    h.Hs = dbl ? 2 * p.H : (p.in_mode == DDPM3D_IN_UP ? p.H / 2 : p.H);
    h.Ws = dbl ? 2 * p.W : (p.in_mode == DDPM3D_IN_UP ? p.W / 2 : p.W);
    h.has_aff = p.affA != nullptr && p.in_mode != DDPM3D_IN_PLANAR2;
    h.ga = f32x4{1.f, 1.f, 1.f, 1.f};
    h.gb = f32x4{0.f, 0.f, 0.f, 0.f};
    if (h.has_aff) {
        h.ga = *reinterpret_cast<const f32x4*>(p.affA + (size_t)n * p.Cin + c0 + q * 4);
        h.gb = *reinterpret_cast<const f32x4*>(p.affB + (size_t)n * p.Cin + c0 + q * 4);
    }
    return h;
}

template <bool FAST>
__device__ __forceinline__ f32x4 halo_fetch(const ConvK& p, const HaloSrc& h, int n, int z, int y, int x,
                                            int q, bool first_chunk) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (p.in_mode == DDPM3D_IN_STRIDE2) {
        // (y, x) are coordinates of the SOURCE grid here (the strided conv's halo is laid out in it)
        if (!((unsigned)z < (unsigned)p.D && (unsigned)y < (unsigned)h.Hs && (unsigned)x < (unsigned)h.Ws)) return v;
        const size_t vox = (((size_t)n * p.D + z) * h.Hs + y) * h.Ws + x;
        v = act_quad(h.src, vox * h.Cs + h.cb + q * 4, h.b16, h.f16);
        if (h.has_aff) v = p.act ? affine_act<1, FAST>(v, h.ga, h.gb) : affine_act<0, FAST>(v, h.ga, h.gb);
        return v;
    }
    const bool inb = (unsigned)z < (unsigned)p.D && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
    if (!inb) return v;
    if (p.in_mode == DDPM3D_IN_SAME) {
        const size_t vox = (((size_t)n * p.D + z) * p.H + y) * p.W + x;
        v = act_quad(h.src, vox * h.Cs + h.cb + q * 4, h.b16, h.f16);
        if (h.has_aff) v = p.act ? affine_act<1, FAST>(v, h.ga, h.gb) : affine_act<0, FAST>(v, h.ga, h.gb);
    } else if (p.in_mode == DDPM3D_IN_UP) {
        const size_t vox = (((size_t)n * p.D + z) * h.Hs + (y >> 1)) * h.Ws + (x >> 1);
        v = act_quad(h.src, vox * h.Cs + h.cb + q * 4, h.b16, h.f16);
        if (h.has_aff) v = p.act ? affine_act<1, FAST>(v, h.ga, h.gb) : affine_act<0, FAST>(v, h.ga, h.gb);
    } else if (p.in_mode == DDPM3D_IN_POOL) {
        // AvgPool3d window order (h, w): ((s00 + s01) + s10) + s11, then * 1/4
        const size_t vox = (((size_t)n * p.D + z) * h.Hs + 2 * y) * h.Ws + 2 * x;
        const size_t e0 = vox * h.Cs + h.cb + q * 4;
        f32x4 s00 = act_quad(h.src, e0, h.b16, h.f16);
        f32x4 s01 = act_quad(h.src, e0 + h.Cs, h.b16, h.f16);
        f32x4 s10 = act_quad(h.src, e0 + (size_t)h.Ws * h.Cs, h.b16, h.f16);
        f32x4 s11 = act_quad(h.src, e0 + (size_t)h.Ws * h.Cs + h.Cs, h.b16, h.f16);
        if (h.has_aff) {
            if (p.act) {
                s00 = affine_act<1, FAST>(s00, h.ga, h.gb); s01 = affine_act<1, FAST>(s01, h.ga, h.gb);
                s10 = affine_act<1, FAST>(s10, h.ga, h.gb); s11 = affine_act<1, FAST>(s11, h.ga, h.gb);
            } else {
                s00 = affine_act<0, FAST>(s00, h.ga, h.gb); s01 = affine_act<0, FAST>(s01, h.ga, h.gb);
                s10 = affine_act<0, FAST>(s10, h.ga, h.gb); s11 = affine_act<0, FAST>(s11, h.ga, h.gb);
            }
        }
        v = (((s00 + s01) + s10) + s11) * 0.25f;
    } else {  // PLANAR2: two single-channel volumes = channels 0 and 1 of the first chunk
        if (q == 0 && first_chunk) {
            const size_t vox = (((size_t)n * p.D + z) * p.H + y) * p.W + x;
            v[0] = p.src0[vox];
            v[1] = p.src1[vox];
        }
    }
    return v;
}

// Two-phase form used by the software pipeline (IN_SAME / IN_UP only, where an item is
// ONE 16-byte load): `halo_raw` issues the load for the next chunk while the current
// chunk's MFMAs run; `halo_finish` applies affine + activation when the data is needed.
__device__ __forceinline__ bool halo_inb(const ConvK& p, int z, int y, int x) {
    return (unsigned)z < (unsigned)p.D && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
}

// Buffer addressing (SRD + 32-bit per-lane offset + SCALAR offset): with flat 64-bit
// pointers LLVM hoists one VGPR address pair per unrolled tap / staging item out of the chunk
// loop (54 pairs for the weights alone), spills them and then waits vmcnt(0) behind every
// reload.  A uniform descriptor leaves nothing per-tap to hoist, and an offset beyond
// num_records reads as 0, which is exactly the conv's zero padding (no branch).
#define DDPM3D_OOB_OFFSET 0xFFFFFFF0u

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 buffer_load16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
}
__device__ __forceinline__ u32x2 buffer_load8(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
}
// one channel quad of an fp32 (16 B) or bf16 (8 B) tensor as raw bits; quad_bits_expand() when consumed
__device__ __forceinline__ u32x4 buffer_load_quad(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, bool b16) {
    if (b16) {
        const u32x2 t = buffer_load8(r, voff, soff);
        return u32x4{t[0], t[1], 0u, 0u};
    }
    return buffer_load16(r, voff, soff);
}
__device__ __forceinline__ f32x4 quad_bits_expand(u32x4 bits, bool b16, bool f16 = false) {
    return b16 ? half4_expand(u32x2{bits[0], bits[1]}, f16) : __builtin_bit_cast(f32x4, bits);
}

// Voxel index (in the SOURCE tensor's D x Hs x Ws grid) of a halo item, or -1 outside the
// conv's zero padding.  Launch-invariant per item: computed once, kept in one VGPR.
__device__ __forceinline__ int halo_vox(const ConvK& p, int n, int z, int y, int x, int Hs, int Ws, int up_shift) {
    if (!halo_inb(p, z, y, x)) return -1;
    return ((n * p.D + z) * Hs + (y >> up_shift)) * Ws + (x >> up_shift);
}

// "no affine" is A = 1, B = 0 (exact) and "no activation" is a bit-mask select: a
// launch-invariant branch here would make LLVM unswitch the whole chunk loop.
template <bool FAST>
__device__ __forceinline__ f32x4 halo_finish(const HaloSrc& h, f32x4 raw, bool inb, unsigned act_mask) {
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float y = fmaf(raw[i], h.ga[i], h.gb[i]);
        const float s = silu_f<FAST>(y);
        const unsigned bits = (__builtin_bit_cast(unsigned, s) & act_mask) |
                              (__builtin_bit_cast(unsigned, y) & ~act_mask);
        r[i] = inb ? __builtin_bit_cast(float, bits) : 0.0f;  // out-of-bounds voxels stay exact zeros
    }
    return r;
}

// Power-of-two activation scale of sample n (split-f16 modes; uniform over the workgroup): the
// maximum b of the sample's in_bound entries, times `gain` (2 for the Winograd-D input transform,
// which adds two planes), is brought into [2^14, 2^15): s = 2^(14 - floor(log2(gain * b))).
struct ActScale { float s, inv; };
// (two halves, so that a kernel can request the bound first and fold it behind its other loads)
__device__ __forceinline__ float act_scale_load(const ConvK& p, int n) {
    const int lane = threadIdx.x & 63;
    float b = 0.0f;
    if (lane < p.in_bound_count) b = p.in_bound[((size_t)n * p.in_bound_count + lane) * p.in_bound_stride];
    return b;
}
__device__ __forceinline__ ActScale act_scale_finish(float b, float gain) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) b = fmaxf(b, __shfl_xor(b, o));   // a NaN entry is ignored
    b = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, b))) * gain;
    const int e = (int)((__builtin_bit_cast(unsigned, b) >> 23) & 0xffu) - 127;   // floor(log2 b), b normal
    int k = 14 - e;
    if (!(b > 0.0f) || e == 128) k = 0;          // all-zero input, or a non-finite bound
    k = k < -60 ? -60 : (k > 60 ? 60 : k);
    ActScale r;
    r.s = __builtin_bit_cast(float, (unsigned)(127 + k) << 23);
    r.inv = __builtin_bit_cast(float, (unsigned)(127 - k) << 23);
    return r;
}
__device__ __forceinline__ ActScale act_scale(const ConvK& p, int n, float gain) {
    return act_scale_finish(act_scale_load(p, n), gain);
}

// Which (tile, cout block, K split) a workgroup owns.  Workgroups go to the 8 XCDs round-robin
// in linear-id order (x fastest), and each XCD has its own L2:
//   - default (activations outweigh weights): consecutive TILES on one XCD (xcd_remap), so the
//     halo planes two tiles share are fetched once per XCD;
//   - p.wstat (weights outweigh activations, the low-resolution levels): (cout block, split)
//     fastest, so one XCD keeps meeting the same 1/8 of the weight image.  Measured before
//     (r01, 512->512 @ 64x4x4, split-K): 255 MB of fabric traffic per launch for 4 MB of
//     activations -- eight private copies of the 28 MB weight image.
struct WgId { int tile, cy, split; };
__device__ __forceinline__ int xcd_remap(int bid, int nwg);
__device__ __forceinline__ WgId wg_id(const ConvK& p) {
    WgId w;
    if (p.wstat) {
        const int L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int nyz = gridDim.y * gridDim.z;
        const int yz = L % nyz;
        w.tile = L / nyz;
        w.cy = yz % gridDim.y;
        w.split = yz / gridDim.y;
    } else {
        w.tile = xcd_remap(blockIdx.x, gridDim.x);
        w.cy = blockIdx.y;
        w.split = blockIdx.z;
    }
    return w;
}

// (Tried and dropped, r01: folding the x8 split scale into the affine and forming lo = s - hi
// as one fma saves 60 of 440 staging instructions but measured 2 % SLOWER on the same box.)

// XCD-aware tile order: consecutive tiles (which share halo planes) on one XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}
