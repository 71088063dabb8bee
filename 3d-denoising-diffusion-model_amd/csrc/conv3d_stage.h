// Lean staging of one (128-voxel output tile -- 8x8x2 or 4x4x8 --, 16-channel chunk) item of the Winograd-D convolutions:
// raw input planes -> GroupNorm/FiLM affine -> SiLU -> F(2,3) input transform along depth ->
// f16 hi/lo split -> LDS image, for conv3d_wz.h (split_pair also serves conv3d_skinny.hip).
//
// Why this file exists (r02, profiles/r02_wzs_stamps_*.txt): in r01's wave-specialised kernel the LOADER waves
// were the critical path -- 8.2k cycles of staging per item against 7.7k cycles of tap loop, the
// compute waves parked at the item barrier for 19-35 % of their life.  A wave issues in order at
// ~4-5 cycles per instruction of ANY kind, and the staging code as compiled was ~1000 VALU plus ~700
// scalar / v_readlane instructions per item (integer division by the chunk count, 64-bit address
// chains, spilled SGPRs, per-value selects, packed-f32 ops that cost 3x beside MFMAs).  Here it is
// ~330 VALU and a dozen scalar instructions:
//   * every per-lane address is a 32-bit offset computed ONCE; per item only four scalar plane
//     offsets change (buffer loads: uniform descriptor + lane offset + scalar offset);
//   * out-of-volume (y, x) lanes never store: their LDS slots are zeroed once per workgroup (the
//     tile column, hence the set of such lanes, is fixed); out-of-volume PLANES are a uniform skip;
//   * "no activation" is exp2(-126) = 0 in the sigmoid's denominator instead of a per-value select;
//   * the pre-split power-of-two scale (act_scale(), conv3d_load.h) is folded into the affine (exact);
//   * hi = cvt_pk(s), lo = cvt_pk(fma_mix(-hi + s)): 2 instructions per value instead of ~6;
//   * scalar float code throughout (no f32x4 arithmetic: it lowers to v_pk_*_f32).
// Value for value the arithmetic is the one of r01's staging (same roundings in the same order), so
// the kernels stay bit-identical to each other and to the committed goldens' tolerances.
#pragma once
#include "conv3d_epilogue.h"

// arithmetic of the staged operands: hi + lo f16 (three MFMAs per product), hi f16 only, bf16
enum { WZ_F16X3 = 0, WZ_F16 = 1, WZ_BF16 = 2 };

// two fp32 -> packed f16 hi (RNE) and packed f16 lo = f16(s - hi); the subtraction is exact
__device__ __forceinline__ void split_pair(float s0, float s1, unsigned& hi, unsigned& lo) {
    float l0, l1;
    asm("v_cvt_pk_f16_f32 %0, %3, %4\n\t"
        "v_fma_mix_f32 %1, %0, -1.0, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %2, %0, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(hi), "=&v"(l0), "=&v"(l1)
        : "v"(s0), "v"(s1));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(l0), "v"(l1));
}

// Geometry of a workgroup tile's LDS image: TX_ x TY_ (x, y) positions x NZP z-pairs = 128 output voxels.
//   8 x 8: ONE z-pair (8x8x2): 4 transformed planes of 10x10 voxels, 80 B per voxel
//   8 x 4: TWO z-pairs (8x4x4; r03): 2 pair images of 4 transformed planes of 10x6 voxels; an item is 6 input planes
//          and 240 staging slots -- one per thread, six planes deep, against two slots x four planes on 156 of the
//          256 threads of the 8x8x2 form: a quarter less staging on the critical path, 10 % fewer SiLUs and halo bytes
//   4 x 4: FOUR z-pairs (4x4x8, the levels below 8x8; r03): 4 pair images of 6x6 voxels, 10 input planes per item
// The pairs' input planes overlap (two new planes per pair).  A pair image's size is a multiple of 256 B, so rows of
// different pairs keep the bank residues of LdsGeom.
template <int TX_, int TY_ = TX_>
struct WzGeomT {
    static constexpr int CK = DDPM3D_CONV_CK;
    static constexpr int TX = TX_, TY = TY_, HX = TX_ + 2, HY = TY_ + 2, NP = 4, VS = 5;
    static constexpr int RP = TX_ * TY_;               // GEMM rows (positions) per z-pair
    static constexpr int NZP = 64 / RP;                // z-pairs per tile
    static constexpr int NPL = 2 * NZP + 2;            // input planes per item
    static constexpr int RY = LdsGeom<TX_, HX, HY>::RY;
    static constexpr int RZ = LdsGeom<TX_, HX, HY>::RZ;
    static constexpr int PAIR = NP * RZ * 16;          // bytes of one z-pair's image
    static constexpr int BUF = NZP * PAIR;             // bytes of the tile's image
    static constexpr int HC = HX * HY * (CK / 4);      // staging slots: (y, x, channel quad)
    static constexpr int NL = (HC + 255) / 256;        // slots per staging thread
    static_assert(RP * NZP == 64 && PAIR % 256 == 0, "128-voxel tiles; pair images keep the bank residues");
};
typedef WzGeomT<8, 8> WzGeom;

// Per-thread, launch-invariant part of the staging (256 staging threads, thread = lt).
template <class G>
struct StageLaneT {
    unsigned vo0[G::NL], vo1[G::NL];             // byte offset of slot i's voxel at plane zb in src0 / src1
    unsigned es0, es1;                           // bytes per element of src0 / src1 (4, or 2 = bf16 / f16)
    bool f16;                                    // the 2-byte elements are IEEE f16, not bf16
    int lds[G::NL];                              // byte offset of slot i inside a plane of the image
    bool ok[G::NL];                              // slot exists and its (y, x) lies inside the volume
    unsigned plane0, plane1;                     // bytes per z-plane of src0 / src1
    int zb;                                      // first source plane the offsets refer to
    float scale;                                 // activation scale S of this sample (power of two)
    float km, ka;                                // sigmoid exponent = fma(yS, km, ka)
    int q;
};
typedef StageLaneT<WzGeom> StageLane;

// zb = the lowest input plane any item of this workgroup reads (clamped to 0)
template <class G = WzGeom>
__device__ __forceinline__ StageLaneT<G> stage_lane(const ConvK& p, int lt, int n, int y0, int x0, int zb, float scale) {
    StageLaneT<G> s;
    s.scale = scale;
    const int up = p.in_mode == DDPM3D_IN_UP ? 1 : 0;
    const int Hs = p.H >> up, Ws = p.W >> up;
    s.q = lt & 3;
    s.zb = zb;
    s.es0 = (p.io & DDPM3D_IO_SRC0_BF16) ? 2u : 4u;
    s.es1 = (p.io & DDPM3D_IO_SRC1_BF16) ? 2u : 4u;
    s.f16 = (p.io & DDPM3D_IO_HALF_IS_F16) != 0;
    s.plane0 = (unsigned)(Hs * Ws) * (unsigned)p.C0 * s.es0;
    s.plane1 = (unsigned)(Hs * Ws) * (unsigned)p.C1 * s.es1;
    // act: e = exp2(-y log2e) with yS = S y;  none: e = exp2(-126) ~ 1e-38, 1 + e == 1, yS * 1 = yS
    s.km = p.act ? -1.44269504088896341f / scale : 0.0f;
    s.ka = p.act ? 0.0f : -126.0f;
#pragma unroll
    for (int i = 0; i < G::NL; ++i) {
        const int idx = lt + i * 256;
        const int hyx = idx >> 2;
        const int hy = hyx / G::HX, hx = hyx - hy * G::HX;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        s.ok[i] = idx < G::HC && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        const unsigned vox = (unsigned)(((n * p.D + zb) * Hs + (y >> up)) * Ws + (x >> up));
        s.vo0[i] = s.ok[i] ? (vox * (unsigned)p.C0 + s.q * 4u) * s.es0 : DDPM3D_OOB_OFFSET;
        s.vo1[i] = s.ok[i] ? (vox * (unsigned)p.C1 + s.q * 4u) * s.es1 : DDPM3D_OOB_OFFSET;
        s.lds[i] = (hy * G::RY + hx * G::VS) * 16 + s.q * 8;
    }
    return s;
}

// the activation scale, once it is known (stage_lane was given 1)
template <class G>
__device__ __forceinline__ void stage_lane_scale(const ConvK& p, StageLaneT<G>& s, float scale) {
    s.scale = scale;
    s.km = p.act ? -1.44269504088896341f / scale : 0.0f;
}

// Slots of halo voxels outside H x W are zero for every item: written once, never again.
// nimages = tile images at lds_images (each G::NZP pair images)
template <class G>
__device__ __forceinline__ void stage_zero_border(const StageLaneT<G>& s, unsigned char* lds_images, int nimages, int lt) {
#pragma unroll
    for (int i = 0; i < G::NL; ++i) {
        if (lt + i * 256 < G::HC && !s.ok[i]) {
            for (int b = 0; b < nimages * G::NZP; ++b)
#pragma unroll
                for (int j = 0; j < G::NP; ++j) {
                    unsigned char* v = lds_images + b * G::PAIR + j * G::RZ * 16 + s.lds[i];
                    *reinterpret_cast<u32x2*>(v) = u32x2{0u, 0u};
                    *reinterpret_cast<u32x2*>(v + 32) = u32x2{0u, 0u};
                }
        }
    }
}

// One item's raw data: the NPL = 2 * NZP + 2 input planes z0-1 .. z0+2*NZP of the thread's slots (NZP
// consecutive z-pairs share their inner planes) + the chunk's affine.
template <class G>
struct StageRawT {
    u32x4 v[G::NL][G::NPL];   // raw bits: four fp32, or four bf16 in the low half
    f32x4 ga, gb;
    unsigned zmask;   // bit k: plane z0 - 1 + k lies inside the volume (uniform)
    bool b16;         // this item's source holds bf16 (uniform)
};
typedef StageRawT<WzGeom> StageRaw;

// issue the loads of item (first z-pair z0, channel chunk `chunk`)
template <class G>
__device__ __forceinline__ void stage_issue(const ConvK& p, const StageLaneT<G>& s, StageRawT<G>& r, int n, int z0, int chunk) {
    const int c0 = chunk * G::CK;
    const bool from0 = c0 < p.C0;
    const __amdgpu_buffer_rsrc_t rs = from0 ? make_rsrc(p.src0, p.src0_bytes) : make_rsrc(p.src1, p.src1_bytes);
    const unsigned plane = from0 ? s.plane0 : s.plane1;
    const unsigned cb4 = (unsigned)(from0 ? c0 : c0 - p.C0) * (from0 ? s.es0 : s.es1);
    r.b16 = (from0 ? s.es0 : s.es1) == 2u;
    r.zmask = 0;
#pragma unroll
    for (int k = 0; k < G::NPL; ++k) {
        const int z = z0 - 1 + k;
        if ((unsigned)z < (unsigned)p.D) {
            r.zmask |= 1u << k;
            const unsigned soff = cb4 + (unsigned)(z - s.zb) * plane;
#pragma unroll
            for (int i = 0; i < G::NL; ++i)
                r.v[i][k] = buffer_load_quad(rs, from0 ? s.vo0[i] : s.vo1[i], soff, r.b16);
        }
    }
    if (p.affA != nullptr) {
        r.ga = *reinterpret_cast<const f32x4*>(p.affA + (size_t)n * p.Cin + c0 + s.q * 4);
        r.gb = *reinterpret_cast<const f32x4*>(p.affB + (size_t)n * p.Cin + c0 + s.q * 4);
    } else {
        r.ga = f32x4{1.f, 1.f, 1.f, 1.f};
        r.gb = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// raw -> image at `buf` (z-pair zp at buf + zp * G::PAIR).  MODE WZ_F16 keeps only the hi
// halves; WZ_BF16 stores bf16 pairs in the hi slots (s.scale is 1 then).
template <int MODE, class G>
__device__ __forceinline__ void stage_write(const StageLaneT<G>& s, const StageRawT<G>& r, unsigned char* buf) {
    float sa[4], sb[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        sa[c] = r.ga[c] * s.scale;
        sb[c] = r.gb[c] * s.scale;
    }
#pragma unroll
    for (int i = 0; i < G::NL; ++i) {
        // S * act(A x + B) of input plane k
        auto plane = [&](const int k, float (&d)[4]) {
            if (r.zmask & (1u << k)) {
                const f32x4 x = quad_bits_expand(r.v[i][k], r.b16, s.f16);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float ys = __builtin_fmaf(x[c], sa[c], sb[c]);
#ifdef DDPM3D_WZ_NOSILU   // measurement only (wrong results): what the two transcendentals per value cost
                    d[c] = ys;
#else
                    const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(ys, s.km, s.ka));
                    d[c] = ys * __builtin_amdgcn_rcpf(1.0f + e);
#endif
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) d[c] = 0.0f;
            }
        };
        // a window of four planes slides over the item's z-pairs (two new planes per pair)
        float d[4][4];
        plane(0, d[0]);
        plane(1, d[1]);
#pragma unroll
        for (int zp = 0; zp < G::NZP; ++zp) {
            const int a = (2 * zp) & 3, b1 = (2 * zp + 1) & 3, b2 = (2 * zp + 2) & 3, b3 = (2 * zp + 3) & 3;
            plane(2 * zp + 2, d[b2]);
            plane(2 * zp + 3, d[b3]);
            if (s.ok[i]) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v[4];   // |v| < 2^15 by the choice of S: nothing to clamp
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        v[c] = j == 0 ? d[a][c] - d[b2][c]
                             : j == 1 ? d[b1][c] + d[b2][c]
                             : j == 2 ? d[b2][c] - d[b1][c]
                                      : d[b1][c] - d[b3][c];
                    unsigned char* vrow = buf + zp * G::PAIR + j * G::RZ * 16 + s.lds[i];
                    if constexpr (MODE == WZ_BF16) {
                        *reinterpret_cast<u32x2*>(vrow) = u32x2{bf16_pack(v[0], v[1]), bf16_pack(v[2], v[3])};
                    } else {
                        unsigned h0, h1, l0, l1;
                        split_pair(v[0], v[1], h0, l0);
                        split_pair(v[2], v[3], h1, l1);
                        *reinterpret_cast<u32x2*>(vrow) = u32x2{h0, h1};
                        if (MODE == WZ_F16X3) *reinterpret_cast<u32x2*>(vrow + 32) = u32x2{l0, l1};
                    }
                }
            }
        }
    }
}

