// Streaming (flash-style) self-attention for the UNet's AttentionBlock
// (unet.py:296-305 with QKVAttentionLegacy, :337-354) on fp32 MFMA, gfx950.
//
//   qkv  [N][T][heads * 3 * CH]  channels-last output of the qkv 1x1 conv; the legacy
//        layout puts heads outermost, then q | k | v (unet.py:346: reshape(bs*heads, 3*ch, T))
//   out  [N][T][heads * CH]
//   w = softmax_s((q * s)^T (k * s)),  s = CH^-1/4  (:348-352, fp32 softmax);  a = w v^T
//
// The reference materialises the T x T weight matrix (4.3 GB per head at T = 32768); here
// one workgroup owns 128 queries (4 waves x 32) of one (sample, head) and streams the keys
// in tiles of 32 with the running max / sum recurrence.
//
// MFMA formulation (v_mfma_f32_32x32x2_f32, exact fp32): the scores are computed TRANSPOSED,
// S^T[key][query] = K Q^T, so a lane owns one query column: its 16 accumulator registers are
// 16 of the tile's 32 keys, the row max / sum are register-local plus ONE exchange with lane^32,
// and P^T is already in the B-operand layout of the second product O^T[c][query] = V^T P^T
// (an MFMA's k index is summed, so each half simply supplies the keys it holds; the V^T
// operand is read from LDS in the matching key order).  No transpose, no LDS round trip.
#include <hip/hip_runtime.h>
#include "../../include/ddpm3d.h"
#include "ops.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CH>
__global__ __launch_bounds__(256, (CH == 128 ? 2 : 3)) void attention_kernel(const float* __restrict__ qkv, int T, int heads,
                                                        float* __restrict__ out) {
    constexpr int KT = 32;            // keys per tile
    constexpr int LS = CH + 1;        // LDS row stride (floats): odd -> conflict-free column reads
    constexpr int CT = CH / 32;       // 32-channel output tiles
    __shared__ float Ks[KT * LS];
    __shared__ float Vs[KT * LS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qi = lane & 31, half = lane >> 5;
    const int nh = blockIdx.y;                 // sample * heads + head
    const int n = nh / heads, head = nh % heads;
    const int C3 = heads * 3 * CH;
    const float* base = qkv + (size_t)n * T * C3 + (size_t)head * 3 * CH;   // q at +0, k at +CH, v at +2CH
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int tq = q0 + qi;                    // this lane's query
    const bool qvalid = tq < T;

    // Q^T operand: lane (query, half) supplies Q[query][2s + half] at k-step s.  Scale folded
    // into Q once: (q s)(k s) = (q s^2) k, s^2 = CH^-1/2.
    float qreg[CH / 2];
    {
        const float scale = 1.0f / sqrtf((float)CH);
        const float* qrow = base + (size_t)(qvalid ? tq : 0) * C3;
#pragma unroll
        for (int s4 = 0; s4 < CH / 4; ++s4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(qrow + s4 * 4);
            qreg[s4 * 2 + 0] = (half ? v[1] : v[0]) * scale;
            qreg[s4 * 2 + 1] = (half ? v[3] : v[2]) * scale;
        }
    }

    f32x16 oacc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[c][i] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;

    for (int k0 = 0; k0 < T; k0 += KT) {
        __syncthreads();
        // stage K and V tiles: 32 keys x CH floats each, 16-byte loads along the channel
        for (int idx = tid; idx < KT * (CH / 4) * 2; idx += 256) {
            const int which = idx / (KT * (CH / 4));        // 0 = K, 1 = V
            const int r = idx % (KT * (CH / 4));
            const int j = r / (CH / 4), c4 = r % (CH / 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k0 + j < T)
                v = *reinterpret_cast<const f32x4*>(base + (size_t)(k0 + j) * C3 + (which + 1) * CH + c4 * 4);
            float* dst = (which ? Vs : Ks) + j * LS + c4 * 4;
            dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
        }
        __syncthreads();

        // S^T[key][query] = sum_c K[key][c] * Q[query][c]
        f32x16 sacc;
#pragma unroll
        for (int i = 0; i < 16; ++i) sacc[i] = 0.0f;
#pragma unroll
        for (int s = 0; s < CH / 2; ++s)
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[qi * LS + 2 * s + half], qreg[s], sacc, 0, 0, 0);

        // this lane holds keys row(r) = (r&3) + 8*(r>>2) + 4*half of its query's column
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (key >= T) sacc[r] = -INFINITY;
            tmax = fmaxf(tmax, sacc[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m_new = fmaxf(m_run, tmax);           // finite: every tile has >= 1 valid key
        const float alpha = expf(m_run - m_new);          // first tile: exp(-inf) = 0
        float psum = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sacc[r] = expf(sacc[r] - m_new);
            psum += sacc[r];
        }
        psum += __shfl_xor(psum, 32);
        l_run = l_run * alpha + psum;
        m_run = m_new;

        // O^T[c][query] = alpha * O^T + sum_key V[key][c] * P^T[key][query]
#pragma unroll
        for (int c = 0; c < CT; ++c) {
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[c][i] *= alpha;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int key = (s & 3) + 8 * (s >> 2) + 4 * half;   // the key this half holds in register s
                oacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[key * LS + c * 32 + qi], sacc[s], oacc[c], 0, 0, 0);
            }
        }
    }

    if (!qvalid) return;
    const float inv = 1.0f / l_run;
    float* orow = out + ((size_t)n * T + tq) * (heads * CH) + (size_t)head * CH;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // registers 4g..4g+3 are channels c*32 + 8g + 4*half + {0,1,2,3}: one 16-byte store
            f32x4 v = {oacc[c][4 * g] * inv, oacc[c][4 * g + 1] * inv, oacc[c][4 * g + 2] * inv,
                       oacc[c][4 * g + 3] * inv};
            *reinterpret_cast<f32x4*>(orow + c * 32 + 8 * g + 4 * half) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same algorithm on v_mfma_f32_32x32x16_f16 with every operand split x = hi + lo into two f16
// (a*b = lo*hi + hi*lo + hi*hi, each f16 x f16 product exact in fp32: the convs' f16x3 arithmetic,
// conv3d.hip) -- 16/3 of the fp32 MFMA rate; r01 config 5: the fp32 kernel was 43 % of a step.
//
//   S^T tile = K Q^T :  A = K[key][16 c] from LDS (f16 hi / lo rows), B = Q^T from registers
//   O^T tile = V^T P^T: A = V^T[c][16 keys] from LDS, B = P^T straight from the score registers.
// As in the fp32 kernel a lane's 16 score registers are 16 of the tile's 32 keys, in the order
// key(r) = (r&3) + 8*(r>>2) + 4*half.  An MFMA sums over its k index, so the second product may
// take the keys in ANY order as long as A and B agree: B's k-step s2 is simply registers
// 8*s2..8*s2+7, and V^T is stored in LDS with the keys of each 16-key group permuted to
// pos(k) = ((k>>2)&1)*8 + ((k>>3)&1)*4 + (k&3), which makes the matching 8 keys one 16-byte read.
// Rows are padded to an ODD number of 16-byte slots: a ds_read_b128 is served in 16-lane groups
// whose rows are distinct mod 16, so odd strides are conflict-free (MI355X_MICROARCH.md, LDS).
//
// Power-of-two scales keep the lo parts normal f16 and the hi parts finite: Q, K, V x S, with S the
// power of two that puts the caller's bound of |qkv| just below 2^15 (qkv_scale below; nothing is
// clamped), P in [0,1] x2^12; the scores are un-scaled (x S^-2, exact) before the fp32 softmax and
// the output's S * 2^12 is folded into the final 1/l.  exp is exp2(x*log2 e) (v_exp_f32).
typedef _Float16 ah8 __attribute__((ext_vector_type(8)));
typedef _Float16 ah4 __attribute__((ext_vector_type(4)));

struct HiLo { _Float16 hi, lo; };
__device__ __forceinline__ HiLo split_f16(float x, float scale) {
    const float s = x * scale;
    HiLo r;
    r.hi = (_Float16)s;
    r.lo = (_Float16)(s - (float)r.hi);
    return r;
}

// S = 2^(14 - floor(log2 b)), b = the largest of sample n's bound entries (uniform over the workgroup)
__device__ __forceinline__ float qkv_scale(const float* __restrict__ bound, int count, int stride, int n) {
    const int lane = threadIdx.x & 63;
    float b = 0.0f;
    if (lane < count) b = bound[((size_t)n * count + lane) * stride];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) b = fmaxf(b, __shfl_xor(b, o));
    b = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, b)));
    const int e = (int)((__builtin_bit_cast(unsigned, b) >> 23) & 0xffu) - 127;
    int k = 14 - e;
    if (!(b > 0.0f) || e == 128) k = 0;
    k = k < -40 ? -40 : (k > 40 ? 40 : k);
    return __builtin_bit_cast(float, (unsigned)(127 + k) << 23);
}

// (waves per SIMD stated explicitly: with the bare bound the compiler keeps the MFMA accumulators
// in AGPRs and spends 144 v_accvgpr_read/write per tile moving them around the softmax)
template <int CH>
__global__ __launch_bounds__(256, (CH == 128 ? 2 : 3)) void attention_x3_kernel(const float* __restrict__ qkv, int T, int heads,
                                                           const float* __restrict__ bound, int bcount, int bstride,
                                                           float* __restrict__ out) {
    constexpr int KT = 32;                    // keys per tile
    constexpr int KS = CH / 16;               // k-steps of the first product
    constexpr int CT = CH / 32;               // 32-channel output tiles
    constexpr int KROW = CH * 2 + 16;         // bytes per K row (f16): CH/8 + 1 slots, odd
    constexpr int VROW = KT * 2 + 16;         // bytes per V^T row (32 keys, f16): 5 slots
    constexpr float P_SCALE = 4096.0f;
    static_assert(((KROW / 16) & 1) == 1 && ((VROW / 16) & 1) == 1, "odd slot strides");
    __shared__ __attribute__((aligned(16))) unsigned char Kh[KT * KROW], Kl[KT * KROW];
    __shared__ __attribute__((aligned(16))) unsigned char Vh[CH * VROW], Vl[CH * VROW];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qi = lane & 31, half = lane >> 5;
    const int nh = blockIdx.y;
    const int n = nh / heads, head = nh % heads;
    const int C3 = heads * 3 * CH;
    const float* base = qkv + (size_t)n * T * C3 + (size_t)head * 3 * CH;
    const float QKV_SCALE = qkv_scale(bound, bcount, bstride, n);
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int tq = q0 + qi;
    const bool qvalid = tq < T;

    // Q^T operand: lane (query, half) supplies Q[query][16 s + 8 half .. +8] at k-step s, with
    // the softmax scale s^2 = CH^-1/2 folded in before the split
    ah8 qhi[KS], qlo[KS];
    {
        const float scale = 1.0f / sqrtf((float)CH);
        const float* qrow = base + (size_t)(qvalid ? tq : 0) * C3;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(qrow + 16 * s + 8 * half);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(qrow + 16 * s + 8 * half + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const HiLo a = split_f16(v0[i] * scale, QKV_SCALE), b = split_f16(v1[i] * scale, QKV_SCALE);
                qhi[s][i] = a.hi; qlo[s][i] = a.lo;
                qhi[s][4 + i] = b.hi; qlo[s][4 + i] = b.lo;
            }
        }
    }

    f32x16 oacc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[c][i] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;

    // The tile's K and V rows are fetched ONE TILE AHEAD into registers (item = (K or V, key j,
    // channel quad c4), NI per thread): the global-load latency runs beside the previous tile's
    // MFMAs instead of between two barriers (without it a tile took ~3.9 us for 0.4 us of MFMA).
    constexpr int NI = KT * (CH / 4) * 2 / 256;
    static_assert(KT * (CH / 4) * 2 % 256 == 0, "whole items per thread");
    f32x4 pre[NI];
    auto fetch = [&](const int k0) {
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int idx = tid + it * 256;
            const int which = idx / (KT * (CH / 4));        // 0 = K, 1 = V
            const int r = idx % (KT * (CH / 4));
            const int j = r / (CH / 4), c4 = r % (CH / 4);
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            pre[it] = k0 + j < T
                          ? *reinterpret_cast<const f32x4*>(base + (size_t)(k0 + j) * C3 + (which + 1) * CH + c4 * 4)
                          : zero;                            // keys beyond T are zeros
        }
    };
    fetch(0);

    for (int k0 = 0; k0 < T; k0 += KT) {
        __syncthreads();
        // split the prefetched rows into the LDS tile
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int idx = tid + it * 256;
            const int which = idx / (KT * (CH / 4));
            const int r = idx % (KT * (CH / 4));
            const int j = r / (CH / 4), c4 = r % (CH / 4);
            const f32x4 v = pre[it];
            ah4 hi, lo;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const HiLo a = split_f16(v[i], QKV_SCALE);
                hi[i] = a.hi; lo[i] = a.lo;
            }
            if (which == 0) {
                *reinterpret_cast<ah4*>(Kh + j * KROW + c4 * 8) = hi;
                *reinterpret_cast<ah4*>(Kl + j * KROW + c4 * 8) = lo;
            } else {
                // V^T[c][pos(key)]: the key order the score registers of a lane half are in
                const int k16 = j & 15;
                const int pos = (j & 16) + ((k16 >> 2) & 1) * 8 + ((k16 >> 3) & 1) * 4 + (k16 & 3);
                // Channel c = 4 c4 + i lives in LDS row i * (CH/4) + c4: the lanes of a wave (consecutive
                // c4) then write CONSECUTIVE rows, 20 dwords apart = 16 distinct banks.  (Rows 4 apart,
                // the natural order, are 80 dwords apart = 4 banks: a 16-way conflict on every one of
                // these 2-byte stores.)  The output rows come out in that order; see the epilogue.
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    *reinterpret_cast<_Float16*>(Vh + (i * (CH / 4) + c4) * VROW + pos * 2) = hi[i];
                    *reinterpret_cast<_Float16*>(Vl + (i * (CH / 4) + c4) * VROW + pos * 2) = lo[i];
                }
            }
        }
        __syncthreads();
        if (k0 + KT < T) fetch(k0 + KT);

        // S^T[key][query] = sum_c K[key][c] * Q[query][c]   (x 2^6)
        f32x16 sacc;
#pragma unroll
        for (int i = 0; i < 16; ++i) sacc[i] = 0.0f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const ah8 khi = *reinterpret_cast<const ah8*>(Kh + qi * KROW + (16 * s + 8 * half) * 2);
            const ah8 klo = *reinterpret_cast<const ah8*>(Kl + qi * KROW + (16 * s + 8 * half) * 2);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(klo, qhi[s], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(khi, qlo[s], sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(khi, qhi[s], sacc, 0, 0, 0);
        }

        // Scores stay in their x 2^6 units (max and differences do not care); the un-scaling and
        // exp's log2(e) are one constant in the exp2 argument.  Only the ragged last tile masks.
        const float EXP2_C = 1.44269504088896341f / (QKV_SCALE * QKV_SCALE);
        if (k0 + KT > T) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (k0 + (r & 3) + 8 * (r >> 2) + 4 * half >= T) sacc[r] = -INFINITY;
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sacc[r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m_new = fmaxf(m_run, tmax);           // finite: every tile has >= 1 valid key
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * EXP2_C);   // first tile: exp2(-inf) = 0
        float psum = 0.0f;
        ah8 phi[2], plo[2];                               // P^T operand: k-step s2 = registers 8 s2 .. 8 s2 + 7
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pr = __builtin_amdgcn_exp2f((sacc[r] - m_new) * EXP2_C);
            psum += pr;
            const float ps = pr * P_SCALE;                 // in [0, 4096]: no clamp needed
            const _Float16 ph = (_Float16)ps;
            phi[r >> 3][r & 7] = ph;
            plo[r >> 3][r & 7] = (_Float16)(ps - (float)ph);
        }
        psum += __shfl_xor(psum, 32);
        l_run = l_run * alpha + psum;
        m_run = m_new;

        // O^T[c][query] = alpha * O^T + sum_key V[key][c] * P^T[key][query]   (x 2^15)
        // (the rescale is skipped when no query of the wave raised its running max: alpha == 1
        // exactly, the common case once the first tiles have been seen)
        const bool rescale = __builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            if (rescale) {
#pragma unroll
                for (int i = 0; i < 16; ++i) oacc[c][i] *= alpha;
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const ah8 vhi = *reinterpret_cast<const ah8*>(Vh + (c * 32 + qi) * VROW + (16 * s2 + 8 * half) * 2);
                const ah8 vlo = *reinterpret_cast<const ah8*>(Vl + (c * 32 + qi) * VROW + (16 * s2 + 8 * half) * 2);
                oacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vlo, phi[s2], oacc[c], 0, 0, 0);
                oacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vhi, plo[s2], oacc[c], 0, 0, 0);
                oacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vhi, phi[s2], oacc[c], 0, 0, 0);
            }
        }
    }

    if (!qvalid) return;
    const float inv = 1.0f / (l_run * QKV_SCALE * P_SCALE);
    float* orow = out + ((size_t)n * T + tq) * (heads * CH) + (size_t)head * CH;
    // accumulator row m of tile c is LDS row R = 32 c + m, i.e. channel (R % (CH/4)) * 4 + R / (CH/4)
#pragma unroll
    for (int c = 0; c < CT; ++c) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int R = c * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            orow[(R % (CH / 4)) * 4 + R / (CH / 4)] = oacc[c][r] * inv;
        }
    }
}

hipError_t ddpm3d_launch_attention(const float* qkv, int N, int T, int heads, int ch, int precision,
                                   const float* bound, int bcount, int bstride, float* out, hipStream_t st) {
    dim3 grid((T + 127) / 128, N * heads);
    if (precision != DDPM3D_PREC_F32) {
        if (ch == 64)
            hipLaunchKernelGGL(attention_x3_kernel<64>, grid, dim3(256), 0, st, qkv, T, heads, bound, bcount, bstride, out);
        else if (ch == 32)
            hipLaunchKernelGGL(attention_x3_kernel<32>, grid, dim3(256), 0, st, qkv, T, heads, bound, bcount, bstride, out);
        else if (ch == 128)
            hipLaunchKernelGGL(attention_x3_kernel<128>, grid, dim3(256), 0, st, qkv, T, heads, bound, bcount, bstride, out);
        else
            return hipErrorInvalidValue;
        return hipGetLastError();
    }
    if (ch == 64)
        hipLaunchKernelGGL(attention_kernel<64>, grid, dim3(256), 0, st, qkv, T, heads, out);
    else if (ch == 32)
        hipLaunchKernelGGL(attention_kernel<32>, grid, dim3(256), 0, st, qkv, T, heads, out);
    else if (ch == 128)
        hipLaunchKernelGGL(attention_kernel<128>, grid, dim3(256), 0, st, qkv, T, heads, out);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}
