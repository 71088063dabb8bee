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
#include "ops.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CH>
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ qkv, int T, int heads,
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

hipError_t ddpm3d_launch_attention(const float* qkv, int N, int T, int heads, int ch, float* out,
                                   hipStream_t st) {
    dim3 grid((T + 127) / 128, N * heads);
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
