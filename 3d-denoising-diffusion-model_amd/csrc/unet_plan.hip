// The whole-network level of the C ABI (include/ddpm3d.h, "whole network"): a UNet forward
// (unet.py:1015-1044, :1687-1694; ResBlock :236-256; AttentionBlock :296-305; Downsample / Upsample
// :102-105, :129-136) compiled ONCE per (model, N, D, H, W) into a flat list of this library's own per-op
// calls -- every buffer carved out of ONE caller-provided device arena -- and replayed by
// ddpm3d_unet_forward.  The same plan the Python host builds (guided_diffusion/engine.py: _Plan), for
// hosts that are not Python: same calls, same arguments, same order, hence bit-identical results
// (tests/test_gpu_model.py::test_native_plan_equals_python_plan).  Host code only: no kernel lives here.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <map>
#include <new>
#include <vector>
#include "ddpm3d.h"

namespace {

struct Act {                 // an NDHWC activation of the plan, with the GroupNorm partial sums its producer writes
    size_t off = 0;          // byte offset in the arena
    int C = 0, D = 0, H = 0, W = 0;
    int esize = 4;           // bytes per element (2 = the 16-bit residual stream of the f16 / bf16 modes)
    size_t stats_off = 0;
    int rows = 0;
    size_t bytes(int N) const { return (size_t)N * D * H * W * C * esize; }
    long long voxels() const { return (long long)D * H * W; }
};

enum StepKind { S_CONV, S_FINALIZE, S_ABSMAX, S_PAD, S_POOL, S_ATTN };

struct Step {
    int kind;
    ddpm3d_conv_desc conv;                       // S_CONV
    // S_FINALIZE
    size_t st0, st1; int C0, rows0, C1, rows1; double count; const float *gamma, *beta; int film, film_off;
    size_t A, B, bound; bool has_ab;
    // S_ABSMAX / S_PAD (input edge; x pointer patched per call)
    size_t per_sample; int two; size_t dst; int C, Cpad, vox;
    // S_POOL
    size_t src, pA, pB; int act, D, H, W, io;
    // S_ATTN
    size_t qkv, qb; int T, heads, ch, prec; bool has_qb;
};

struct Planner {
    const ddpm3d_unet_desc& m;
    int N, D, H, W;
    char* base;                                   // arena (NULL while sizing)
    size_t top = 0;
    std::map<std::pair<size_t, int>, std::vector<size_t>> pool;   // (bytes, esize) -> free activation buffers
    std::vector<Step> steps;
    std::vector<int> film_steps, bias_steps;      // finalize steps / conv steps whose film pointer is per call
    std::vector<int> bias_offs;
    std::vector<int> ws_steps;                    // conv steps that use the shared split-K workspace
    size_t ws_bytes = 0, ws_off = 0;
    int first_step = -1, last_step = -1, absmax_step = -1, pad_step = -1;
    size_t in_absmax = 0;
    int half_esize;                               // 2 in the f16 / bf16 modes
    bool scaled, f16;
    char err[256] = "";
    int code = DDPM3D_EINVAL;                     // what a failed build returns (DDPM3D_E2BIG: split the batch)

    Planner(const ddpm3d_unet_desc& d, int n, int dd, int hh, int ww, char* b)
        : m(d), N(n), D(dd), H(hh), W(ww), base(b) {
        half_esize = (m.arithmetic == DDPM3D_PREC_F16 || m.arithmetic == DDPM3D_PREC_BF16) ? 2 : 4;
        f16 = m.arithmetic == DDPM3D_PREC_F16;
        scaled = m.arithmetic == DDPM3D_PREC_F16X3 || m.arithmetic == DDPM3D_PREC_F16;
    }
    bool fail(const char* fmt, ...) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(err, sizeof(err), fmt, ap);
        va_end(ap);
        return false;
    }
    size_t alloc(size_t bytes) {                  // 256-byte aligned bump allocation
        const size_t o = (top + 255) & ~(size_t)255;
        top = o + bytes;
        return o;
    }
    void* at(size_t off) const { return base + off; }

    Act new_act(int C, int d, int h, int w, bool fp32 = false) {
        Act a;
        a.C = C; a.D = d; a.H = h; a.W = w;
        a.esize = fp32 ? 4 : half_esize;
        const size_t b = a.bytes(N);
        auto& fr = pool[{b, a.esize}];
        if (!fr.empty()) { a.off = fr.back(); fr.pop_back(); }
        else a.off = alloc(b);
        return a;
    }
    void release(const Act& a) { pool[{a.bytes(N), a.esize}].push_back(a.off); }

    // gn_finalize over the virtual concat of srcs; gamma NULL = bounds only (skipped when nothing reads them)
    bool finalize(const Act* s0, const Act* s1, const float* gamma, const float* beta, int film_off, bool force_bound,
                  size_t* A, size_t* B, size_t* bound, bool* has) {
        *has = false;
        if (!gamma && !(scaled || force_bound)) return true;
        if (s1 && s1->voxels() != s0->voxels()) return fail("concat of tensors with different spatial size");
        const int Cn = s0->C + (s1 ? s1->C : 0);
        Step st;
        memset(&st, 0, sizeof(st));
        st.kind = S_FINALIZE;
        st.st0 = s0->stats_off; st.C0 = s0->C; st.rows0 = s0->rows;
        st.st1 = s1 ? s1->stats_off : 0; st.C1 = s1 ? s1->C : 0; st.rows1 = s1 ? s1->rows : 0;
        st.count = (double)s0->voxels();
        st.gamma = gamma; st.beta = beta;
        st.has_ab = gamma != nullptr;
        if (gamma) { st.A = alloc((size_t)N * Cn * 4); st.B = alloc((size_t)N * Cn * 4); }
        st.bound = alloc((size_t)N * 32 * 2 * 4);
        st.film = film_off >= 0;
        st.film_off = film_off >= 0 ? film_off : 0;
        if (st.film) film_steps.push_back((int)steps.size());
        steps.push_back(st);
        *A = st.A; *B = st.B; *bound = st.bound; *has = true;
        return true;
    }

    struct ConvArgs {
        const Act* src0 = nullptr; const Act* src1 = nullptr;
        Act* out = nullptr;
        size_t A = 0, B = 0; bool aff = false;
        int act = DDPM3D_ACT_NONE, in_mode = DDPM3D_IN_SAME;
        const Act* res = nullptr; int res_mode = DDPM3D_RES_NONE;
        bool planar = false, want_stats = true, bias_per_n = false;
        bool ncdhw_out = false;                    // the network's last conv: out pointer patched per call
        size_t bound = 0; int bound_first = 0, bound_count = 0, bound_stride = 0; bool has_bound = false;
        int film_off = 0;
    };
    bool conv_step(const ddpm3d_conv_weights& pc, ConvArgs a, int* step_index = nullptr) {
        ddpm3d_conv_desc d;
        memset(&d, 0, sizeof(d));
        const bool wz = pc.w_packed_wz && !a.planar && (a.in_mode == DDPM3D_IN_SAME || a.in_mode == DDPM3D_IN_UP);
        d.precision = wz ? pc.precision_wz : pc.precision;
        d.w_packed = wz ? pc.w_packed_wz : pc.w_packed;
        d.bias = pc.bias;
        d.Cout = pc.Cout; d.ksize = pc.ksize;
        const Act* geo = a.out ? a.out : a.src0;
        d.N = N; d.D = geo->D; d.H = geo->H; d.W = geo->W;
        d.out_layout = a.ncdhw_out ? DDPM3D_OUT_NCDHW : DDPM3D_OUT_NDHWC;
        if (a.planar) {
            d.in_mode = DDPM3D_IN_PLANAR2; d.Cin = 2; d.C0 = 1; d.C1 = 1;
        } else {
            d.in_mode = a.in_mode;
            d.src0 = (const float*)at(a.src0->off); d.C0 = a.src0->C;
            if (a.src1) { d.src1 = (const float*)at(a.src1->off); d.C1 = a.src1->C; }
            d.Cin = d.C0 + d.C1;
            if (d.Cin != pc.Cin) return fail("conv %dx%d fed %d channels", pc.Cout, pc.Cin, d.Cin);
            if (a.src0->esize == 2) d.io_dtype |= DDPM3D_IO_SRC0_BF16;
            if (a.src1 && a.src1->esize == 2) d.io_dtype |= DDPM3D_IO_SRC1_BF16;
        }
        if (a.out) {
            d.out = (float*)at(a.out->off);
            if (a.out->esize == 2) d.io_dtype |= DDPM3D_IO_OUT_BF16;
        }
        if (a.aff) { d.aff_a = (const float*)at(a.A); d.aff_b = (const float*)at(a.B); }
        d.act = a.act;
        if (a.res) {
            if (a.res->C != pc.Cout) return fail("residual has %d channels, the conv writes %d", a.res->C, pc.Cout);
            d.res = (const float*)at(a.res->off);
            if (a.res->esize == 2) d.io_dtype |= DDPM3D_IO_RES_BF16;
        }
        d.res_mode = a.res_mode;
        if (d.io_dtype && f16) d.io_dtype |= DDPM3D_IO_HALF_IS_F16;
        if (scaled) {
            if (!a.has_bound) return fail("conv step without an input bound");
            d.in_bound = (const float*)at(a.bound) + a.bound_first;
            d.in_bound_count = a.bound_count; d.in_bound_stride = a.bound_stride;
        }
        // how the library itself will run this descriptor: statistics rows, split-K scratch (needs real-looking
        // pointers only where the validation looks at them: none while sizing)
        int rows = 0, split = 1;
        size_t need = 0;
        const int rc = ddpm3d_conv_plan(&d, &rows, &need, &split);
        if (rc != DDPM3D_OK) { code = rc; return fail("%s", ddpm3d_last_error()); }
        if (a.out && a.want_stats) {
            a.out->rows = rows;
            a.out->stats_off = alloc((size_t)N * rows * pc.Cout * 2 * sizeof(double));
            d.stats = (double*)at(a.out->stats_off);
            d.stats_rows = rows;
        }
        if (need) { ws_steps.push_back((int)steps.size()); if (need > ws_bytes) ws_bytes = need; }
        Step st;
        memset(&st, 0, sizeof(st));
        st.kind = S_CONV;
        st.conv = d;
        if (a.bias_per_n) { bias_steps.push_back((int)steps.size()); bias_offs.push_back(a.film_off); }
        if (step_index) *step_index = (int)steps.size();
        steps.push_back(st);
        return true;
    }

    bool resblock(const ddpm3d_layer& e, const Act* s0, const Act* s1, Act* y_out) {
        int d = s0->D, h = s0->H, w = s0->W, im = DDPM3D_IN_SAME, rm = DDPM3D_RES_SAME;
        if (e.updown == DDPM3D_UPDOWN_DOWN) {
            if ((s0->H | s0->W) & 1) return fail("Downsample needs even H, W (got %dx%d)", s0->H, s0->W);
            h /= 2; w /= 2; im = DDPM3D_IN_POOL; rm = DDPM3D_RES_POOL;
        } else if (e.updown == DDPM3D_UPDOWN_UP) {
            h *= 2; w *= 2; im = DDPM3D_IN_UP; rm = DDPM3D_RES_UP;
        }
        size_t A1, B1, bnd1; bool has1;
        if (!finalize(s0, s1, e.norm1_gamma, e.norm1_beta, -1, false, &A1, &B1, &bnd1, &has1)) return false;
        Act h1 = new_act(e.conv1.Cout, d, h, w);
        ConvArgs c1;
        c1.src0 = s0; c1.src1 = s1; c1.out = &h1; c1.A = A1; c1.B = B1; c1.aff = true; c1.act = DDPM3D_ACT_SILU;
        c1.in_mode = im; c1.bound = bnd1; c1.bound_first = 0; c1.bound_count = 32; c1.bound_stride = 2; c1.has_bound = has1;
        Act pooled;
        bool use_pooled = false;
        if (e.updown == DDPM3D_UPDOWN_DOWN && e.conv1.w_packed_wz && !s1) {
            // h_upd(in_rest(x)) as a pass of its own: conv1 then reads a plain tensor and runs its Winograd-D form
            pooled = new_act(s0->C, d, h, w, true);
            Step st;
            memset(&st, 0, sizeof(st));
            st.kind = S_POOL;
            st.src = s0->off; st.pA = A1; st.pB = B1; st.act = DDPM3D_ACT_SILU;
            st.D = d; st.H = h; st.W = w; st.C = s0->C; st.dst = pooled.off;
            st.io = (s0->esize == 2 ? DDPM3D_IO_SRC0_BF16 : 0) | ((s0->esize == 2 && f16) ? DDPM3D_IO_HALF_IS_F16 : 0);
            steps.push_back(st);
            c1.src0 = &pooled; c1.aff = false; c1.act = DDPM3D_ACT_NONE; c1.in_mode = DDPM3D_IN_SAME;
            use_pooled = true;
        }
        size_t A2, B2, bnd2; bool has2;
        if (m.film) {
            if (!conv_step(e.conv1, c1)) return false;
            if (!finalize(&h1, nullptr, e.norm2_gamma, e.norm2_beta, e.film_off, false, &A2, &B2, &bnd2, &has2)) return false;
        } else {
            c1.bias_per_n = true; c1.film_off = e.film_off;     // additive embedding: conv1's bias is the film row slice
            if (!conv_step(e.conv1, c1)) return false;
            if (!finalize(&h1, nullptr, e.norm2_gamma, e.norm2_beta, -1, false, &A2, &B2, &bnd2, &has2)) return false;
        }
        if (use_pooled) release(pooled);
        Act y = new_act(e.conv2.Cout, d, h, w);
        ConvArgs c2;
        c2.src0 = &h1; c2.out = &y; c2.A = A2; c2.B = B2; c2.aff = true; c2.act = DDPM3D_ACT_SILU;
        c2.bound = bnd2; c2.bound_count = 32; c2.bound_stride = 2; c2.has_bound = has2;
        if (e.skip.w_packed) {
            if (e.updown != DDPM3D_UPDOWN_NONE) return fail("up/down ResBlock with a channel change is not in the reference");
            ConvArgs cs;        // y = skip(x) on the RAW block input (its range: entry 1 of the same finalize)
            cs.src0 = s0; cs.src1 = s1; cs.out = &y; cs.want_stats = false;
            cs.bound = bnd1; cs.bound_first = 1; cs.bound_count = 32; cs.bound_stride = 2; cs.has_bound = has1;
            if (!conv_step(e.skip, cs)) return false;
            c2.res = &y; c2.res_mode = DDPM3D_RES_SAME;
        } else {
            if (s1) return fail("ResBlock with an Identity skip over a concatenated input (%d + %d -> %d channels)",
                                s0->C, s1->C, e.conv2.Cout);
            c2.res = s0; c2.res_mode = rm;
        }
        if (!conv_step(e.conv2, c2)) return false;
        release(h1);
        *y_out = y;
        return true;
    }

    bool attention(const ddpm3d_layer& e, const Act* x, Act* y_out) {
        const int Cn = x->C;
        if (e.heads <= 0 || Cn % e.heads) return fail("attention: %d channels not divisible by %d heads", Cn, e.heads);
        const int ch = Cn / e.heads;
        if (ch != 32 && ch != 64 && ch != 128) return fail("attention with %d channels per head (32, 64 or 128 are built)", ch);
        size_t A, B, bnd; bool has;
        if (!finalize(x, nullptr, e.norm1_gamma, e.norm1_beta, -1, false, &A, &B, &bnd, &has)) return false;
        Act qkv = new_act(3 * Cn, x->D, x->H, x->W, true);
        ConvArgs cq;
        cq.src0 = x; cq.out = &qkv; cq.A = A; cq.B = B; cq.aff = true; cq.act = DDPM3D_ACT_NONE;
        cq.bound = bnd; cq.bound_count = 32; cq.bound_stride = 2; cq.has_bound = has;
        if (!conv_step(e.conv1, cq)) return false;
        size_t qA, qB, qb; bool hasq;
        if (!finalize(&qkv, nullptr, nullptr, nullptr, -1, m.arithmetic != DDPM3D_PREC_F32, &qA, &qB, &qb, &hasq)) return false;
        Act a = new_act(Cn, x->D, x->H, x->W, true);
        Step st;
        memset(&st, 0, sizeof(st));
        st.kind = S_ATTN;
        st.qkv = qkv.off; st.T = (int)x->voxels(); st.heads = e.heads; st.ch = ch;
        st.prec = m.arithmetic == DDPM3D_PREC_F32 ? DDPM3D_PREC_F32 : DDPM3D_PREC_F16X3;
        st.qb = qb; st.has_qb = hasq; st.dst = a.off;
        steps.push_back(st);
        release(qkv);
        Act y = new_act(Cn, x->D, x->H, x->W);
        ConvArgs cp;
        cp.src0 = &a; cp.out = &y; cp.res = x; cp.res_mode = DDPM3D_RES_SAME;
        cp.bound = qb; cp.bound_first = 1; cp.bound_count = 32; cp.bound_stride = 2; cp.has_bound = hasq;
        if (!conv_step(e.conv2, cp)) return false;
        release(a);
        *y_out = y;
        return true;
    }

    bool layer(const ddpm3d_layer& e, const Act* s0, const Act* s1, Act* out) {
        // a description with a hole in it is refused here, not discovered by a kernel
        const bool n1 = e.norm1_gamma && e.norm1_beta, n2 = e.norm2_gamma && e.norm2_beta;
        const bool c1 = e.conv1.w_packed && e.conv1.bias, c2 = e.conv2.w_packed && e.conv2.bias;
        if (e.kind == DDPM3D_LAYER_RES && !(n1 && n2 && c1 && c2 && (!e.skip.w_packed || e.skip.bias)))
            return fail("ResBlock description without its norms / convs");
        if (e.kind == DDPM3D_LAYER_ATTN && !(n1 && c1 && c2)) return fail("AttentionBlock description without norm / qkv / proj_out");
        if ((e.kind == DDPM3D_LAYER_UPCONV || e.kind == DDPM3D_LAYER_DOWNCONV) && !c1) return fail("resampling conv without weights");
        if (e.kind == DDPM3D_LAYER_RES) return resblock(e, s0, s1, out);
        if (e.kind == DDPM3D_LAYER_ATTN) return attention(e, s0, out);
        if (e.kind == DDPM3D_LAYER_UPCONV || e.kind == DDPM3D_LAYER_DOWNCONV) {
            const bool up = e.kind == DDPM3D_LAYER_UPCONV;
            if (!up && ((s0->H | s0->W) & 1)) return fail("Downsample needs even H, W (got %dx%d)", s0->H, s0->W);
            Act y = new_act(e.conv1.Cout, s0->D, up ? s0->H * 2 : s0->H / 2, up ? s0->W * 2 : s0->W / 2);
            size_t A, B, bnd; bool has;
            if (!finalize(s0, nullptr, nullptr, nullptr, -1, false, &A, &B, &bnd, &has)) return false;
            ConvArgs c;
            c.src0 = s0; c.out = &y; c.in_mode = up ? DDPM3D_IN_UP : DDPM3D_IN_STRIDE2;
            c.bound = bnd; c.bound_first = 1; c.bound_count = 32; c.bound_stride = 2; c.has_bound = has;
            if (!conv_step(e.conv1, c)) return false;
            *out = y;
            return true;
        }
        return fail("unknown layer kind %d", e.kind);
    }

    bool build() {
        Act h = new_act(m.first.Cout, D, H, W);
        Act xin;
        if (m.planar) {
            in_absmax = alloc((size_t)N * 2 * 4);
            Step st;
            memset(&st, 0, sizeof(st));
            st.kind = S_ABSMAX; st.per_sample = (size_t)D * H * W; st.two = 1; st.dst = in_absmax;
            if (scaled) { absmax_step = (int)steps.size(); steps.push_back(st); }
            ConvArgs c;
            c.out = &h; c.planar = true; c.bound = in_absmax; c.bound_count = 2; c.bound_stride = 1; c.has_bound = true;
            if (!conv_step(m.first, c, &first_step)) return false;
        } else {
            xin = new_act(m.cin_pad, D, H, W, true);
            in_absmax = alloc((size_t)N * 4);
            Step st;
            memset(&st, 0, sizeof(st));
            st.kind = S_ABSMAX; st.per_sample = (size_t)m.in_channels * D * H * W; st.two = 0; st.dst = in_absmax;
            if (scaled) { absmax_step = (int)steps.size(); steps.push_back(st); }
            Step sp;
            memset(&sp, 0, sizeof(sp));
            sp.kind = S_PAD; sp.C = m.in_channels; sp.Cpad = m.cin_pad; sp.vox = D * H * W; sp.dst = xin.off;
            pad_step = (int)steps.size();
            steps.push_back(sp);
            ConvArgs c;
            c.src0 = &xin; c.out = &h; c.bound = in_absmax; c.bound_count = 1; c.bound_stride = 1; c.has_bound = true;
            if (!conv_step(m.first, c, &first_step)) return false;
        }
        std::vector<Act> hs;
        hs.push_back(h);
        int li = 0;
        for (int b = 0; b < m.n_input_blocks; ++b) {
            for (int i = 0; i < m.input_block_layers[b]; ++i) {
                Act prev = h;
                if (!layer(m.layers[li++], &prev, nullptr, &h)) return false;
                if (i > 0) release(prev);          // a block's intermediate; its input stays on the skip stack
            }
            hs.push_back(h);
        }
        for (int i = 0; i < m.n_middle_layers; ++i) {
            Act prev = h;
            if (!layer(m.layers[li++], &prev, nullptr, &h)) return false;
            if (prev.off != hs.back().off) release(prev);
        }
        for (int b = 0; b < m.n_output_blocks; ++b) {
            Act skip = hs.back();
            hs.pop_back();
            Act s0 = h, s1 = skip;
            bool two = true;
            for (int i = 0; i < m.output_block_layers[b]; ++i) {
                if (!layer(m.layers[li++], &s0, two ? &s1 : nullptr, &h)) return false;
                release(s0);
                if (two) release(s1);
                s0 = h;
                two = false;
            }
        }
        if (li != m.n_layers) return fail("the block sizes cover %d of %d layers", li, m.n_layers);
        size_t A, B, bnd; bool has;
        if (!finalize(&h, nullptr, m.out_gamma, m.out_beta, -1, false, &A, &B, &bnd, &has)) return false;
        ConvArgs c;
        c.src0 = &h; c.A = A; c.B = B; c.aff = true; c.act = DDPM3D_ACT_SILU; c.ncdhw_out = true; c.want_stats = false;
        c.bound = bnd; c.bound_count = 32; c.bound_stride = 2; c.has_bound = has;
        if (!conv_step(m.out, c, &last_step)) return false;
        release(h);
        if (ws_bytes) ws_off = alloc(ws_bytes);
        for (int i : ws_steps) { steps[i].conv.workspace = at(ws_off); steps[i].conv.workspace_bytes = ws_bytes; }
        return true;
    }
};

thread_local char g_unet_err[256] = "";

}  // namespace

struct ddpm3d_unet_plan {
    std::vector<Step> steps;
    std::vector<int> film_steps, bias_steps, bias_offs;
    int first_step, last_step, absmax_step, pad_step;
    int N, planar, out_channels;
    char* base;
};

extern "C" {

const char* ddpm3d_unet_last_error(void) { return g_unet_err; }

static bool desc_ok(const ddpm3d_unet_desc* m, int N, int D, int H, int W) {
    return m && N > 0 && D > 0 && H > 0 && W > 0 && m->layers && m->n_layers > 0 && m->first.w_packed && m->out.w_packed &&
           m->out_gamma && m->out_beta && (m->n_input_blocks == 0 || m->input_block_layers) &&
           (m->n_output_blocks == 0 || m->output_block_layers) &&
           (m->arithmetic == DDPM3D_PREC_F32 || m->arithmetic == DDPM3D_PREC_F16X3 || m->arithmetic == DDPM3D_PREC_F16 ||
            m->arithmetic == DDPM3D_PREC_BF16);
}

size_t ddpm3d_unet_plan_bytes(const ddpm3d_unet_desc* m, int N, int D, int H, int W) {
    if (!desc_ok(m, N, D, H, W)) { snprintf(g_unet_err, sizeof(g_unet_err), "unet_plan_bytes: bad description"); return 0; }
    // sizing pass: the same planning against a base that is never dereferenced (non-null, so that the per-op
    // validation sees what it will see at run time)
    Planner p(*m, N, D, H, W, reinterpret_cast<char*>(4096));
    if (!p.build()) { snprintf(g_unet_err, sizeof(g_unet_err), "%s", p.err); return 0; }
    return (p.top + 255) & ~(size_t)255;
}

int ddpm3d_unet_plan_create(const ddpm3d_unet_desc* m, int N, int D, int H, int W, void* arena, size_t arena_bytes,
                            ddpm3d_unet_plan** plan) {
    if (!desc_ok(m, N, D, H, W) || !arena || !plan || (reinterpret_cast<uintptr_t>(arena) & 255)) {
        snprintf(g_unet_err, sizeof(g_unet_err), "unet_plan_create: bad description, or an arena that is not 256-byte aligned");
        return DDPM3D_EINVAL;
    }
    Planner p(*m, N, D, H, W, static_cast<char*>(arena));
    if (!p.build()) { snprintf(g_unet_err, sizeof(g_unet_err), "%s", p.err); return p.code; }
    if (p.top > arena_bytes) {
        snprintf(g_unet_err, sizeof(g_unet_err), "unet_plan_create: the plan needs %zu bytes, the arena has %zu", p.top, arena_bytes);
        return DDPM3D_EINVAL;
    }
    ddpm3d_unet_plan* pl = new (std::nothrow) ddpm3d_unet_plan();
    if (!pl) return DDPM3D_EINVAL;
    pl->steps.swap(p.steps);
    pl->film_steps.swap(p.film_steps);
    pl->bias_steps.swap(p.bias_steps);
    pl->bias_offs.swap(p.bias_offs);
    pl->first_step = p.first_step; pl->last_step = p.last_step; pl->absmax_step = p.absmax_step; pl->pad_step = p.pad_step;
    pl->N = N; pl->planar = m->planar; pl->out_channels = m->out.Cout;
    pl->base = static_cast<char*>(arena);
    *plan = pl;
    return DDPM3D_OK;
}

void ddpm3d_unet_plan_destroy(ddpm3d_unet_plan* plan) { delete plan; }

int ddpm3d_unet_forward(ddpm3d_unet_plan* pl, const float* x, const float* low_res, const float* film_rows,
                        int film_stride, float* out, void* stream) {
    if (!pl || !x || !film_rows || !out || (pl->planar && !low_res)) {
        snprintf(g_unet_err, sizeof(g_unet_err), "unet_forward: null argument");
        return DDPM3D_EINVAL;
    }
    char* b = pl->base;
    if (pl->planar) {
        pl->steps[pl->first_step].conv.src0 = x;
        pl->steps[pl->first_step].conv.src1 = low_res;
    }
    for (size_t i = 0; i < pl->bias_steps.size(); ++i) {
        ddpm3d_conv_desc& d = pl->steps[pl->bias_steps[i]].conv;
        d.bias = film_rows + pl->bias_offs[i];
        d.bias_stride_n = film_stride;
    }
    pl->steps[pl->last_step].conv.out = out;
    for (const Step& s : pl->steps) {
        int rc = DDPM3D_OK;
        switch (s.kind) {
            case S_CONV: rc = ddpm3d_conv3d(&s.conv, stream); break;
            case S_FINALIZE:
                rc = ddpm3d_gn_finalize((const double*)(b + s.st0), s.C0, s.rows0, s.C1 ? (const double*)(b + s.st1) : nullptr,
                                        s.C1, s.rows1, pl->N, 32, s.count, 1e-5f, s.gamma, s.beta,
                                        s.film ? film_rows : nullptr, s.film ? film_stride : 0, s.film_off,
                                        s.has_ab ? (float*)(b + s.A) : nullptr, s.has_ab ? (float*)(b + s.B) : nullptr,
                                        (float*)(b + s.bound), stream);
                break;
            case S_ABSMAX:
                rc = ddpm3d_absmax(x, s.two ? low_res : nullptr, pl->N, s.per_sample, (float*)(b + s.dst), stream);
                break;
            case S_PAD: rc = ddpm3d_ncdhw_to_ndhwc_pad(x, pl->N, s.C, s.vox, s.Cpad, (float*)(b + s.dst), stream); break;
            case S_POOL:
                rc = ddpm3d_pool_act(b + s.src, (const float*)(b + s.pA), (const float*)(b + s.pB), s.act, 1, pl->N, s.D, s.H,
                                     s.W, s.C, b + s.dst, s.io, stream);
                break;
            case S_ATTN:
                rc = ddpm3d_attention_p((const float*)(b + s.qkv), pl->N, s.T, s.heads, s.ch, s.prec,
                                        s.has_qb ? (const float*)(b + s.qb) + 1 : nullptr, 32, 2, (float*)(b + s.dst), stream);
                break;
        }
        if (rc != DDPM3D_OK) {
            snprintf(g_unet_err, sizeof(g_unet_err), "unet_forward: %s", ddpm3d_last_error());
            return rc;
        }
    }
    return DDPM3D_OK;
}

}  // extern "C"
