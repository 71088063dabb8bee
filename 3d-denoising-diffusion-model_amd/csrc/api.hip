// C ABI (include/ddpm3d.h): argument validation + launcher calls.  Nothing
// here allocates or synchronises; every entry point only enqueues on `stream`.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include "conv3d_params.h"
#include "ddpm3d.h"
#include "ops.h"

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
static int launched(hipError_t e, const char* what) {
    if (e == hipSuccess) return DDPM3D_OK;
    return fail(DDPM3D_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
}
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" {

int ddpm3d_abi_version(void) { return DDPM3D_ABI_VERSION; }
const char* ddpm3d_last_error(void) { return g_err; }

static bool prec_ok(int p) { return p >= DDPM3D_PREC_F32 && p <= DDPM3D_PREC_BF16_WZ; }
static bool prec_wz(int p) { return p == DDPM3D_PREC_F16X3_WZ || p == DDPM3D_PREC_F16_WZ || p == DDPM3D_PREC_BF16_WZ; }
// modes whose activation scale comes from in_bound (the bf16 modes need none: fp32's exponent range)
static bool prec_scaled(int p) { return p != DDPM3D_PREC_F32 && p != DDPM3D_PREC_BF16 && p != DDPM3D_PREC_BF16_WZ; }

// the Winograd-D form exists for 3x3x3 layers whose couts fill whole 128-wide workgroups
static bool wz_layer_ok(int Cout, int Cin, int ksize) {
    return ksize == 3 && Cout % 128 == 0 && Cin % DDPM3D_CONV_CK == 0;
}

size_t ddpm3d_packed_weight_bytes(int Cout, int Cin, int ksize, int precision) {
    if (Cout <= 0 || Cin <= 0 || (ksize != 1 && ksize != 3) || !prec_ok(precision)) return 0;
    if (prec_wz(precision) && !wz_layer_ok(Cout, Cin, ksize)) return 0;
    return ddpm3d_packed_bytes(Cout, Cin, ksize, precision);
}

int ddpm3d_pack_conv_weight(const float* w, int Cout, int Cin, int ksize, int precision, void* out,
                            void* stream) {
    if (!w || !out || Cout <= 0 || Cin <= 0 || (ksize != 1 && ksize != 3) || !prec_ok(precision))
        return fail(DDPM3D_EINVAL, "pack_conv_weight: bad arguments (Cout=%d Cin=%d k=%d precision=%d)", Cout,
                    Cin, ksize, precision);
    if (!aligned16(out)) return fail(DDPM3D_EINVAL, "pack_conv_weight: w_packed must be 16-byte aligned");
    if (prec_wz(precision) && !wz_layer_ok(Cout, Cin, ksize))
        return fail(DDPM3D_ENOSUP, "pack_conv_weight: the Winograd-D form needs ksize 3, Cout %% 128 == 0, "
                                   "Cin %% 16 == 0 (got Cout=%d Cin=%d k=%d)", Cout, Cin, ksize);
    return launched(ddpm3d_launch_pack(w, Cout, Cin, ksize, precision, out, (hipStream_t)stream),
                    "pack_conv_weight");
}

int ddpm3d_conv_stats_rows(int N, int D, int H, int W, int Cin, int Cout, int ksize, int precision) {
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || !prec_ok(precision)) return 0;
    return ddpm3d_conv_cfg(N, D, H, W, Cin, Cout, ksize, precision).stats_rows;
}

size_t ddpm3d_conv_workspace_bytes(int N, int D, int H, int W, int Cin, int Cout, int ksize, int precision) {
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || !prec_ok(precision)) return 0;
    return ddpm3d_conv_cfg(N, D, H, W, Cin, Cout, ksize, precision).workspace_bytes;
}

// Validation and routing shared by ddpm3d_conv3d and ddpm3d_conv_kernel_family: fills the launch record and
// says which kernel takes the call.
enum { ROUTE_GENERAL = 0, ROUTE_SKINNY = 1, ROUTE_PW = 2 };
static int conv_prepare(const ddpm3d_conv_desc* d, ConvK& k, ConvCfg& c, int& route, bool for_launch) {
    if (!d) return fail(DDPM3D_EINVAL, "conv3d: null descriptor");
    if (d->N <= 0 || d->D <= 0 || d->H <= 0 || d->W <= 0 || d->Cout <= 0 || d->Cin <= 0)
        return fail(DDPM3D_EINVAL, "conv3d: non-positive shape N=%d D=%d H=%d W=%d Cin=%d Cout=%d", d->N, d->D,
                    d->H, d->W, d->Cin, d->Cout);
    if (d->ksize != 1 && d->ksize != 3) return fail(DDPM3D_EINVAL, "conv3d: ksize %d (1 or 3)", d->ksize);
    if (!prec_ok(d->precision)) return fail(DDPM3D_ENOSUP, "conv3d: precision mode %d not implemented", d->precision);
    if (d->C0 + d->C1 != d->Cin) return fail(DDPM3D_EINVAL, "conv3d: C0+C1 != Cin");
    // (a family query may come before the caller has attached its per-call buffers)
    if (for_launch && (!d->src0 || !d->w_packed || !d->bias || !d->out)) return fail(DDPM3D_EINVAL, "conv3d: null buffer");
    if (d->in_mode == DDPM3D_IN_PLANAR2) {
        if (d->Cin != 2 || d->C0 != 1 || d->C1 != 1 || (for_launch && !d->src1) || d->aff_a)
            return fail(DDPM3D_EINVAL, "conv3d: planar2 input needs C0=C1=1, two planes, no affine");
    } else {
        if (d->in_mode < 0 || d->in_mode > DDPM3D_IN_STRIDE2) return fail(DDPM3D_EINVAL, "conv3d: in_mode %d", d->in_mode);
        if (d->in_mode == DDPM3D_IN_STRIDE2 && d->ksize != 3)
            return fail(DDPM3D_EINVAL, "conv3d: the strided input mode is the 3x3x3 Downsample conv's");
        if (d->Cin % DDPM3D_CONV_CK) return fail(DDPM3D_EINVAL, "conv3d: Cin=%d not a multiple of %d", d->Cin, DDPM3D_CONV_CK);
        if (d->C1 > 0 && (d->C0 % DDPM3D_CONV_CK || (for_launch && !d->src1)))
            return fail(DDPM3D_EINVAL, "conv3d: concat needs C0 %% %d == 0 and src1", DDPM3D_CONV_CK);
        if (!aligned16(d->src0) || (d->src1 && !aligned16(d->src1)))
            return fail(DDPM3D_EINVAL, "conv3d: sources must be 16-byte aligned");
        if (d->in_mode == DDPM3D_IN_UP && ((d->H | d->W) & 1))
            return fail(DDPM3D_EINVAL, "conv3d: upsampled input needs even H, W (got %d, %d)", d->H, d->W);
    }
    if ((d->aff_a == nullptr) != (d->aff_b == nullptr)) return fail(DDPM3D_EINVAL, "conv3d: aff_a/aff_b must come together");
    if (d->aff_a && (!aligned16(d->aff_a) || !aligned16(d->aff_b)))
        return fail(DDPM3D_EINVAL, "conv3d: affine tables must be 16-byte aligned");
    if (!aligned16(d->w_packed)) return fail(DDPM3D_EINVAL, "conv3d: w_packed must be 16-byte aligned");
    if (d->res_mode < 0 || d->res_mode > 3 || (d->res_mode != DDPM3D_RES_NONE && !d->res))
        return fail(DDPM3D_EINVAL, "conv3d: residual mode %d without / with bad buffer", d->res_mode);
    if (d->res_mode == DDPM3D_RES_UP && ((d->H | d->W) & 1))
        return fail(DDPM3D_EINVAL, "conv3d: upsampled residual needs even H, W");
    if (d->bias_stride_n < 0) return fail(DDPM3D_EINVAL, "conv3d: negative bias_stride_n");
    if (d->out_layout != DDPM3D_OUT_NDHWC && d->out_layout != DDPM3D_OUT_NCDHW)
        return fail(DDPM3D_EINVAL, "conv3d: out_layout %d", d->out_layout);

    c = ddpm3d_conv_cfg(d->N, d->D, d->H, d->W, d->Cin, d->Cout, d->ksize, d->precision);
    if (d->kernel_hint & DDPM3D_HINT_SPLITK_MASK) {
        // measurements (tools/splitk_sweep.py): force the split factor; the caller sizes statistics and workspace
        // with ddpm3d_conv_plan on the same descriptor
        const int s_forced = (d->kernel_hint & DDPM3D_HINT_SPLITK_MASK) >> DDPM3D_HINT_SPLITK_SHIFT;
        const int nch = ddpm3d_cin_pad(d->Cin) / DDPM3D_CONV_CK;
        if (s_forced > nch || (s_forced != c.S && c.WN != 4))
            return fail(DDPM3D_EINVAL, "conv3d: a forced split factor needs Cout > 64 and S <= Cin / 16");
        ddpm3d_conv_cfg_split(c, s_forced, d->N, d->D, d->H, d->W, d->Cout);
    }
    if (prec_wz(d->precision) &&
        !(wz_layer_ok(d->Cout, d->Cin, d->ksize) && c.WN == 4 && c.MT == 4 &&
          (d->in_mode == DDPM3D_IN_SAME || d->in_mode == DDPM3D_IN_UP)))
        return fail(DDPM3D_ENOSUP, "conv3d: the Winograd-D form needs ksize 3, Cout %% 128 == 0 "
                                   "and input mode SAME or UP; use the F16X3 packing for this call");
    if (for_launch && c.S > 1 && (!d->workspace || d->workspace_bytes < c.workspace_bytes || !aligned16(d->workspace)))
        return fail(DDPM3D_EINVAL, "conv3d: this shape is split %d-way over Cin and needs %zu bytes of "
                                   "16-byte aligned workspace (got %zu)", c.S, c.workspace_bytes,
                    d->workspace ? d->workspace_bytes : (size_t)0);
    memset(&k, 0, sizeof(k));
    k.src0 = d->src0; k.src1 = d->src1; k.affA = d->aff_a; k.affB = d->aff_b;
    k.w = (const float*)d->w_packed; k.bias = d->bias; k.res = d->res; k.out = d->out; k.stats = d->stats;
    k.N = d->N; k.D = d->D; k.H = d->H; k.W = d->W; k.Cin = d->Cin; k.Cout = d->Cout;
    k.C0 = d->C0; k.C1 = d->C1;
    k.CinPad = ddpm3d_cin_pad(d->Cin); k.CoutPad = ddpm3d_cout_pad(d->Cout);
    k.in_mode = d->in_mode; k.act = d->act; k.bias_stride_n = d->bias_stride_n;
    k.res_mode = d->res_mode; k.out_layout = d->out_layout;
    k.tilesZ = c.tilesZ; k.tilesY = c.tilesY; k.tilesX = c.tilesX;
    k.stats_rows = c.stats_rows;
    k.reduce_vox = c.reduce_vox;
    k.hint = d->kernel_hint;
    k.io = d->io_dtype;
    if (d->io_dtype & ~(DDPM3D_IO_SRC0_BF16 | DDPM3D_IO_SRC1_BF16 | DDPM3D_IO_OUT_BF16 | DDPM3D_IO_RES_BF16 |
                        DDPM3D_IO_HALF_IS_F16))
        return fail(DDPM3D_EINVAL, "conv3d: unknown io_dtype bits %#x", d->io_dtype);
    if ((d->io_dtype & DDPM3D_IO_OUT_BF16) && d->out_layout != DDPM3D_OUT_NDHWC)
        return fail(DDPM3D_EINVAL, "conv3d: a 16-bit output needs the NDHWC layout");
    if ((d->io_dtype & (DDPM3D_IO_SRC0_BF16 | DDPM3D_IO_SRC1_BF16)) && d->in_mode == DDPM3D_IN_PLANAR2)
        return fail(DDPM3D_EINVAL, "conv3d: the planar input volumes are fp32");
    if (prec_scaled(d->precision)) {
        if (!d->in_bound || d->in_bound_count <= 0 || d->in_bound_count > 64 || d->in_bound_stride <= 0)
            return fail(DDPM3D_EINVAL, "conv3d: the split-f16 precisions need in_bound (1..64 entries per sample): "
                                       "the activation scale is derived from it, nothing is clamped");
        k.in_bound = d->in_bound;
        k.in_bound_count = d->in_bound_count;
        k.in_bound_stride = d->in_bound_stride;
    }
    k.ksplit = c.S;
    k.chunks_per_split = (k.CinPad / DDPM3D_CONV_CK + c.S - 1) / c.S;
    // 1x1 convs: whole 32-channel blocks per split (conv1x1.hip walks K in those; any range suits the general kernel)
    if (d->ksize == 1 && c.S > 1) k.chunks_per_split = (k.chunks_per_split + 1) & ~1;
    k.partial = (float*)d->workspace;
    {
        // extents for the kernel's buffer descriptors (32-bit offsets)
        const bool dbl = d->in_mode == DDPM3D_IN_POOL || d->in_mode == DDPM3D_IN_STRIDE2;
        const long long Hs = dbl ? 2LL * d->H : (d->in_mode == DDPM3D_IN_UP ? d->H / 2 : d->H);
        const long long Ws = dbl ? 2LL * d->W : (d->in_mode == DDPM3D_IN_UP ? d->W / 2 : d->W);
        const long long vox = (long long)d->N * d->D * Hs * Ws;
        const long long b0 = vox * d->C0 * ((d->io_dtype & DDPM3D_IO_SRC0_BF16) ? 2 : 4);
        const long long b1 = vox * d->C1 * ((d->io_dtype & DDPM3D_IO_SRC1_BF16) ? 2 : 4);
        const size_t wb = ddpm3d_packed_bytes(d->Cout, d->Cin, d->ksize, d->precision);
        if (b0 >= 0xFFFFFFF0LL || b1 >= 0xFFFFFFF0LL || wb >= 0xFFFFFFF0ULL)
            return fail(DDPM3D_E2BIG, "conv3d: a source tensor or the weights exceed 4 GiB; split the batch");
        // the epilogue addresses one SAMPLE of the output (or residual) with 32-bit offsets
        if ((long long)d->D * d->H * d->W * d->Cout * 4 >= 0xFFFFFFF0LL)
            return fail(DDPM3D_E2BIG, "conv3d: one output sample exceeds 4 GiB; tile the volume");
        k.src0_bytes = (unsigned)b0; k.src1_bytes = (unsigned)b1; k.w_bytes = (unsigned)wb;
        // workgroup -> XCD order: keep on one XCD whichever operand is the larger stream
        k.wstat = (d->kernel_hint & DDPM3D_HINT_WSTAT_ON)    ? 1
                  : (d->kernel_hint & DDPM3D_HINT_WSTAT_OFF) ? 0
                                                             : ((long long)wb > b0 + b1 ? 1 : 0);
    }
    if (c.PREC != DDPM3D_PREC_F32)  // output scales sit behind the f16 image
        k.wscale = (const float*)((const char*)d->w_packed +
                                  ddpm3d_packed_bytes(d->Cout, d->Cin, d->ksize, d->precision) -
                                  (size_t)ddpm3d_cout_pad(d->Cout) * 4);
    if (d->stats && d->stats_rows != k.stats_rows)
        return fail(DDPM3D_EINVAL, "conv3d: stats_rows=%d, this shape writes %d", d->stats_rows, k.stats_rows);
    if (d->stats && !aligned16(d->stats)) return fail(DDPM3D_EINVAL, "conv3d: stats must be 16-byte aligned");
    if (d->stats && d->out_layout != DDPM3D_OUT_NDHWC)
        return fail(DDPM3D_EINVAL, "conv3d: statistics only with NDHWC output");
    const long long blocks = (long long)k.N * k.tilesZ * k.tilesY * k.tilesX;
    if (blocks > 0x7fffffffLL) return fail(DDPM3D_EINVAL, "conv3d: grid too large");
    // one or two output channels (the network's last layer): its own kernel (conv3d_skinny.hip)
    if (d->ksize == 3 && d->Cout <= 2 && d->in_mode == DDPM3D_IN_SAME && d->C1 == 0 && !d->stats &&
        d->res_mode == DDPM3D_RES_NONE &&
        ddpm3d_skinny_ok(k.CinPad, d->precision, !(d->io_dtype & DDPM3D_IO_SRC0_BF16) ? 0
                                                  : ((d->io_dtype & DDPM3D_IO_HALF_IS_F16) ? 2 : 1)))
        route = ROUTE_SKINNY;
    else
        route = ddpm3d_pw_ok(k, c, d->ksize) ? ROUTE_PW : ROUTE_GENERAL;
    return DDPM3D_OK;
}

int ddpm3d_conv3d(const ddpm3d_conv_desc* d, void* stream) {
    ConvK k;
    ConvCfg c;
    int route = ROUTE_GENERAL;
    const int ok = conv_prepare(d, k, c, route, true);
    if (ok != DDPM3D_OK) return ok;
    if (route == ROUTE_SKINNY)
        return launched(ddpm3d_launch_conv_skinny(k, d->precision, (hipStream_t)stream), "conv3d (skinny)");
    const int rc = route == ROUTE_PW ? launched(ddpm3d_launch_conv_pw(k, c, (hipStream_t)stream), "conv3d (1x1)")
                                     : launched(ddpm3d_launch_conv(k, c, (hipStream_t)stream), "conv3d");
    if (rc != DDPM3D_OK || c.S == 1) return rc;
    return launched(ddpm3d_launch_splitk_reduce(k, (hipStream_t)stream), "conv3d split-K reduce");
}

int ddpm3d_conv_plan(const ddpm3d_conv_desc* d, int* stats_rows, size_t* workspace_bytes, int* split) {
    ConvK k;
    ConvCfg c;
    int route = ROUTE_GENERAL;
    const int ok = conv_prepare(d, k, c, route, false);
    if (ok != DDPM3D_OK) return ok;
    if (stats_rows) *stats_rows = c.stats_rows;
    if (workspace_bytes) *workspace_bytes = c.workspace_bytes;
    if (split) *split = c.S;
    return DDPM3D_OK;
}

int ddpm3d_conv_kernel_family(const ddpm3d_conv_desc* d, char* name, int name_len) {
    if (!name || name_len <= 0) return fail(DDPM3D_EINVAL, "conv_kernel_family: no buffer");
    ConvK k;
    ConvCfg c;
    int route = ROUTE_GENERAL;
    const int ok = conv_prepare(d, k, c, route, false);   // (the split-K workspace may be attached later)
    if (ok != DDPM3D_OK) return ok;
    const int tile = 1 << c.TXL;    // 8: 8x8x2 / 8x4x4 tiles, 4: 4x4x8
    int n;
    if (route == ROUTE_SKINNY) n = snprintf(name, (size_t)name_len, "conv3d_p%d_k3_skinny", d->precision);
    else if (route == ROUTE_PW) n = snprintf(name, (size_t)name_len, "conv1x1_p%d_t%d", d->precision, tile);
    else n = snprintf(name, (size_t)name_len, "conv3d_p%d_k%d_wn%d_t%d", d->precision, d->ksize, c.WN, tile);
    if (n < 0 || n >= name_len) return fail(DDPM3D_EINVAL, "conv_kernel_family: buffer of %d bytes is too short", name_len);
    return DDPM3D_OK;
}

int ddpm3d_gn_finalize(const double* stats0, int C0, int rows0, const double* stats1, int C1, int rows1,
                       int N, int groups, double count, float eps, const float* gamma, const float* beta,
                       const float* film, int film_stride, int film_off, float* aff_a, float* aff_b,
                       float* bound, void* stream) {
    const int C = C0 + C1;
    if (!stats0 || N <= 0 || groups <= 0 || C0 <= 0 || C1 < 0 || rows0 <= 0 || count <= 0)
        return fail(DDPM3D_EINVAL, "gn_finalize: bad arguments");
    if (!aligned16(stats0) || (stats1 && !aligned16(stats1)))
        return fail(DDPM3D_EINVAL, "gn_finalize: statistics must be 16-byte aligned");
    if (gamma ? (!beta || !aff_a || !aff_b) : !bound)
        return fail(DDPM3D_EINVAL, "gn_finalize: gamma needs beta, aff_a, aff_b; without gamma only `bound` is written");
    if (C % groups) return fail(DDPM3D_EINVAL, "gn_finalize: C=%d not divisible by %d groups", C, groups);
    const int cg = C / groups;
    if (C1 > 0 && (!stats1 || rows1 <= 0 || C0 % cg))
        return fail(DDPM3D_EINVAL, "gn_finalize: a group straddles the concat boundary (C0=%d, group=%d)", C0, cg);
    return launched(ddpm3d_launch_gn_finalize(stats0, C0, rows0, stats1, C1, rows1, N, groups, count, eps,
                                              gamma, beta, film, film_stride, film_off, aff_a, aff_b, bound,
                                              (hipStream_t)stream),
                    "gn_finalize");
}

int ddpm3d_absmax(const float* x0, const float* x1, int N, size_t per_sample, float* bound, void* stream) {
    if (!x0 || !bound || N <= 0 || per_sample == 0) return fail(DDPM3D_EINVAL, "absmax: bad arguments");
    return launched(ddpm3d_launch_absmax(x0, x1, N, per_sample, bound, (hipStream_t)stream), "absmax");
}

int ddpm3d_gn_stats_rows(int voxels) { return voxels > 0 ? ddpm3d_gn_stats_rows_impl(voxels) : 0; }

int ddpm3d_gn_stats(const float* x, int N, int voxels, int C, double* stats, void* stream) {
    if (!x || !stats || N <= 0 || voxels <= 0 || C <= 0 || (C & 3) || !aligned16(x) || !aligned16(stats))
        return fail(DDPM3D_EINVAL, "gn_stats: bad arguments (C must be a multiple of 4, x 16-byte aligned)");
    return launched(ddpm3d_launch_gn_stats(x, N, voxels, C, stats, (hipStream_t)stream), "gn_stats");
}

int ddpm3d_timestep_embedding(const float* t, int rows, int dim, const float* freqs, float* out, void* stream) {
    if (!t || !out || !freqs || rows <= 0 || dim <= 1) return fail(DDPM3D_EINVAL, "timestep_embedding: bad arguments");
    return launched(ddpm3d_launch_timestep_embedding(t, rows, dim, freqs, out, (hipStream_t)stream),
                    "timestep_embedding");
}

int ddpm3d_linear(const float* in, int rows, int K, const float* w, const float* bias, int O, int silu_in,
                  float* out, int out_stride, void* stream) {
    if (!in || !w || !bias || !out || rows <= 0 || K <= 0 || O <= 0 || out_stride < O)
        return fail(DDPM3D_EINVAL, "linear: bad arguments");
    return launched(ddpm3d_launch_linear(in, rows, K, w, bias, O, silu_in, out, out_stride, (hipStream_t)stream),
                    "linear");
}

int ddpm3d_pool_act(const void* src, const float* aff_a, const float* aff_b, int act, int fast_act, int N, int D,
                    int H, int W, int C, void* out, int io_dtype, void* stream) {
    if (!src || !out || N <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3))
        return fail(DDPM3D_EINVAL, "pool_act: bad arguments (C must be a multiple of 4)");
    if ((aff_a == nullptr) != (aff_b == nullptr)) return fail(DDPM3D_EINVAL, "pool_act: aff_a/aff_b must come together");
    if (act && !aff_a) return fail(DDPM3D_EINVAL, "pool_act: an activation needs the affine tables (pass A = 1, B = 0)");
    if (io_dtype & ~(DDPM3D_IO_SRC0_BF16 | DDPM3D_IO_OUT_BF16 | DDPM3D_IO_HALF_IS_F16))
        return fail(DDPM3D_EINVAL, "pool_act: io_dtype bits %#x", io_dtype);
    if (!aligned16(src) || !aligned16(out) || (aff_a && (!aligned16(aff_a) || !aligned16(aff_b))))
        return fail(DDPM3D_EINVAL, "pool_act: buffers must be 16-byte aligned");
    if ((long long)N * D * H * W * (C / 4) > 0x7fffffffLL * 256)
        return fail(DDPM3D_E2BIG, "pool_act: grid too large; split the batch");
    return launched(ddpm3d_launch_pool_act((const float*)src, aff_a, aff_b, act, fast_act, N, D, H, W, C, (float*)out,
                                           (io_dtype & DDPM3D_IO_SRC0_BF16) != 0, (io_dtype & DDPM3D_IO_OUT_BF16) != 0,
                                           (io_dtype & DDPM3D_IO_HALF_IS_F16) != 0, (hipStream_t)stream),
                    "pool_act");
}

int ddpm3d_add_embedding(float* emb, const float* table, const int64_t* idx, int rows, int dim, int num_classes,
                         void* stream) {
    if (!emb || !table || !idx || rows <= 0 || dim <= 0 || num_classes <= 0)
        return fail(DDPM3D_EINVAL, "add_embedding: bad arguments");
    (void)num_classes;   // the indices live on the device; the caller validates their range
    return launched(ddpm3d_launch_add_embedding(emb, table, idx, rows, dim, num_classes, (hipStream_t)stream), "add_embedding");
}

int ddpm3d_ncdhw_to_ndhwc(const float* in, int N, int C, int voxels, float* out, void* stream) {
    if (!in || !out || N <= 0 || C <= 0 || voxels <= 0) return fail(DDPM3D_EINVAL, "ncdhw_to_ndhwc: bad arguments");
    return launched(ddpm3d_launch_transpose(in, N, C, voxels, out, (hipStream_t)stream), "ncdhw_to_ndhwc");
}
int ddpm3d_ndhwc_to_ncdhw(const float* in, int N, int C, int voxels, float* out, void* stream) {
    if (!in || !out || N <= 0 || C <= 0 || voxels <= 0) return fail(DDPM3D_EINVAL, "ndhwc_to_ncdhw: bad arguments");
    return launched(ddpm3d_launch_transpose(in, N, voxels, C, out, (hipStream_t)stream), "ndhwc_to_ncdhw");
}

int ddpm3d_ncdhw_to_ndhwc_pad(const float* in, int N, int C, int voxels, int Cpad, float* out, void* stream) {
    if (!in || !out || N <= 0 || C <= 0 || voxels <= 0 || Cpad < C)
        return fail(DDPM3D_EINVAL, "ncdhw_to_ndhwc_pad: bad arguments");
    return launched(ddpm3d_launch_to_ndhwc_pad(in, N, C, voxels, Cpad, out, (hipStream_t)stream), "ncdhw_to_ndhwc_pad");
}

int ddpm3d_subsample_hw2(const float* in, int N, int D, int H, int W, int C, float* out, void* stream) {
    if (!in || !out || N <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || (H & 1) || (W & 1) || (C & 3) ||
        !aligned16(in) || !aligned16(out))
        return fail(DDPM3D_EINVAL, "subsample_hw2: needs even H, W, C %% 4 == 0 and 16-byte aligned tensors");
    return launched(ddpm3d_launch_subsample_hw2(in, N, D, H, W, C, out, (hipStream_t)stream), "subsample_hw2");
}

int ddpm3d_attention_p(const float* qkv, int N, int T, int heads, int head_channels, int precision,
                       const float* qkv_bound, int bound_count, int bound_stride, float* out, void* stream) {
    if (!qkv || !out || N <= 0 || T <= 0 || heads <= 0) return fail(DDPM3D_EINVAL, "attention: bad arguments");
    if (head_channels != 32 && head_channels != 64 && head_channels != 128)
        return fail(DDPM3D_ENOSUP, "attention: %d channels per head (32, 64 or 128 are built)", head_channels);
    if (precision != DDPM3D_PREC_F32 && precision != DDPM3D_PREC_F16X3)
        return fail(DDPM3D_ENOSUP, "attention: precision %d (F32 and F16X3 are built)", precision);
    if (!aligned16(qkv) || !aligned16(out)) return fail(DDPM3D_EINVAL, "attention: buffers must be 16-byte aligned");
    if (precision != DDPM3D_PREC_F32 && (!qkv_bound || bound_count <= 0 || bound_count > 64 || bound_stride <= 0))
        return fail(DDPM3D_EINVAL, "attention: the split-f16 arithmetic needs qkv_bound (1..64 entries per sample)");
    return launched(ddpm3d_launch_attention(qkv, N, T, heads, head_channels, precision, qkv_bound, bound_count,
                                            bound_stride, out, (hipStream_t)stream),
                    "attention");
}

int ddpm3d_attention(const float* qkv, int N, int T, int heads, int head_channels, float* out, void* stream) {
    return ddpm3d_attention_p(qkv, N, T, heads, head_channels, DDPM3D_PREC_F32, nullptr, 0, 0, out, stream);
}

static int step_args_ok(const float* mo, const float* x, const float* noise, const float* coef,
                        const int64_t* t, int N, int voxels, float* sample) {
    return mo && x && noise && coef && t && sample && N > 0 && voxels > 0;
}

int ddpm3d_p_sample_step(const float* model_out, const float* x, const float* noise, const float* coef,
                         const int64_t* t_idx, int N, int voxels, int flags, float* sample,
                         float* pred_xstart, void* stream) {
    if (!step_args_ok(model_out, x, noise, coef, t_idx, N, voxels, sample))
        return fail(DDPM3D_EINVAL, "p_sample_step: bad arguments");
    return launched(ddpm3d_launch_sample_step(false, model_out, x, noise, coef, t_idx, N, voxels, flags, 0.0f,
                                              sample, pred_xstart, (hipStream_t)stream),
                    "p_sample_step");
}

int ddpm3d_ddim_step(const float* model_out, const float* x, const float* noise, const float* coef,
                     const int64_t* t_idx, int N, int voxels, int flags, float eta, float* sample,
                     float* pred_xstart, void* stream) {
    if (!step_args_ok(model_out, x, noise, coef, t_idx, N, voxels, sample))
        return fail(DDPM3D_EINVAL, "ddim_step: bad arguments");
    return launched(ddpm3d_launch_sample_step(true, model_out, x, noise, coef, t_idx, N, voxels, flags, eta,
                                              sample, pred_xstart, (hipStream_t)stream),
                    "ddim_step");
}

double ddpm3d_mfma_probe_flops_per_iter(int kind) { return ddpm3d_probe_flops_per_iter(kind); }

int ddpm3d_mfma_probe(int kind, int iters, int blocks, float* out, uint64_t* clocks, void* stream) {
    if (!out || !clocks || iters <= 0 || blocks <= 0 || ddpm3d_probe_flops_per_iter(kind) == 0.0)
        return fail(DDPM3D_EINVAL, "mfma_probe: bad arguments (kind=%d iters=%d blocks=%d)", kind, iters, blocks);
    return launched(ddpm3d_launch_mfma_probe(kind, iters, blocks, out, (unsigned long long*)clocks,
                                             (hipStream_t)stream), "mfma_probe");
}

}  // extern "C"
