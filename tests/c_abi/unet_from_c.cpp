// The whole-network level of the C ABI from a host that is neither Python nor torch: a small SuperRes UNet
// (first conv on the two input planes, one encoder ResBlock, two middle ResBlocks, two decoder ResBlocks on the
// virtual concat [h, skip] with their 1x1 skip convs, GroupNorm + SiLU + last conv; FiLM conditioning) is
// described with ddpm3d_unet_desc, compiled by ddpm3d_unet_plan_create into one hipMalloc'ed arena and run by
// ddpm3d_unet_forward; the result is compared with a plain fp64 evaluation of the same network on the host
// (unet.py:1015-1044, :236-256, nn.py:93-100).  Built and run by tests/test_gpu_script.py.
//
//   hipcc -I include tests/c_abi/unet_from_c.cpp -L 3d-denoising-diffusion-model_amd/csrc -lddpm3d -o unet_from_c
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "ddpm3d.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
#define CHECK_ABI(x) do { int r_ = (x); if (r_ != DDPM3D_OK) { printf("ddpm3d error %d (%s / %s) at line %d\n", r_, ddpm3d_last_error(), ddpm3d_unet_last_error(), __LINE__); return 1; } } while (0)

static unsigned g_seed = 12345u;
static float rnd() {   // uniform in [-1, 1)
    g_seed = g_seed * 1664525u + 1013904223u;
    return (float)((g_seed >> 8) & 0xFFFFFF) / 8388608.0f - 1.0f;
}

enum { D = 4, H = 8, W = 8, VOX = D * H * W, MC = 32 };

struct Conv { int Co, Ci, k; std::vector<float> w, b; void* d_packed = nullptr; float* d_b = nullptr; };
struct Norm { int C; std::vector<float> g, b; float* d_g = nullptr; float* d_b = nullptr; };

static Conv make_conv(int Co, int Ci, int k) {
    Conv c; c.Co = Co; c.Ci = Ci; c.k = k;
    c.w.resize((size_t)Co * Ci * k * k * k); c.b.resize(Co);
    const float s = 1.0f / sqrtf((float)(Ci * k * k * k));
    for (auto& v : c.w) v = rnd() * s;
    for (auto& v : c.b) v = rnd() * 0.1f;
    return c;
}
static Norm make_norm(int C) {
    Norm n; n.C = C; n.g.resize(C); n.b.resize(C);
    for (auto& v : n.g) v = 1.0f + 0.2f * rnd();
    for (auto& v : n.b) v = 0.2f * rnd();
    return n;
}
static int upload(Conv& c, hipStream_t st) {
    float* dw;
    CHECK_HIP(hipMalloc(&dw, c.w.size() * 4));
    CHECK_HIP(hipMemcpy(dw, c.w.data(), c.w.size() * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMalloc(&c.d_b, c.b.size() * 4));
    CHECK_HIP(hipMemcpy(c.d_b, c.b.data(), c.b.size() * 4, hipMemcpyHostToDevice));
    const size_t n = ddpm3d_packed_weight_bytes(c.Co, c.Ci, c.k, DDPM3D_PREC_F32);
    if (!n) { printf("packed_weight_bytes refused %dx%d k%d\n", c.Co, c.Ci, c.k); return 1; }
    CHECK_HIP(hipMalloc(&c.d_packed, n));
    CHECK_ABI(ddpm3d_pack_conv_weight(dw, c.Co, c.Ci, c.k, DDPM3D_PREC_F32, c.d_packed, st));
    CHECK_HIP(hipStreamSynchronize(st));
    CHECK_HIP(hipFree(dw));
    return 0;
}
static int upload(Norm& n) {
    CHECK_HIP(hipMalloc(&n.d_g, n.C * 4)); CHECK_HIP(hipMemcpy(n.d_g, n.g.data(), n.C * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMalloc(&n.d_b, n.C * 4)); CHECK_HIP(hipMemcpy(n.d_b, n.b.data(), n.C * 4, hipMemcpyHostToDevice));
    return 0;
}
static ddpm3d_conv_weights weights_of(const Conv& c) {
    ddpm3d_conv_weights w;
    memset(&w, 0, sizeof(w));
    w.w_packed = c.d_packed; w.bias = c.d_b; w.Cout = c.Co; w.Cin = c.Ci; w.ksize = c.k; w.precision = DDPM3D_PREC_F32;
    return w;
}

// ---- the host evaluation, fp64, NCDHW
typedef std::vector<double> T;
static T conv(const T& x, const Conv& c) {
    T y((size_t)c.Co * VOX);
    const int r = c.k / 2;
    for (int co = 0; co < c.Co; ++co)
        for (int z = 0; z < D; ++z) for (int yy = 0; yy < H; ++yy) for (int xx = 0; xx < W; ++xx) {
            double a = c.b[co];
            for (int ci = 0; ci < c.Ci; ++ci)
                for (int dz = -r; dz <= r; ++dz) for (int dy = -r; dy <= r; ++dy) for (int dx = -r; dx <= r; ++dx) {
                    const int z2 = z + dz, y2 = yy + dy, x2 = xx + dx;
                    if (z2 < 0 || z2 >= D || y2 < 0 || y2 >= H || x2 < 0 || x2 >= W) continue;
                    a += (double)c.w[((((size_t)co * c.Ci + ci) * c.k + dz + r) * c.k + dy + r) * c.k + dx + r] *
                         x[(size_t)ci * VOX + (z2 * H + y2) * W + x2];
                }
            y[(size_t)co * VOX + (z * H + yy) * W + xx] = a;
        }
    return y;
}
// SiLU(GroupNorm32(x) [* (1 + scale) + shift])
static T norm_act(const T& x, const Norm& n, const float* film) {
    const int cg = n.C / 32;
    T y(x.size());
    for (int g = 0; g < 32; ++g) {
        double s1 = 0, s2 = 0;
        for (int c = g * cg; c < (g + 1) * cg; ++c) for (int v = 0; v < VOX; ++v) { const double t = x[(size_t)c * VOX + v]; s1 += t; s2 += t * t; }
        const double cnt = (double)cg * VOX, mean = s1 / cnt, var = s2 / cnt - mean * mean, rstd = 1.0 / sqrt(var + 1e-5);
        for (int c = g * cg; c < (g + 1) * cg; ++c)
            for (int v = 0; v < VOX; ++v) {
                double t = (x[(size_t)c * VOX + v] - mean) * rstd * n.g[c] + n.b[c];
                if (film) t = t * (1.0 + film[c]) + film[n.C + c];
                y[(size_t)c * VOX + v] = t / (1.0 + exp(-t));
            }
    }
    return y;
}
struct Res { Norm n1, n2; Conv c1, c2, skip; bool has_skip; int film_off; };
static T resblock(const T& x, const Res& r, const float* film_row) {
    T h = conv(norm_act(x, r.n1, nullptr), r.c1);
    h = conv(norm_act(h, r.n2, film_row + r.film_off), r.c2);
    T s = r.has_skip ? conv(x, r.skip) : x;
    for (size_t i = 0; i < h.size(); ++i) h[i] += s[i];
    return h;
}
static T cat(const T& a, const T& b) { T c(a); c.insert(c.end(), b.begin(), b.end()); return c; }

int main() {
    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));
    if (ddpm3d_abi_version() != DDPM3D_ABI_VERSION) { printf("ABI mismatch\n"); return 1; }

    // ---- parameters
    Conv first = make_conv(MC, 2, 3), last = make_conv(2, MC, 3);
    Norm out_norm = make_norm(MC);
    std::vector<Res> res(5);
    int film_total = 0;
    for (int i = 0; i < 5; ++i) {
        const int cin = i >= 3 ? 2 * MC : MC;             // the two decoder blocks read the concat [h, skip]
        res[i].n1 = make_norm(cin); res[i].c1 = make_conv(MC, cin, 3);
        res[i].n2 = make_norm(MC); res[i].c2 = make_conv(MC, MC, 3);
        res[i].has_skip = cin != MC;
        if (res[i].has_skip) res[i].skip = make_conv(MC, cin, 1);
        res[i].film_off = film_total; film_total += 2 * MC;
    }
    std::vector<float> film(film_total), x(VOX), lr(VOX);
    for (auto& v : film) v = 0.3f * rnd();
    for (auto& v : x) v = rnd();
    for (auto& v : lr) v = 0.5f + 0.5f * rnd();

    // ---- host evaluation
    T in(2 * VOX);
    for (int v = 0; v < VOX; ++v) { in[v] = x[v]; in[VOX + v] = lr[v]; }
    T h0 = conv(in, first);
    T h1 = resblock(h0, res[0], film.data());
    T m = resblock(resblock(h1, res[1], film.data()), res[2], film.data());
    T d0 = resblock(cat(m, h1), res[3], film.data());          // hs.pop() = h1, then h0 (unet.py:1040-1042)
    T d1 = resblock(cat(d0, h0), res[4], film.data());
    T ref = conv(norm_act(d1, out_norm, nullptr), last);

    // ---- device side: upload, describe, plan, run
    if (upload(first, st) || upload(last, st) || upload(out_norm)) return 1;
    for (auto& r : res) {
        if (upload(r.n1) || upload(r.n2) || upload(r.c1, st) || upload(r.c2, st)) return 1;
        if (r.has_skip && upload(r.skip, st)) return 1;
    }
    std::vector<ddpm3d_layer> layers(5);
    for (int i = 0; i < 5; ++i) {
        ddpm3d_layer& L = layers[i];
        memset(&L, 0, sizeof(L));
        L.kind = DDPM3D_LAYER_RES; L.updown = DDPM3D_UPDOWN_NONE; L.film_off = res[i].film_off;
        L.norm1_gamma = res[i].n1.d_g; L.norm1_beta = res[i].n1.d_b; L.norm2_gamma = res[i].n2.d_g; L.norm2_beta = res[i].n2.d_b;
        L.conv1 = weights_of(res[i].c1); L.conv2 = weights_of(res[i].c2);
        if (res[i].has_skip) L.skip = weights_of(res[i].skip);
    }
    const int32_t in_blocks[1] = {1}, out_blocks[2] = {1, 1};
    ddpm3d_unet_desc md;
    memset(&md, 0, sizeof(md));
    md.n_layers = 5; md.layers = layers.data();
    md.n_input_blocks = 1; md.input_block_layers = in_blocks;
    md.n_middle_layers = 2;
    md.n_output_blocks = 2; md.output_block_layers = out_blocks;
    md.first = weights_of(first); md.out = weights_of(last);
    md.out_gamma = out_norm.d_g; md.out_beta = out_norm.d_b;
    md.film = 1; md.planar = 1; md.in_channels = 2; md.cin_pad = 16; md.arithmetic = DDPM3D_PREC_F32;

    const size_t bytes = ddpm3d_unet_plan_bytes(&md, 1, D, H, W);
    if (!bytes) { printf("plan_bytes refused: %s\n", ddpm3d_unet_last_error()); return 1; }
    void* arena;
    CHECK_HIP(hipMalloc(&arena, bytes));
    ddpm3d_unet_plan* plan = nullptr;
    CHECK_ABI(ddpm3d_unet_plan_create(&md, 1, D, H, W, arena, bytes, &plan));
    float *dx, *dlr, *dfilm, *dout;
    CHECK_HIP(hipMalloc(&dx, VOX * 4)); CHECK_HIP(hipMalloc(&dlr, VOX * 4));
    CHECK_HIP(hipMalloc(&dfilm, film_total * 4)); CHECK_HIP(hipMalloc(&dout, 2 * VOX * 4));
    CHECK_HIP(hipMemcpy(dx, x.data(), VOX * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dlr, lr.data(), VOX * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dfilm, film.data(), film_total * 4, hipMemcpyHostToDevice));
    std::vector<float> got(2 * VOX), again(2 * VOX);
    for (int rep = 0; rep < 2; ++rep) {              // twice: a plan is replayable, bit for bit
        CHECK_ABI(ddpm3d_unet_forward(plan, dx, dlr, dfilm, 0, dout, st));
        CHECK_HIP(hipStreamSynchronize(st));
        CHECK_HIP(hipMemcpy(rep ? again.data() : got.data(), dout, 2 * VOX * 4, hipMemcpyDeviceToHost));
    }
    double err = 0, mag = 0;
    for (int i = 0; i < 2 * VOX; ++i) { err = fmax(err, fabs((double)got[i] - ref[i])); mag = fmax(mag, fabs(ref[i])); }
    const bool same = memcmp(got.data(), again.data(), got.size() * 4) == 0;
    printf("unet forward from C: arena %zu bytes, max rel err vs fp64 host evaluation %.3e, replay %s\n", bytes, err / mag,
           same ? "bit-identical" : "DIFFERS");
    // a plan with a too-small arena is refused, not overrun
    ddpm3d_unet_plan* bad = nullptr;
    const int rc = ddpm3d_unet_plan_create(&md, 1, D, H, W, arena, bytes / 2, &bad);
    ddpm3d_unet_plan_destroy(plan);
    if (!(err / mag < 2e-5) || !same || rc != DDPM3D_EINVAL || !(mag > 0.1)) { printf("FAIL\n"); return 1; }
    printf("PASS\n");
    return 0;
}
