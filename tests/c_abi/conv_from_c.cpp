// The C ABI used WITHOUT Python or torch: a host program (hipcc, or any C/C++ compiler + the HIP runtime
// for allocation) links libddpm3d.so, runs one fused convolution -- GroupNorm affine + SiLU prologue,
// 3x3x3, 32 -> 32 channels on a 4x8x8 volume, residual, GroupNorm partial sums -- in the exact fp32 and
// in the f16x3 arithmetic, and checks both against a scalar loop on the host.  Built and run by
// tests/test_gpu_script.py::test_c_abi_from_plain_cpp; what INTEGRATION.md section 2 describes, in C.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ddpm3d.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
#define CHECK_ABI(x) do { int r_ = (x); if (r_ != DDPM3D_OK) { printf("ddpm3d error %d: %s (line %d)\n", r_, ddpm3d_last_error(), __LINE__); return 3; } } while (0)

static float frand(unsigned* s) { *s = *s * 1664525u + 1013904223u; return ((*s >> 8) & 0xffff) / 32768.0f - 1.0f; }

int main(void) {
    if (ddpm3d_abi_version() != DDPM3D_ABI_VERSION) { printf("ABI mismatch\n"); return 1; }
    const int N = 1, D = 4, H = 8, W = 8, Ci = 32, Co = 32, vox = D * H * W;
    float *x = (float*)malloc(sizeof(float) * vox * Ci), *w = (float*)malloc(sizeof(float) * Co * Ci * 27);
    float *b = (float*)malloc(sizeof(float) * Co), *A = (float*)malloc(sizeof(float) * Ci), *B = (float*)malloc(sizeof(float) * Ci);
    float *res = (float*)malloc(sizeof(float) * vox * Co), *ref = (float*)malloc(sizeof(float) * vox * Co);
    float *got = (float*)malloc(sizeof(float) * vox * Co), *act = (float*)malloc(sizeof(float) * vox * Ci);
    unsigned s = 12345u;
    for (int i = 0; i < vox * Ci; ++i) x[i] = frand(&s);                 /* NDHWC */
    for (int i = 0; i < Co * Ci * 27; ++i) w[i] = 0.05f * frand(&s);     /* OIDHW */
    for (int i = 0; i < Co; ++i) b[i] = frand(&s);
    for (int i = 0; i < Ci; ++i) { A[i] = 1.0f + 0.1f * frand(&s); B[i] = 0.1f * frand(&s); }
    for (int i = 0; i < vox * Co; ++i) res[i] = frand(&s);
    float amax = 0.0f;
    for (int v = 0; v < vox; ++v)
        for (int c = 0; c < Ci; ++c) {
            const float y = x[v * Ci + c] * A[c] + B[c];
            act[v * Ci + c] = y / (1.0f + expf(-y));
            amax = fmaxf(amax, fabsf(act[v * Ci + c]));
        }
    for (int z = 0; z < D; ++z) for (int y = 0; y < H; ++y) for (int xx = 0; xx < W; ++xx)
        for (int co = 0; co < Co; ++co) {
            double acc = b[co];
            for (int dz = 0; dz < 3; ++dz) for (int dy = 0; dy < 3; ++dy) for (int dx = 0; dx < 3; ++dx) {
                const int zz = z + dz - 1, yy = y + dy - 1, xq = xx + dx - 1;
                if (zz < 0 || zz >= D || yy < 0 || yy >= H || xq < 0 || xq >= W) continue;
                const float* a = act + ((zz * H + yy) * W + xq) * Ci;
                for (int ci = 0; ci < Ci; ++ci) acc += (double)a[ci] * w[((co * Ci + ci) * 3 + dz) * 9 + dy * 3 + dx];
            }
            const int v = (z * H + y) * W + xx;
            ref[v * Co + co] = (float)acc + res[v * Co + co];
        }

    float *dx, *dw, *db, *dA, *dB, *dres, *dout, *dbound;
    double* dstats;   /* GroupNorm partial sums are fp64 */
    CHECK_HIP(hipMalloc(&dx, sizeof(float) * vox * Ci));
    CHECK_HIP(hipMalloc(&dw, sizeof(float) * Co * Ci * 27));
    CHECK_HIP(hipMalloc(&db, sizeof(float) * Co));
    CHECK_HIP(hipMalloc(&dA, sizeof(float) * Ci));
    CHECK_HIP(hipMalloc(&dB, sizeof(float) * Ci));
    CHECK_HIP(hipMalloc(&dres, sizeof(float) * vox * Co));
    CHECK_HIP(hipMalloc(&dout, sizeof(float) * vox * Co));
    CHECK_HIP(hipMalloc(&dbound, sizeof(float)));
    CHECK_HIP(hipMemcpy(dx, x, sizeof(float) * vox * Ci, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dw, w, sizeof(float) * Co * Ci * 27, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(db, b, sizeof(float) * Co, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dA, A, sizeof(float) * Ci, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dB, B, sizeof(float) * Ci, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dres, res, sizeof(float) * vox * Co, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dbound, &amax, sizeof(float), hipMemcpyHostToDevice));   /* upper bound of |act| */
    /* sized for the larger of the two arithmetic modes run below (the split over Cin is chosen per mode) */
    int rows = ddpm3d_conv_stats_rows(N, D, H, W, Ci, Co, 3, DDPM3D_PREC_F16X3);
    if (ddpm3d_conv_stats_rows(N, D, H, W, Ci, Co, 3, DDPM3D_PREC_F32) != rows) { printf("FAIL: rows differ by mode\n"); return 1; }
    CHECK_HIP(hipMalloc(&dstats, sizeof(double) * Co * rows * 2));
    size_t ws_bytes = ddpm3d_conv_workspace_bytes(N, D, H, W, Ci, Co, 3, DDPM3D_PREC_F16X3);
    if (ddpm3d_conv_workspace_bytes(N, D, H, W, Ci, Co, 3, DDPM3D_PREC_F32) > ws_bytes)
        ws_bytes = ddpm3d_conv_workspace_bytes(N, D, H, W, Ci, Co, 3, DDPM3D_PREC_F32);
    void* dws = NULL;
    if (ws_bytes) CHECK_HIP(hipMalloc(&dws, ws_bytes));
    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));

    const int precs[2] = {DDPM3D_PREC_F32, DDPM3D_PREC_F16X3};
    const double tol[2] = {2e-6, 4e-6};
    for (int m = 0; m < 2; ++m) {
        void* dwp;
        CHECK_HIP(hipMalloc(&dwp, ddpm3d_packed_weight_bytes(Co, Ci, 3, precs[m])));
        CHECK_ABI(ddpm3d_pack_conv_weight(dw, Co, Ci, 3, precs[m], dwp, st));
        ddpm3d_conv_desc d;
        memset(&d, 0, sizeof(d));
        d.N = N; d.D = D; d.H = H; d.W = W; d.Cin = Ci; d.Cout = Co; d.ksize = 3; d.in_mode = DDPM3D_IN_SAME;
        d.src0 = dx; d.C0 = Ci; d.aff_a = dA; d.aff_b = dB; d.act = DDPM3D_ACT_SILU; d.precision = precs[m];
        d.w_packed = dwp; d.bias = db; d.res_mode = DDPM3D_RES_SAME; d.res = dres; d.out = dout;
        d.out_layout = DDPM3D_OUT_NDHWC; d.stats = dstats; d.stats_rows = rows;
        d.workspace = dws; d.workspace_bytes = ws_bytes;
        d.in_bound = dbound; d.in_bound_count = 1; d.in_bound_stride = 1;
        CHECK_ABI(ddpm3d_conv3d(&d, st));
        CHECK_HIP(hipStreamSynchronize(st));
        CHECK_HIP(hipMemcpy(got, dout, sizeof(float) * vox * Co, hipMemcpyDeviceToHost));
        double err = 0.0, mag = 0.0;
        for (int i = 0; i < vox * Co; ++i) { err = fmax(err, fabs((double)got[i] - ref[i])); mag = fmax(mag, fabs((double)ref[i])); }
        double* stats = (double*)malloc(sizeof(double) * Co * rows * 2);
        CHECK_HIP(hipMemcpy(stats, dstats, sizeof(double) * Co * rows * 2, hipMemcpyDeviceToHost));
        double s1 = 0.0, r1 = 0.0;
        for (int r = 0; r < rows; ++r) s1 += stats[(0 * rows + r) * 2];          /* channel 0: sum over the volume */
        for (int v = 0; v < vox; ++v) r1 += ref[v * Co];
        printf("precision %d: max rel err %.3e, channel-0 sum %.6f (host %.6f)\n", precs[m], err / mag, s1, r1);
        if (!(err / mag < tol[m]) || fabs(s1 - r1) > 1e-3 * (1.0 + fabs(r1))) { printf("MISMATCH\n"); return 4; }
        free(stats);
        CHECK_HIP(hipFree(dwp));
    }
    /* errors are return codes with a message, never exceptions or aborts */
    ddpm3d_conv_desc bad;
    memset(&bad, 0, sizeof(bad));
    if (ddpm3d_conv3d(&bad, st) != DDPM3D_EINVAL || strlen(ddpm3d_last_error()) == 0) { printf("no error report\n"); return 5; }
    printf("C ABI OK\n");
    return 0;
}
