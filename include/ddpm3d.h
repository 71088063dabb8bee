/*
 * ddpm3d -- C ABI of the MI355X-native 3-D DDPM denoising sampler.
 *
 * The reference (Zachary-Luk/3D-Denoising-Diffusion-Model) has no FFI, plugin
 * or operator interface: its hot path is plain Python calling ATen.  The
 * entry points below are therefore what a binding for that path binds --
 * one per ATen-op family the path issues -- and each
 * one cites the reference lines whose arithmetic it replaces.  The Python
 * mirror of the reference API (3d-denoising-diffusion-model_amd/
 * guided_diffusion/) reaches them through ctypes; INTEGRATION.md shows the
 * stub.
 *
 * Conventions
 *   - Plain C: pointers are DEVICE pointers (hipMalloc / torch caching
 *     allocator), sizes are ints, `stream` is a hipStream_t passed as void*.
 *   - Every call only ENQUEUES work on `stream` (no allocation, no
 *     synchronisation, graph-capturable).  The caller owns every buffer and
 *     keeps it alive until the stream has passed the call.
 *   - Return 0 on success, a negative DDPM3D_E* code otherwise;
 *     ddpm3d_last_error() gives a thread-local message.  No C++ exception
 *     crosses the boundary.
 *   - Activations are channels-last fp32: [N][D][H][W][C] ("NDHWC").  The
 *     reference's NCDHW tensors are converted at the API edge only
 *     (ddpm3d_ncdhw_to_ndhwc / the planar input mode of the first conv /
 *     the NCDHW store mode of the last conv).
 *   - GroupNorm is never a pass of its own: every conv epilogue emits
 *     per-(sample, row-tile, channel) partial sums (sum, sum of squares;
 *     accumulated and stored in fp64, so that the variance survives a mean
 *     hundreds of standard deviations large) of what it stores;
 *     ddpm3d_gn_finalize folds them (fp64) into per-(n, c)
 *     affine coefficients A, B; the NEXT conv applies
 *     y = SiLU(A*x + B) while it stages its input tile into LDS.
 */
#ifndef DDPM3D_H
#define DDPM3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DDPM3D_ABI_VERSION 12

enum {
    DDPM3D_OK = 0,
    DDPM3D_EINVAL = -1,   /* bad descriptor (shape / alignment / mode)        */
    DDPM3D_ELAUNCH = -2,  /* HIP refused the launch                           */
    DDPM3D_ENOSUP = -3,   /* valid request the library does not implement    */
    DDPM3D_E2BIG = -4     /* a tensor of the call passes the 32-bit byte offsets the kernels
                             address with (4 GiB): split the batch / tile the volume and
                             call again -- nothing was enqueued                         */
};

/* input staging modes of ddpm3d_conv3d (where the conv's input voxel (z,y,x)
 * is read from) */
enum {
    DDPM3D_IN_SAME = 0,    /* source has the conv's D,H,W                              */
    DDPM3D_IN_POOL = 1,    /* source is D,2H,2W; input = mean of the 2x2 (H,W) window
                              of act(A*x+B)  -- Downsample, unet.py:129-136, applied
                              after norm+act and before the conv, unet.py:237-242      */
    DDPM3D_IN_UP = 2,      /* source is D,H/2,W/2; nearest (y>>1, x>>1) -- Upsample,
                              unet.py:102-105                                          */
    DDPM3D_IN_PLANAR2 = 3, /* src0, src1 are two single-channel NCDHW volumes (x and
                              low_res, unet.py:1690-1693); Cin = 2                     */
    DDPM3D_IN_STRIDE2 = 4  /* source is D,2H,2W and the conv has stride (1,2,2), pad 1:
                              Downsample(use_conv=True), unet.py:129-133 (ksize 3)     */
};

/* residual modes of the conv epilogue (out = conv + bias + residual) */
enum {
    DDPM3D_RES_NONE = 0,
    DDPM3D_RES_SAME = 1,   /* skip(x) + h with x at the output resolution, unet.py:256 */
    DDPM3D_RES_POOL = 2,   /* x_upd = Downsample(x), unet.py:241                       */
    DDPM3D_RES_UP = 3      /* x_upd = Upsample(x),   unet.py:241                       */
};

enum { DDPM3D_ACT_NONE = 0, DDPM3D_ACT_SILU = 1 };
enum { DDPM3D_OUT_NDHWC = 0, DDPM3D_OUT_NCDHW = 1 };

/*
 * 3x3x3 (pad 1, stride 1) or 1x1x1 convolution as an implicit GEMM on the
 * matrix cores, with everything element-wise around it fused in.
 * Replaces, per call: nn.Conv3d (nn.py:22-32; unet.py:185,211,222,810,996),
 * the preceding GroupNorm32-apply + SiLU (+ FiLM) (nn.py:93-100,
 * unet.py:183-184,207-208,248-252), AvgPool3d / nearest Upsample
 * (unet.py:102-105,129-136), th.cat of the skip (unet.py:1041), the residual
 * add (unet.py:256) and the statistics pass of the FOLLOWING GroupNorm.
 */
typedef struct ddpm3d_conv_desc {
    /* geometry of the convolution's OUTPUT grid */
    int32_t N, D, H, W;
    int32_t Cin;            /* = C0 + C1                                             */
    int32_t Cout;
    int32_t ksize;          /* 3 or 1                                                */
    int32_t in_mode;        /* DDPM3D_IN_*                                           */
    /* virtual concat along C: channels [0,C0) from src0, [C0,Cin) from src1 */
    const float* src0;
    const float* src1;      /* NULL when C1 == 0                                     */
    int32_t C0, C1;
    /* prologue y = act(A[n][c]*x + B[n][c]); aff_a == NULL -> y = x */
    const float* aff_a;     /* [N][Cin]                                              */
    const float* aff_b;     /* [N][Cin]                                              */
    int32_t act;            /* DDPM3D_ACT_*                                          */
    int32_t precision;      /* DDPM3D_PREC_* (must match how w_packed was packed)    */
    /* weights in the layout ddpm3d_pack_conv_weight produces; bias [Cout]
     * (bias_stride_n = 0), or one row of Cout values per sample, rows
     * bias_stride_n floats apart (additive timestep embedding, unet.py:254-255) */
    const void* w_packed;
    const float* bias;
    int32_t bias_stride_n;
    int32_t res_mode;       /* DDPM3D_RES_*                                          */
    const float* res;       /* [N][..][Cout] NDHWC at the resolution res_mode implies */
    float* out;
    int32_t out_layout;     /* DDPM3D_OUT_*                                          */
    int32_t stats_rows;     /* rows per sample of `stats` (from ddpm3d_conv_stats_rows) */
    double* stats;          /* [N][Cout][stats_rows][2] fp64 (sum, sum of squares), channel-major,
                               16-byte aligned, or NULL                               */
    /* scratch for split-K partial sums (low-resolution levels, where the voxel
     * tiles alone cannot fill 256 CUs); >= ddpm3d_conv_workspace_bytes(...) bytes,
     * may be shared by all convs of a stream, NULL when that query returns 0 */
    void* workspace;
    size_t workspace_bytes;
    /* 0 = the library picks the workgroup order from the shape.  DDPM3D_HINT_* bits select among
     * launch orders of IDENTICAL arithmetic (bit-identical outputs); they exist for tests and A/B
     * measurements and never change a result. */
    int32_t kernel_hint;
    /* Range of the convolution's INPUT as the matrix cores see it, for the split-f16 modes (every
     * precision but DDPM3D_PREC_F32, which ignores it; REQUIRED otherwise).  Sample n's entries
     * in_bound[(n * in_bound_count + i) * in_bound_stride], i < in_bound_count <= 64, are upper
     * bounds of |act(A*x + B)| over (parts of) that sample's input; the kernel takes their maximum b
     * and scales the activations by the power of two that puts b just below 2^15 before the f16
     * hi/lo split (the epilogue multiplies by the exact inverse).  So no input magnitude is clipped
     * and small-magnitude tensors keep their low bits; a bound that is too SMALL makes f16 overflow
     * and the output non-finite (never silently wrong).  Producers: ddpm3d_gn_finalize (normalised
     * and raw bounds from the GroupNorm partial sums), ddpm3d_absmax (tensors without statistics). */
    int32_t in_bound_count;
    const float* in_bound;
    int32_t in_bound_stride;
    /* DDPM3D_IO_* bits: which of the activation tensors hold 16-bit (bf16, or f16 with
     * DDPM3D_IO_HALF_IS_F16) instead of fp32 elements (same NDHWC layout, 2 bytes per element, 8-byte
     * aligned).  Statistics, affine tables, bias and
     * the NCDHW output are fp32 always.  This is the bf16 mode's placement of the reference's
     * fp16 torso (unet.py:999-1005, :1035, :1043): the residual stream in 16 bits, GroupNorm and the
     * edges of the network in fp32. */
    int32_t io_dtype;
} ddpm3d_conv_desc;

enum {
    DDPM3D_IO_SRC0_BF16 = 1,
    DDPM3D_IO_SRC1_BF16 = 2,
    DDPM3D_IO_OUT_BF16 = 4,    /* with DDPM3D_OUT_NDHWC only */
    DDPM3D_IO_RES_BF16 = 8,
    /* the tensors flagged above hold IEEE f16 instead of bf16 (all of them): the reference's own
     * --use_fp16 storage of the torso (unet.py:1035 `h = x.type(self.dtype)`, fp16_util.py:15-22).
     * Values beyond 65504 become inf, as they do there. */
    DDPM3D_IO_HALF_IS_F16 = 16
};

/* ddpm3d_conv_desc.kernel_hint */
enum {
    DDPM3D_HINT_WSTAT_OFF = 0x100,/* workgroup -> XCD order: tiles fastest (activation-stationary)   */
    DDPM3D_HINT_WSTAT_ON = 0x200, /*   cout blocks / K splits fastest (weight-stationary)            */
    /* bits 12..14: issue order of a tap in the f16x3 Winograd-D kernel (conv3d_wz.h, IL + 1; 0 = the
     * library picks by shape) */
    DDPM3D_HINT_WZ_ORDER_SHIFT = 12,
    DDPM3D_HINT_WZ_ORDER_MASK = 0x7000,
    /* bits 16..21: force the split factor over Cin of a conv with Cout > 64 (measurement only: size statistics
     * and workspace with ddpm3d_conv_plan on the same descriptor; 0 = the library's own choice) */
    DDPM3D_HINT_SPLITK_SHIFT = 16,
    DDPM3D_HINT_SPLITK_MASK = 0x3F0000
};

int ddpm3d_abi_version(void);
const char* ddpm3d_last_error(void);

/* Arithmetic of the convolution's products.  Inputs, outputs and accumulators are
 * fp32 in every mode.
 *   DDPM3D_PREC_F32   v_mfma_f32_32x32x2_f32: exact fp32 products.
 *   DDPM3D_PREC_F16X3  every fp32 operand x is split hi + lo into two f16 (after a
 *                      power-of-two scaling) and a*b = hi*hi + hi*lo + lo*hi on
 *                      v_mfma_f32_32x32x16_f16 (each f16xf16 product is exact in
 *                      fp32).  Operand representation error ~2^-23, i.e. below the
 *                      fp32 accumulation error both modes share; 16/3 the MFMA rate.
 *   DDPM3D_PREC_F16   one f16 MFMA per product on the f16-ROUNDED (scaled) operands,
 *                      fp32 accumulate: the analogue of the reference's --use_fp16 torso
 *                      (unet.py:999-1005, fp16_util.py:15-22); ~2^-11 operand error, judged
 *                      by PSNR, not by the 1e-3 parity bar.  Same packed image as F16X3. */
enum {
    DDPM3D_PREC_F32 = 0,
    DDPM3D_PREC_F16X3 = 1,
    DDPM3D_PREC_F16 = 2,
    /* F16X3 arithmetic on the Winograd F(2,3)-along-depth form of a 3x3x3 conv: 4 products
     * per two outputs instead of 6 (the weights are transformed at pack time, the inputs
     * while they are staged, the outputs in the epilogue).  Available for ksize 3, Cout a
     * multiple of 128, input modes SAME / UP (tiles of 128 voxels: 8x4x4 -- two z-pairs per workgroup -- where
     * H % 8 == 0 and D % 4 == 0, otherwise 8x8x2 where H and W >= 8, and 4x4x8 -- four z-pairs --
     * below that); other calls return DDPM3D_ENOSUP and must use the F16X3
     * packing of the same weights. */
    DDPM3D_PREC_F16X3_WZ = 3,
    /* F16 arithmetic (one MFMA per product on f16-rounded operands, as DDPM3D_PREC_F16) on the
     * same Winograd-D form and the same packed image as DDPM3D_PREC_F16X3_WZ (its hi halves);
     * same availability rule. */
    DDPM3D_PREC_F16_WZ = 4,
    /* One bf16 MFMA per product on bf16-ROUNDED operands (8 significant bits), fp32 accumulate: the
     * arithmetic BASELINE config 4 names.  bf16 has fp32's exponent range, so no scaling and no
     * in_bound; judged by PSNR like DDPM3D_PREC_F16.  Own packed image (hi parts only). */
    DDPM3D_PREC_BF16 = 5,
    DDPM3D_PREC_BF16_WZ = 6    /* the same on the Winograd-D form; availability as F16X3_WZ */
};

/* bytes of the packed form of an (Cout, Cin, k, k, k) weight for a precision mode */
size_t ddpm3d_packed_weight_bytes(int Cout, int Cin, int ksize, int precision);
/* OIDHW (torch Conv3d.weight / Conv1d.weight with k=1) -> packed; device to device */
int ddpm3d_pack_conv_weight(const float* w_oidhw, int Cout, int Cin, int ksize, int precision,
                            void* w_packed, void* stream);

/* rows per sample of the statistics buffer this conv writes, and the scratch
 * it needs (both depend on how the shape is tiled / split; `precision` = the DDPM3D_PREC_* of the call
 * since ABI 12: the split over Cin is chosen per arithmetic mode) */
int ddpm3d_conv_stats_rows(int N, int D, int H, int W, int Cin, int Cout, int ksize, int precision);
size_t ddpm3d_conv_workspace_bytes(int N, int D, int H, int W, int Cin, int Cout, int ksize, int precision);
int ddpm3d_conv3d(const ddpm3d_conv_desc* desc, void* stream);
/* Which kernel family ddpm3d_conv3d runs this descriptor on, as a NUL-terminated name:
 * "conv3d_p<precision>_k<ksize>_wn<waves along Cout>_t<tile width>" (direct and Winograd-D forms),
 * "conv1x1_p<precision>_t<tile width>" (the register-fed 1x1 GEMM), "conv3d_p<precision>_k3_skinny" (Cout <= 2).
 * Validates like ddpm3d_conv3d (the split-K workspace excepted) and launches nothing: measurement
 * bookkeeping for callers that attribute time per family (ABI 12). */
int ddpm3d_conv_kernel_family(const ddpm3d_conv_desc* desc, char* name, int name_len);
/* How ddpm3d_conv3d will run THIS descriptor (kernel_hint included): statistics rows per sample, workspace bytes
 * and the split factor over Cin.  Any out pointer may be NULL.  Validates like ddpm3d_conv_kernel_family;
 * launches nothing (ABI 12). */
int ddpm3d_conv_plan(const ddpm3d_conv_desc* desc, int* stats_rows, size_t* workspace_bytes, int* split);

/*
 * ---- The whole network (ABI 12; SURVEY 8b's `unet_forward(handle, ...)` granularity) ----------------------------
 * A UNet forward -- UNetModel[_noatt] / SuperResModel[_noatt].forward, unet.py:1015-1044, :687-716, :1687-1694;
 * ResBlock :236-256, AttentionBlock :296-305, Downsample / Upsample :102-105, :129-136, TimestepEmbedSequential
 * :72-78 -- compiled ONCE per (model, N, D, H, W) into a flat list of the per-op calls of this header, every
 * intermediate buffer carved out of ONE caller-provided device arena, and replayed by ddpm3d_unet_forward: what the
 * Python host's launch plan (guided_diffusion/engine.py) does, for hosts that are not Python.  Same calls, same
 * arguments, same order: bit-identical to the Python plan.  The timestep path (timestep_embedding, time_embed, the
 * fused emb_layers Linear: "film rows") stays with the caller, who evaluates it for all steps of a schedule at once.
 *
 * The description borrows every pointer (packed weights, biases, GroupNorm parameters: device memory that must
 * outlive the plan); layers are listed in execution order.
 */
typedef struct ddpm3d_conv_weights {
    const void* w_packed;      /* ddpm3d_pack_conv_weight image in `precision`; NULL = layer absent             */
    const void* w_packed_wz;   /* the Winograd-D image of the same layer (`precision_wz`), or NULL               */
    const float* bias;         /* [Cout]                                                                         */
    int32_t Cout, Cin, ksize;
    int32_t precision, precision_wz;
} ddpm3d_conv_weights;

enum { DDPM3D_LAYER_RES = 1, DDPM3D_LAYER_ATTN = 2, DDPM3D_LAYER_DOWNCONV = 3, DDPM3D_LAYER_UPCONV = 4 };
enum { DDPM3D_UPDOWN_NONE = 0, DDPM3D_UPDOWN_DOWN = 1, DDPM3D_UPDOWN_UP = 2 };

typedef struct ddpm3d_layer {
    int32_t kind;              /* DDPM3D_LAYER_*                                                                 */
    int32_t updown;            /* ResBlock(down=True / up=True), unet.py:187-197                                 */
    int32_t heads;             /* attention heads                                                                */
    int32_t film_off;          /* ResBlock: offset of its emb_layers output inside a film row                    */
    const float* norm1_gamma;  /* ResBlock in_layers.0 / AttentionBlock norm                                     */
    const float* norm1_beta;
    const float* norm2_gamma;  /* ResBlock out_layers.0                                                          */
    const float* norm2_beta;
    ddpm3d_conv_weights conv1; /* ResBlock in_layers.2 / attention qkv / Downsample op / Upsample conv           */
    ddpm3d_conv_weights conv2; /* ResBlock out_layers.3 / attention proj_out                                     */
    ddpm3d_conv_weights skip;  /* ResBlock skip_connection (w_packed NULL = Identity)                            */
} ddpm3d_layer;

typedef struct ddpm3d_unet_desc {
    int32_t n_layers;
    const ddpm3d_layer* layers;              /* input blocks 1.., middle block, output blocks, in execution order */
    int32_t n_input_blocks;                  /* input blocks AFTER block 0 (the first conv)                       */
    const int32_t* input_block_layers;       /* layers per input block                                            */
    int32_t n_middle_layers;
    int32_t n_output_blocks;
    const int32_t* output_block_layers;
    ddpm3d_conv_weights first;               /* input_blocks.0.0                                                  */
    const float* out_gamma;                  /* out.0                                                             */
    const float* out_beta;
    ddpm3d_conv_weights out;                 /* out.2 (stored NCDHW)                                              */
    int32_t film;                            /* use_scale_shift_norm (unet.py:248-255)                            */
    int32_t planar;                          /* the first conv reads x and low_res as two planes (SuperRes)       */
    int32_t in_channels, cin_pad;            /* otherwise: (N, in_channels, voxels) input, padded to cin_pad      */
    int32_t arithmetic;                      /* DDPM3D_PREC_F32 / _F16X3 / _F16 / _BF16: which convs need input
                                                bounds, and whether the residual stream is stored in 16 bits      */
} ddpm3d_unet_desc;

typedef struct ddpm3d_unet_plan ddpm3d_unet_plan;
/* bytes of device arena a plan of this model and shape needs (0 = refused: ddpm3d_unet_last_error) */
size_t ddpm3d_unet_plan_bytes(const ddpm3d_unet_desc* model, int N, int D, int H, int W);
/* arena: 256-byte aligned device memory of at least that size, owned by the caller, private to the plan */
int ddpm3d_unet_plan_create(const ddpm3d_unet_desc* model, int N, int D, int H, int W, void* arena, size_t arena_bytes,
                            ddpm3d_unet_plan** plan);
/* x (and low_res when planar): (N, 1 or in_channels, D, H, W) fp32; film_rows: row n at film_rows + n * film_stride
 * (0 = one row for the batch) holds the fused emb_layers output of sample n's timestep; out: (N, Cout, D, H, W).
 * Enqueue-only, like every call it is made of; one forward of a plan at a time. */
int ddpm3d_unet_forward(ddpm3d_unet_plan* plan, const float* x, const float* low_res, const float* film_rows,
                        int film_stride, float* out, void* stream);
void ddpm3d_unet_plan_destroy(ddpm3d_unet_plan* plan);
const char* ddpm3d_unet_last_error(void);

/*
 * GroupNorm32 statistics -> affine coefficients (nn.py:93-100: 32 groups,
 * eps 1e-5, affine gamma/beta), optionally composed with FiLM
 * h*(1+scale)+shift (unet.py:248-252).  The normalised tensor is the virtual
 * concat of up to two tensors with partial sums stats0 [N][C0][rows0][2] and
 * stats1 [N][C1][rows1][2]; `count` = voxels per channel.
 *   A[n][c] = rstd*gamma[c]*(1+scale[n][c]);
 *   B[n][c] = (beta[c]-mean*rstd*gamma[c])*(1+scale[n][c]) + shift[n][c]
 * film = [N][film_stride] rows holding scale at [film_off, +C) and shift at
 * [film_off + C, +C); NULL -> no FiLM.
 */
int ddpm3d_gn_finalize(const double* stats0, int C0, int rows0,
                       const double* stats1, int C1, int rows1,
                       int N, int groups, double count, float eps,
                       const float* gamma, const float* beta,
                       const float* film, int film_stride, int film_off,
                       float* aff_a, float* aff_b, float* bound, void* stream);
/* `bound` (may be NULL) = [N][groups][2] upper bounds for ddpm3d_conv_desc.in_bound, from the same
 * partial sums: a value of channel c is at most xmax = sqrt(largest sum of squares of any statistics
 * row of its group), so
 *   bound[n][g][0] = max_c |A[n][c]| * xmax + max_c |B[n][c]|   >= |act(A*x + B)|  (|SiLU(y)| <= |y|)
 *   bound[n][g][1] = xmax                                       >= |x|  (the tensor read raw)
 * gamma == NULL (then beta, film, aff_a, aff_b are ignored): bounds only, for tensors that are
 * consumed without a GroupNorm. */

/* bound[n * count + t] = max |x| over sample n of tensor t, for up to two tensors of `per_sample`
 * floats per sample (count = 1 or 2; x1 may be NULL): ddpm3d_conv_desc.in_bound of inputs that have
 * no statistics (the first conv's x and low_res volumes, attention outputs). */
int ddpm3d_absmax(const float* x0, const float* x1, int N, size_t per_sample, float* bound, void* stream);

/* partial sums of an NDHWC tensor that no conv epilogue produced;
 * stats [N][C][rows][2] with rows = ddpm3d_gn_stats_rows(voxels) */
int ddpm3d_gn_stats_rows(int voxels);
int ddpm3d_gn_stats(const float* x, int N, int voxels, int C, double* stats, void* stream);

/* nn.py:103-121 timestep_embedding: out[r] = [cos(t_r f) | sin(t_r f)] (+0 pad).
 * freqs = [dim/2] fp32 table.  The reference evaluates
 * exp(-ln(max_period) * arange(half) / half) on the HOST and moves it to the
 * device (nn.py:113-115), so the table is a host-computed constant here too. */
int ddpm3d_timestep_embedding(const float* t, int rows, int dim, const float* freqs,
                              float* out, void* stream);
/* nn.Linear (time_embed unet.py:799-803, emb_layers unet.py:199-205):
 * out[r][o] = bias[o] + sum_k f(in[r][k]) * w[o][k],  f = SiLU if silu_in */
int ddpm3d_linear(const float* in, int rows, int K, const float* w, const float* bias,
                  int O, int silu_in, float* out, int out_stride, void* stream);

/* The down-sampling ResBlock's h_upd(in_rest(x)) (unet.py:194-195, :238-242: GroupNorm32 + SiLU, then
 * Downsample(use_conv=False) = AvgPool3d((1,2,2))) as a pass of its own:
 *   out[n][z][y][x][c] = mean over (2y + {0,1}, 2x + {0,1}) of act(aff_a[n][c] * src[n][z][.][.][c] + aff_b[n][c])
 * src = [N][D][2H][2W][C], out = [N][D][H][W][C] (H, W = the OUTPUT extents), NDHWC, C % 4 == 0; window order
 * ((s00 + s01) + s10) + s11, then * 1/4 -- the DDPM3D_IN_POOL prologue of ddpm3d_conv3d, which computes exactly
 * this while staging.  As a separate pass it lets the conv that follows read a plain tensor (DDPM3D_IN_SAME, no
 * affine) and so run its Winograd-D form.  aff_a / aff_b NULL = no affine (then act must be 0); fast_act != 0 =
 * the v_exp / v_rcp SiLU of the non-exact conv modes (what their own prologue evaluates), 0 = expf and an IEEE
 * divide.  io_dtype: DDPM3D_IO_SRC0_BF16 / DDPM3D_IO_OUT_BF16 / DDPM3D_IO_HALF_IS_F16 as in ddpm3d_conv_desc. */
int ddpm3d_pool_act(const void* src, const float* aff_a, const float* aff_b, int act, int fast_act, int N, int D,
                    int H, int W, int C, void* out, int io_dtype, void* stream);

/* Class conditioning (unet.py:476-478, :703-705): emb[r][:] += table[idx[r]][:] with table =
 * label_emb.weight [num_classes][dim] and idx = the batch's labels (int64, device).  A label outside
 * [0, num_classes) adds nothing to its row (never an out-of-bounds read, ABI 12); a caller that wants
 * nn.Embedding's error checks the labels itself, as the Python host does once per sampling loop. */
int ddpm3d_add_embedding(float* emb, const float* table, const int64_t* idx, int rows, int dim, int num_classes,
                         void* stream);

/*
 * Self-attention core of AttentionBlock (unet.py:296-305) with the legacy head layout
 * (QKVAttentionLegacy, unet.py:337-354): qkv = [N][T][heads*3*ch] (per head: q | k | v).
 * The other order (QKVAttention, unet.py:361-389: q | k | v each heads*ch wide, same scale, same
 * softmax, same output order) differs only in which rows of the qkv 1x1 conv feed which slot: the
 * host permutes that conv's output channels when it packs the weights and calls this same kernel.
 * qkv layout here:
 * out = [N][T][heads*ch];  out = softmax_fp32((q s)^T (k s)) v^T with s = ch^-1/4.
 * Streaming softmax: the T x T weight matrix the reference materialises (:349-353) never
 * exists.  The GroupNorm and the qkv / proj_out 1x1 convs around it are ddpm3d_conv3d calls.
 */
int ddpm3d_attention(const float* qkv, int N, int T, int heads, int head_channels,
                     float* out, void* stream);
/* The same with the arithmetic of the two products chosen like a conv's: DDPM3D_PREC_F32 (exact fp32
 * MFMA, what ddpm3d_attention runs) or DDPM3D_PREC_F16X3 (fp32-grade products from three f16 MFMAs
 * on hi/lo-split q, k, v and softmax weights; 16/3 of the fp32 MFMA rate).  The softmax itself is
 * fp32 in both (unet.py:351). */
int ddpm3d_attention_p(const float* qkv, int N, int T, int heads, int head_channels, int precision,
                       const float* qkv_bound, int bound_count, int bound_stride,
                       float* out, void* stream);
/* qkv_bound: as ddpm3d_conv_desc.in_bound, upper bounds of |qkv| per sample (required for
 * DDPM3D_PREC_F16X3, ignored for DDPM3D_PREC_F32). */

/* layout changes at the API edge */
int ddpm3d_ncdhw_to_ndhwc(const float* in, int N, int C, int voxels, float* out, void* stream);
int ddpm3d_ndhwc_to_ncdhw(const float* in, int N, int C, int voxels, float* out, void* stream);
/* (N, C, voxels) -> (N, voxels, Cpad) with channels [C, Cpad) zero: the network input of the models
 * whose first conv reads an ordinary multi-channel tensor (create_model's RGB UNetModel,
 * script_util.py:130-184), padded to the convs' 16-channel granularity. */
int ddpm3d_ncdhw_to_ndhwc_pad(const float* in, int N, int C, int voxels, int Cpad, float* out, void* stream);

/* out[n][z][y][x][:] = in[n][z][2y][2x][:] on NDHWC tensors (even H, W; C % 4 == 0).
 * Downsample(use_conv=True) (unet.py:129-133, `resblock_updown=False`): a 3x3x3 conv with
 * stride (1,2,2), pad 1 is the stride-1 conv kept at the even (y, x) -- ddpm3d_conv3d at full
 * resolution, this call, then ddpm3d_gn_stats for the next GroupNorm.  (Correct, not fast:
 * 3/4 of that conv's work is discarded; the published model uses resblock_updown=True.) */
int ddpm3d_subsample_hw2(const float* in, int N, int D, int H, int W, int C, float* out, void* stream);

/*
 * One reverse-diffusion update for a batch (everything after the network call):
 * gaussian_diffusion.py:262-326 (p_mean_variance), :430-438 (p_sample) and
 * :566-584 (ddim_sample).  coef = [T][DDPM3D_NCOEF] fp32 table (fp64-computed,
 * fp32-applied like _extract_into_tensor, :897-910); t_idx[n] selects the row.
 * model_out is NCDHW (N, 2 or 1, voxels); x, noise, sample, pred_xstart are
 * (N, 1, voxels).  pred_xstart may be NULL.
 */
enum {
    DDPM3D_C_SQRT_RECIP_ACP = 0,
    DDPM3D_C_SQRT_RECIPM1_ACP = 1,
    DDPM3D_C_POST_MEAN_COEF1 = 2,
    DDPM3D_C_POST_MEAN_COEF2 = 3,
    DDPM3D_C_MIN_LOG = 4,       /* posterior_log_variance_clipped; FIXED_*: the fixed log-variance */
    DDPM3D_C_MAX_LOG = 5,       /* log(betas)                                                      */
    DDPM3D_C_ACP = 6,
    DDPM3D_C_ACP_PREV = 7,
    DDPM3D_NCOEF = 8
};
enum {
    DDPM3D_F_LEARN_SIGMA = 1,   /* ModelVarType.LEARNED_RANGE (model_out has 2 channels) */
    DDPM3D_F_PREDICT_XSTART = 2,
    DDPM3D_F_CLIP = 4
};
int ddpm3d_p_sample_step(const float* model_out, const float* x, const float* noise,
                         const float* coef, const int64_t* t_idx, int N, int voxels,
                         int flags, float* sample, float* pred_xstart, void* stream);
int ddpm3d_ddim_step(const float* model_out, const float* x, const float* noise,
                     const float* coef, const int64_t* t_idx, int N, int voxels,
                     int flags, float eta, float* sample, float* pred_xstart, void* stream);

/*
 * Device calibration (measurement only; replaces nothing in the reference).  Enqueues a
 * register-only MFMA loop -- no memory traffic, pseudo-random operands, `blocks` workgroups of four
 * waves, each wave holding the dominant conv kernel's 64 x 32 x 4 fp32 accumulator tile -- so the
 * caller can time what THIS device sustains on the matrix pipes under its power cap (boards of one
 * pool differ by several per cent) and price a kernel against it (bench.py:
 * roofline.device_sustained_tflops).  out: blocks*256 floats (checksums, keeps the loop alive);
 * clocks: blocks*2 uint64 = {shader cycles, 100 MHz ticks} spent in the loop per workgroup
 * (in-kernel clock = cycles / ticks * 0.1 GHz).  FLOPs issued = blocks * iters *
 * ddpm3d_mfma_probe_flops_per_iter(kind).
 */
enum {
    DDPM3D_PROBE_F16_32X32X16 = 0,   /* v_mfma_f32_32x32x16_f16: what the f16x3 / f16 convs issue */
    DDPM3D_PROBE_F16_16X16X32 = 1,   /* v_mfma_f32_16x16x32_f16                                  */
    DDPM3D_PROBE_F32_32X32X2 = 2,    /* v_mfma_f32_32x32x2_f32: the exact mode                    */
    DDPM3D_PROBE_BF16_32X32X16 = 3,
    DDPM3D_PROBE_BF16_16X16X32 = 4
};
double ddpm3d_mfma_probe_flops_per_iter(int kind);
int ddpm3d_mfma_probe(int kind, int iters, int blocks, float* out, uint64_t* clocks, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DDPM3D_H */
