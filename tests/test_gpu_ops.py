"""
Per-kernel parity on the GPU, through the C ABI, against torch-CPU fp32
functional ops (the oracle's building blocks).  Tolerance: the fp32 MFMA is an
exact k-ordered fmaf chain, so differences vs the CPU are summation-order
rounding only -- 2e-5 relative to the output's max is the bar for K up to
~7000 terms.
"""

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err, rel_err_per_channel

pytestmark = pytest.mark.gpu

TOL = 2e-5


@pytest.fixture(scope="module")
def hc():
    import hipcall
    return hipcall


def rnd(*shape, seed=0, scale=1.0):
    g = np.random.default_rng(seed)
    return torch.from_numpy((g.standard_normal(shape) * scale).astype(np.float32))


def ref_stats(y_ncdhw):
    """(sum, sumsq) per (n, c) in fp64"""
    y = y_ncdhw.double()
    return y.sum(dim=(2, 3, 4)), (y * y).sum(dim=(2, 3, 4))


def check_stats(stats, y_ncdhw):
    s = stats.double().sum(dim=2).cpu()  # [N][C][rows][2] -> [N][C][2]
    r1, r2 = ref_stats(y_ncdhw)
    assert torch.allclose(s[..., 0], r1, rtol=1e-4, atol=1e-2)
    assert torch.allclose(s[..., 1], r2, rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("N,D,Hh,W,ci,co", [
    (1, 4, 8, 8, 16, 32),      # one tile, WN=1
    (2, 5, 12, 9, 32, 48),     # ragged in every dim, WN=2 with a partial cout tile
    (1, 3, 16, 16, 48, 128),   # WN=4
    (1, 9, 4, 4, 32, 96),      # 4x4 tiles (low-resolution levels), partial D tile, idle wave
    (1, 2, 2, 2, 16, 160),     # H,W below the tile, two cout blocks
    (1, 2, 6, 6, 32, 64),      # 96^3-patch style non-power-of-two H,W on 4x4 tiles
    (1, 16, 4, 4, 128, 128),   # low-resolution level: split-K over Cin + reduce kernel
    (2, 9, 8, 8, 96, 160),     # split-K with a ragged D tile, two cout blocks, batch 2
    (1, 64, 4, 4, 512, 384),   # the published net's 4x4 level shape (16-way split)
])
@pytest.mark.parametrize("precision", [0, 1])
def test_conv3d_k3_plain(hc, N, D, Hh, W, ci, co, precision):
    """precision 0 = exact fp32 MFMA; 1 = fp32 products as three f16 MFMAs.  Same tolerance:
    the split's operand error (~2^-23) is below the fp32 accumulation error both share."""
    x = rnd(N, ci, D, Hh, W, seed=1)
    w = rnd(co, ci, 3, 3, 3, seed=2, scale=0.05)
    b = rnd(co, seed=3)
    ref = F.conv3d(x, w, b, padding=1)
    out, stats, _ = hc.conv3d([hc.to_ndhwc(x).cuda()], w.cuda(), b.cuda(), (D, Hh, W), precision=precision)
    got = hc.to_ncdhw(out.cpu())
    assert rel_err(got.numpy(), ref.numpy()) < TOL
    check_stats(stats, ref)


@pytest.mark.parametrize("precision", [0, 1])
def test_conv3d_exact_integer_mapping(hc, precision):
    """Small-integer data makes every product and sum exact (in the split-f16 mode too: the
    scaled integers are exact f16 with lo = 0): any wrong tap, channel permutation or
    row/column swap shows up as a mismatch, not as rounding."""
    g = np.random.default_rng(5)
    x = torch.from_numpy(g.integers(-3, 4, (1, 16, 3, 9, 10)).astype(np.float32))
    w = torch.from_numpy(g.integers(-2, 3, (40, 16, 3, 3, 3)).astype(np.float32))
    b = torch.from_numpy(g.integers(-5, 6, (40,)).astype(np.float32))
    ref = F.conv3d(x, w, b, padding=1)
    out, _, _ = hc.conv3d([hc.to_ndhwc(x).cuda()], w.cuda(), b.cuda(), (3, 9, 10), precision=precision)
    assert torch.equal(hc.to_ncdhw(out.cpu()), ref)


def test_conv3d_f16x3_accuracy_is_fp32_grade(hc):
    """Against an fp64 reference, the split-f16 mode must be as accurate as the exact fp32
    mode (within 2x), on a K = 27*256 reduction with realistic operand ranges, including
    tiny weights / activations whose f16 low parts would be subnormal without the scaling."""
    x = rnd(1, 256, 4, 16, 16, seed=11)
    x = (x / (1 + torch.exp(-x)))                     # post-SiLU range
    x[:, :32] *= 1e-3                                  # some very small activations
    w = rnd(128, 256, 3, 3, 3, seed=12, scale=0.01)
    w[:16] *= 1e-3                                     # some very small output channels
    b = rnd(128, seed=13)
    ref = F.conv3d(x.double(), w.double(), b.double(), padding=1)
    errs = []
    for precision in (0, 1, 3):
        out, _, _ = hc.conv3d([hc.to_ndhwc(x).cuda()], w.cuda(), b.cuda(), (4, 16, 16), precision=precision)
        got = hc.to_ncdhw(out.cpu()).double()
        # per-output-channel relative error (small channels must not drown in the big ones)
        e = ((got - ref).abs().amax(dim=(0, 2, 3, 4)) / ref.abs().amax(dim=(0, 2, 3, 4))).max()
        errs.append(float(e))
    assert errs[0] < 2e-6 and errs[1] < 2e-6, errs
    assert errs[1] < 2 * errs[0] + 2e-7, errs
    # Winograd-D form: the transforms add a few roundings (F(2,3) is the benign Winograd size)
    assert errs[2] < 4e-6 and errs[2] < 4 * errs[0] + 4e-7, errs


@pytest.mark.parametrize("N,D,Hh,W,ci,co", [
    (1, 4, 16, 16, 48, 128),     # even D
    (2, 5, 8, 24, 32, 256),      # odd D (last z-pair half valid), batch 2, two cout blocks
    (1, 1, 9, 12, 16, 128),      # D = 1, ragged H / W
    (1, 64, 8, 8, 256, 384),     # the published net's 8x8 level: split-K + reduce kernel
    (1, 8, 8, 12, 32, 128),      # 8x4x4 tiles (H % 8 == 0, D % 4 == 0: two z-pairs per tile), ragged W
    (2, 4, 24, 16, 16, 256),     # 8x4x4 tiles, batch 2, six tile rows
    (1, 64, 4, 4, 512, 512),     # its 4x4 level: 4x4x8 tiles (four z-pairs per tile), split-K + reduce
    (2, 5, 4, 6, 32, 128),       # 4x4x8 tiles, ragged in all three extents (D = 5: pairs 2, 3 of the tile empty / half)
    (1, 11, 7, 3, 16, 256),      # W < 4, D not a multiple of 8, two cout blocks
])
def test_conv3d_winograd_depth_form(hc, N, D, Hh, W, ci, co):
    """precision 3: f16x3 arithmetic on the Winograd F(2,3)-along-depth form (weights transformed
    at pack time, inputs while staged, outputs in the epilogue).  Same tolerance as the direct form."""
    x = rnd(N, ci, D, Hh, W, seed=1)
    w = rnd(co, ci, 3, 3, 3, seed=2, scale=0.05)
    b = rnd(co, seed=3)
    ref = F.conv3d(x, w, b, padding=1)
    out, stats, _ = hc.conv3d([hc.to_ndhwc(x).cuda()], w.cuda(), b.cuda(), (D, Hh, W), precision=3)
    assert rel_err(hc.to_ncdhw(out.cpu()).numpy(), ref.numpy()) < TOL
    check_stats(stats, ref)


def test_conv3d_winograd_depth_exact_and_fused_paths(hc):
    """Integer data stays exact through the transforms (halves of small integers are exact); the
    GroupNorm+SiLU prologue, nearest-upsampled input / residual and concat work in this form too."""
    import guided_diffusion._hip as H
    g = np.random.default_rng(5)
    x = torch.from_numpy(g.integers(-3, 4, (1, 16, 5, 9, 10)).astype(np.float32))
    w = torch.from_numpy((2 * g.integers(-2, 3, (128, 16, 3, 3, 3))).astype(np.float32))
    b = torch.from_numpy(g.integers(-5, 6, (128,)).astype(np.float32))
    out, _, _ = hc.conv3d([hc.to_ndhwc(x).cuda()], w.cuda(), b.cuda(), (5, 9, 10), precision=3)
    assert torch.equal(hc.to_ncdhw(out.cpu()), F.conv3d(x, w, b, padding=1))
    # up-sampled block paths with a virtual concat of two sources
    xa, xb = rnd(1, 32, 3, 4, 6, seed=1), rnd(1, 32, 3, 4, 6, seed=2)
    xc = torch.cat([xa, xb], 1)
    gamma, beta = 1 + 0.1 * rnd(64, seed=3), 0.1 * rnd(64, seed=4)
    w2 = rnd(128, 64, 3, 3, 3, seed=6, scale=0.05)
    b2 = rnd(128, seed=7)
    r = rnd(1, 128, 3, 4, 6, seed=8)
    up = lambda t: F.interpolate(t, (t.shape[2], t.shape[3] * 2, t.shape[4] * 2), mode="nearest")
    ref = F.conv3d(up(F.silu(F.group_norm(xc, 32, gamma, beta, 1e-5))), w2, b2, padding=1) + up(r)
    A, B = _gn_affine(hc, [xa, xb], gamma, beta)
    out, stats, _ = hc.conv3d([hc.to_ndhwc(xa).cuda(), hc.to_ndhwc(xb).cuda()], w2.cuda(), b2.cuda(), (3, 8, 12),
                              in_mode=H.IN_UP, aff=(A, B), act=H.ACT_SILU, res=hc.to_ndhwc(r).cuda(),
                              res_mode=H.RES_UP, precision=3)
    assert rel_err(hc.to_ncdhw(out.cpu()).numpy(), ref.numpy()) < TOL
    check_stats(stats, ref)
    # the same on 8x4x4 tiles (H % 8 == 0, D % 4 == 0; r03): exact integers, up-sampled block path with a concat
    x8 = torch.from_numpy(g.integers(-3, 4, (1, 16, 8, 8, 10)).astype(np.float32))
    out, _, _ = hc.conv3d([hc.to_ndhwc(x8).cuda()], w.cuda(), b.cuda(), (8, 8, 10), precision=3)
    assert torch.equal(hc.to_ncdhw(out.cpu()), F.conv3d(x8, w, b, padding=1))
    xa, xb = rnd(1, 32, 4, 4, 6, seed=21), rnd(1, 32, 4, 4, 6, seed=22)
    xc = torch.cat([xa, xb], 1)
    r = rnd(1, 128, 4, 4, 6, seed=28)
    ref = F.conv3d(up(F.silu(F.group_norm(xc, 32, gamma, beta, 1e-5))), w2, b2, padding=1) + up(r)
    A, B = _gn_affine(hc, [xa, xb], gamma, beta)
    out, stats, _ = hc.conv3d([hc.to_ndhwc(xa).cuda(), hc.to_ndhwc(xb).cuda()], w2.cuda(), b2.cuda(), (4, 8, 12),
                              in_mode=H.IN_UP, aff=(A, B), act=H.ACT_SILU, res=hc.to_ndhwc(r).cuda(),
                              res_mode=H.RES_UP, precision=3)
    assert rel_err(hc.to_ncdhw(out.cpu()).numpy(), ref.numpy()) < TOL
    check_stats(stats, ref)
    # the same on 4x4x8 tiles (H, W < 8; r03): exact integers over a whole tile column, and the up-sampled
    # block path from a 2x3 grid
    x4 = torch.from_numpy(g.integers(-3, 4, (1, 16, 19, 4, 4)).astype(np.float32))
    out, _, _ = hc.conv3d([hc.to_ndhwc(x4).cuda()], w.cuda(), b.cuda(), (19, 4, 4), precision=3)
    assert torch.equal(hc.to_ncdhw(out.cpu()), F.conv3d(x4, w, b, padding=1))
    xa, xb = rnd(1, 32, 9, 2, 3, seed=11), rnd(1, 32, 9, 2, 3, seed=12)
    xc = torch.cat([xa, xb], 1)
    r = rnd(1, 128, 9, 2, 3, seed=18)
    ref = F.conv3d(up(F.silu(F.group_norm(xc, 32, gamma, beta, 1e-5))), w2, b2, padding=1) + up(r)
    A, B = _gn_affine(hc, [xa, xb], gamma, beta)
    out, stats, _ = hc.conv3d([hc.to_ndhwc(xa).cuda(), hc.to_ndhwc(xb).cuda()], w2.cuda(), b2.cuda(), (9, 4, 6),
                              in_mode=H.IN_UP, aff=(A, B), act=H.ACT_SILU, res=hc.to_ndhwc(r).cuda(),
                              res_mode=H.RES_UP, precision=3)
    assert rel_err(hc.to_ncdhw(out.cpu()).numpy(), ref.numpy()) < TOL
    check_stats(stats, ref)
    # shapes / modes the form does not cover are refused, not silently mis-computed
    lib = H.load()
    assert lib.ddpm3d_packed_weight_bytes(96, 32, 3, 3) == 0
    with pytest.raises(RuntimeError, match="Winograd"):
        hc.conv3d([hc.to_ndhwc(rnd(1, 16, 4, 8, 8)).cuda()], w.cuda(), b.cuda(), (4, 4, 4), in_mode=H.IN_POOL,
                  precision=3)


def test_conv3d_f16_mode_is_half_precision_grade(hc):
    """precision 2 (one f16 MFMA per product, the --use_fp16 analogue): operands carry 11 bits,
    so the bar is ~1e-3 of the output range, far above the fp32 modes and far below 'wrong'."""
    x = rnd(1, 64, 4, 16, 16, seed=21)
    w = rnd(128, 64, 3, 3, 3, seed=22, scale=0.03)
    b = rnd(128, seed=23)
    ref = F.conv3d(x, w, b, padding=1)
    out, stats, _ = hc.conv3d([hc.to_ndhwc(x).cuda()], w.cuda(), b.cuda(), (4, 16, 16), precision=2)
    e = rel_err(hc.to_ncdhw(out.cpu()).numpy(), ref.numpy())
    assert 1e-6 < e < 2e-3, e
    # integer data is exact in f16 too: same mapping as the other modes
    g = np.random.default_rng(5)
    xi = torch.from_numpy(g.integers(-3, 4, (1, 16, 3, 9, 10)).astype(np.float32))
    wi = torch.from_numpy(g.integers(-2, 3, (40, 16, 3, 3, 3)).astype(np.float32))
    bi = torch.from_numpy(g.integers(-5, 6, (40,)).astype(np.float32))
    out, _, _ = hc.conv3d([hc.to_ndhwc(xi).cuda()], wi.cuda(), bi.cuda(), (3, 9, 10), precision=2)
    assert torch.equal(hc.to_ncdhw(out.cpu()), F.conv3d(xi, wi, bi, padding=1))


@pytest.mark.parametrize("N,D,Hh,W,ci,co", [
    (1, 4, 16, 16, 64, 128),
    (2, 5, 8, 24, 32, 256),      # odd D, batch 2, two cout blocks
    (1, 64, 8, 8, 256, 384),     # split-K + reduce kernel
    (1, 64, 4, 4, 256, 128),     # 4x4x8 tiles + split-K
    (2, 7, 5, 4, 32, 128),       # 4x4x8 tiles, ragged
])
def test_conv3d_f16_winograd_depth_form(hc, N, D, Hh, W, ci, co):
    """precision 4: the f16 mode (one MFMA per product) on the Winograd-D form.  The transformed
    operands are rounded to f16 (weights (g0 +- g1 + g2)/2 at pack time, inputs d_a +- d_b while
    staged), so the bar is the f16 mode's: ~1e-3 of the output range."""
    x = rnd(N, ci, D, Hh, W, seed=31)
    w = rnd(co, ci, 3, 3, 3, seed=32, scale=0.03)
    b = rnd(co, seed=33)
    ref = F.conv3d(x, w, b, padding=1)
    out, stats, _ = hc.conv3d([hc.to_ndhwc(x).cuda()], w.cuda(), b.cuda(), (D, Hh, W), precision=4)
    e = rel_err(hc.to_ncdhw(out.cpu()).numpy(), ref.numpy())
    assert 1e-6 < e < 2e-3, e
    # statistics describe the tensor that was written (not the exact conv): compare to it
    check_stats(stats, hc.to_ncdhw(out.cpu()))
    # small even integers: every transformed operand ((g0 +- g1 + g2)/2, d_a +- d_b) is an integer
    # that f16 holds exactly, so the result is exact
    g = np.random.default_rng(6)
    xi = torch.from_numpy(g.integers(-3, 4, (1, 16, 4, 9, 10)).astype(np.float32))
    wi = torch.from_numpy((2 * g.integers(-2, 3, (128, 16, 3, 3, 3))).astype(np.float32))
    bi = torch.from_numpy(g.integers(-5, 6, (128,)).astype(np.float32))
    out, _, _ = hc.conv3d([hc.to_ndhwc(xi).cuda()], wi.cuda(), bi.cuda(), (4, 9, 10), precision=4)
    assert torch.equal(hc.to_ncdhw(out.cpu()), F.conv3d(xi, wi, bi, padding=1))


@pytest.mark.parametrize("precision,N,D,Hh,W,ci,co", [
    (5, 1, 4, 16, 16, 64, 128),       # direct form
    (5, 1, 9, 4, 4, 32, 96),          # 4x4 tiles, partial cout tile
    (5, 1, 64, 4, 4, 512, 384),       # split-K + reduce kernel
    (6, 1, 4, 16, 16, 64, 128),       # Winograd-D form
    (6, 2, 5, 8, 24, 32, 256),        # odd D, batch 2, two cout blocks
    (6, 1, 64, 8, 8, 256, 384),       # Winograd-D + split-K
    (6, 1, 64, 4, 4, 512, 384),       # Winograd-D on 4x4x8 tiles + split-K
    (6, 2, 9, 4, 7, 32, 128),         # 4x4x8 tiles, ragged
])
def test_conv3d_bf16_mode(hc, precision, N, D, Hh, W, ci, co):
    """precisions 5 / 6: one bf16 MFMA per product on bf16-rounded operands (8 significant bits), fp32
    accumulate -- BASELINE config 4's arithmetic.  Bar: ~2^-9 per operand over a random-sign sum, a
    few 1e-3 of the output range (f16 mode: ~3e-4); far below 'wrong' (a dropped tap is > 1e-1).
    No scaling and no in_bound in this mode: bf16 has fp32's exponent range."""
    x = rnd(N, ci, D, Hh, W, seed=41) * 300.0           # far outside f16's comfortable range
    w = rnd(co, ci, 3, 3, 3, seed=42, scale=0.03)
    b = rnd(co, seed=43)
    ref = F.conv3d(x, w, b, padding=1)
    bound = torch.full((N, 1), float("nan"), device="cuda")     # ignored by the bf16 modes
    out, stats, _ = hc.conv3d([hc.to_ndhwc(x).cuda()], w.cuda(), b.cuda(), (D, Hh, W), precision=precision,
                              bound=bound)
    e = rel_err(hc.to_ncdhw(out.cpu()).numpy(), ref.numpy())
    assert 1e-5 < e < 1e-2, e
    check_stats(stats, hc.to_ncdhw(out.cpu()))
    # small integers are exact in bf16 too (|v| <= 256): same mapping as every other mode
    g = np.random.default_rng(7)
    xi = torch.from_numpy(g.integers(-3, 4, (1, 16, 4, 9, 10)).astype(np.float32))
    wi = torch.from_numpy((2 * g.integers(-2, 3, (128, 16, 3, 3, 3))).astype(np.float32))
    bi = torch.from_numpy(g.integers(-5, 6, (128,)).astype(np.float32))
    out, _, _ = hc.conv3d([hc.to_ndhwc(xi).cuda()], wi.cuda(), bi.cuda(), (4, 9, 10), precision=precision,
                          bound=bound)
    assert torch.equal(hc.to_ncdhw(out.cpu()), F.conv3d(xi, wi, bi, padding=1))


@pytest.mark.parametrize("N,D,Hh,W,ci,co,k", [
    (1, 4, 16, 16, 64, 128, 3),     # direct form, 8x8 tiles
    (1, 8, 4, 4, 256, 128, 3),      # 4x4 tiles, split over Cin (slabs + reduce kernel)
    (2, 3, 8, 8, 96, 64, 1),        # 1x1 conv
    (1, 6, 16, 16, 64, 2, 3),       # the last-layer kernel (conv3d_skinny.hip)
])
def test_conv3d_bf16_mode_vs_bf16_rounded_operands(hc, N, D, Hh, W, ci, co, k):
    """the bf16 mode has no counterpart in the reference (fp16 only), so it is PARITY
    UNPINNED against it; what CAN be pinned is the arithmetic it claims -- the fp32-activated input and
    the weights each rounded ONCE to bf16, exact products, fp32 accumulation.  Every direct-form path
    (8x8 tiles, 4x4 tiles with split-K, 1x1, the last-layer kernel) against torch's fp32 conv on those
    rounded operands, at the fp32 bar per output channel.  (The Winograd-D form rounds the TRANSFORMED
    operands instead and is held to the half-precision bar in test_conv3d_bf16_mode.)"""
    import guided_diffusion._hip as H
    x = rnd(N, ci, D, Hh, W, seed=61) * 3.0
    w = rnd(co, ci, k, k, k, seed=62, scale=0.05)
    b = rnd(co, seed=63)
    A = 1.0 + 0.1 * rnd(N, ci, seed=64)
    B = 0.1 * rnd(N, ci, seed=65)
    xa = F.silu(x * A[:, :, None, None, None] + B[:, :, None, None, None])
    ref = F.conv3d(xa.bfloat16().float().double(), w.bfloat16().float().double(), b.double(), padding=k // 2).float()
    kw = dict(out_layout=H.OUT_NCDHW, want_stats=False) if co <= 2 else {}
    out, _, _ = hc.conv3d([hc.to_ndhwc(x).cuda()], w.cuda(), b.cuda(), (D, Hh, W), aff=(A.cuda(), B.cuda()),
                          act=H.ACT_SILU, precision=5, **kw)
    got = out.cpu() if co <= 2 else hc.to_ncdhw(out.cpu())
    # (the kernel's SiLU uses v_exp / v_rcp, ~2 ulp: a value that lands on the other side of a bf16
    # rounding boundary moves one product by 2^-8 -- a handful of such flips per output)
    assert rel_err_per_channel(got.numpy(), ref.numpy()) < 2e-4


@pytest.mark.parametrize("precision", [0, 1, 2, 3, 4])
def test_conv3d_f16_tensors(hc, precision):
    """ddpm3d_conv_desc.io_dtype with DDPM3D_IO_HALF_IS_F16: sources (virtual concat of an f16 and an fp32
    tensor), residual and output stored as IEEE f16 -- the storage of the reference's --use_fp16 torso
    (unet.py:1035, fp16_util.py:15-22), VERDICT r02 #9.  The kernels must read exactly the f16 values
    (checked against torch on the same rounded tensors) and round the result to nearest-even f16;
    statistics are taken before that rounding.  Every path: direct / Winograd-D forms, ragged tiles
    (general epilogue), split-K reduce, the last-layer kernel reading an f16 tensor."""
    import guided_diffusion._hip as H
    N, D, Hh, W = 2, 5, 16, 16
    x0 = rnd(N, 32, D, Hh, W, seed=51).half()
    x1 = rnd(N, 32, D, Hh, W, seed=52)
    res = rnd(N, 128, D, Hh, W, seed=53).half()
    w = rnd(128, 64, 3, 3, 3, seed=54, scale=0.05)
    b = rnd(128, seed=55)
    A = 1.0 + 0.1 * rnd(N, 64, seed=56)
    B = 0.1 * rnd(N, 64, seed=57)
    xin = torch.cat([x0.float(), x1], 1)
    xin = F.silu(xin * A[:, :, None, None, None] + B[:, :, None, None, None])
    ref = F.conv3d(xin, w, b, padding=1) + res.float()
    out, stats, _ = hc.conv3d([hc.to_ndhwc(x0).cuda(), hc.to_ndhwc(x1).cuda()], w.cuda(), b.cuda(), (D, Hh, W),
                              aff=(A.cuda(), B.cuda()), act=H.ACT_SILU, res=hc.to_ndhwc(res).cuda(),
                              res_mode=H.RES_SAME, precision=precision, out_f16=True)
    assert out.dtype == torch.float16
    got = hc.to_ncdhw(out.cpu().float())
    tol = 2e-5 if precision in (0, 1, 3) else 2e-3
    assert rel_err(got.numpy(), ref.numpy()) < tol + 2.0 ** -11      # half an f16 ulp of the stored value
    if precision in (0, 1, 3):
        assert rel_err(got.numpy(), ref.half().float().numpy()) < 2.0 ** -10 * 1.01   # at most one f16 ulp apart
        check_stats(stats, ref)
    if precision in (1, 2):
        # ragged extents (general epilogue) and a 4x4-tile split-K level, f16 in and out
        for (n2, d2, h2, w2, ci2, co2) in ((1, 3, 9, 11, 32, 64), (1, 8, 4, 4, 256, 128)):
            xs = rnd(n2, ci2, d2, h2, w2, seed=71).half()
            ws = rnd(co2, ci2, 3, 3, 3, seed=72, scale=0.05)
            bs = rnd(co2, seed=73)
            r2 = F.conv3d(xs.float(), ws, bs, padding=1)
            o2, _, _ = hc.conv3d([hc.to_ndhwc(xs).cuda()], ws.cuda(), bs.cuda(), (d2, h2, w2), precision=precision,
                                 out_f16=True)
            assert rel_err(hc.to_ncdhw(o2.cpu().float()).numpy(), r2.numpy()) < (2e-5 if precision == 1 else 2e-3) + 2.0 ** -11
        # the last-layer kernel (Cout = 2, NCDHW fp32 output) reading the f16 residual stream
        xs = rnd(1, 64, 6, 16, 16, seed=81).half()
        ws = rnd(2, 64, 3, 3, 3, seed=82, scale=0.05)
        bs = rnd(2, seed=83)
        A2, B2 = 1.0 + 0.1 * rnd(1, 64, seed=84), 0.1 * rnd(1, 64, seed=85)
        r2 = F.conv3d(F.silu(xs.float() * A2[:, :, None, None, None] + B2[:, :, None, None, None]), ws, bs, padding=1)
        o2, _, _ = hc.conv3d([hc.to_ndhwc(xs).cuda()], ws.cuda(), bs.cuda(), (6, 16, 16), aff=(A2.cuda(), B2.cuda()),
                             act=H.ACT_SILU, precision=precision, out_layout=H.OUT_NCDHW, want_stats=False)
        assert rel_err_per_channel(o2.cpu().numpy(), r2.numpy()) < (2e-5 if precision == 1 else 3e-3)


@pytest.mark.parametrize("precision", [0, 3, 5, 6])
def test_conv3d_bf16_tensors(hc, precision):
    """ddpm3d_conv_desc.io_dtype: sources (virtual concat of a bf16 and an fp32 tensor), residual and
    output stored in bf16 -- the bf16 mode's residual stream.  The kernel must read exactly the
    bf16 values (checked against torch on the same rounded tensors) and round its result to
    nearest-even bf16; statistics are taken BEFORE that rounding (fp32)."""
    import guided_diffusion._hip as H
    N, D, Hh, W = 2, 5, 16, 16
    x0 = rnd(N, 32, D, Hh, W, seed=51).bfloat16()
    x1 = rnd(N, 32, D, Hh, W, seed=52)
    res = rnd(N, 128, D, Hh, W, seed=53).bfloat16()
    w = rnd(128, 64, 3, 3, 3, seed=54, scale=0.05)
    b = rnd(128, seed=55)
    A = 1.0 + 0.1 * rnd(N, 64, seed=56)
    B = 0.1 * rnd(N, 64, seed=57)
    xin = torch.cat([x0.float(), x1], 1)
    xin = F.silu(xin * A[:, :, None, None, None] + B[:, :, None, None, None])
    ref = F.conv3d(xin, w, b, padding=1) + res.float()
    out, stats, _ = hc.conv3d([hc.to_ndhwc(x0).cuda(), hc.to_ndhwc(x1).cuda()], w.cuda(), b.cuda(), (D, Hh, W),
                              aff=(A.cuda(), B.cuda()), act=H.ACT_SILU, res=hc.to_ndhwc(res).cuda(),
                              res_mode=H.RES_SAME, precision=precision, out_bf16=True)
    assert out.dtype == torch.bfloat16
    got = hc.to_ncdhw(out.cpu().float())
    tol = 2e-5 if precision in (0, 3) else 1e-2
    # the stored value is the fp32 result rounded to bf16: half a bf16 ulp = 2^-9 relative
    assert rel_err(got.numpy(), ref.numpy()) < tol + 2.0 ** -8
    if precision in (0, 3):
        assert rel_err(got.numpy(), ref.bfloat16().float().numpy()) < 2.0 ** -7 * 1.01   # at most one bf16 ulp apart
        check_stats(stats, ref)


def test_conv3d_k1_concat(hc):
    xa, xb = rnd(2, 32, 3, 8, 8, seed=1), rnd(2, 16, 3, 8, 8, seed=2)
    w = rnd(64, 48, 1, 1, 1, seed=3, scale=0.1)
    b = rnd(64, seed=4)
    ref = F.conv3d(torch.cat([xa, xb], 1), w, b)
    out, stats, _ = hc.conv3d([hc.to_ndhwc(xa).cuda(), hc.to_ndhwc(xb).cuda()], w.cuda(), b.cuda(), (3, 8, 8))
    assert rel_err(hc.to_ncdhw(out.cpu()).numpy(), ref.numpy()) < TOL
    check_stats(stats, ref)


def test_conv3d_planar_first_layer(hc):
    x, lr = rnd(2, 1, 4, 10, 12, seed=1), rnd(2, 1, 4, 10, 12, seed=2)
    w = rnd(32, 2, 3, 3, 3, seed=3, scale=0.2)
    b = rnd(32, seed=4)
    ref = F.conv3d(torch.cat([x, lr], 1), w, b, padding=1)
    out, _, _ = hc.conv3d([x.cuda(), lr.cuda()], w.cuda(), b.cuda(), (4, 10, 12), planar=True)
    assert rel_err(hc.to_ncdhw(out.cpu()).numpy(), ref.numpy()) < TOL


def test_conv3d_ncdhw_store_small_cout(hc):
    x = rnd(1, 32, 4, 8, 8, seed=1)
    w = rnd(2, 32, 3, 3, 3, seed=2, scale=0.05)
    b = rnd(2, seed=3)
    ref = F.conv3d(x, w, b, padding=1)
    import guided_diffusion._hip as H
    out, _, _ = hc.conv3d([hc.to_ndhwc(x).cuda()], w.cuda(), b.cuda(), (4, 8, 8), out_layout=H.OUT_NCDHW)
    assert rel_err(out.cpu().numpy(), ref.numpy()) < TOL


def _gn_affine(hc, x_list, gamma, beta, film=None):
    """A, B from the stand-alone statistics kernel + finalize."""
    stats = [hc.gn_stats(hc.to_ndhwc(x).cuda()) for x in x_list]
    vox = x_list[0][0, 0].numel()
    if film is not None:
        return hc.gn_finalize(stats, vox, gamma.cuda(), beta.cuda(), film.cuda(), film.shape[1], 0)
    return hc.gn_finalize(stats, vox, gamma.cuda(), beta.cuda())


@pytest.mark.parametrize("C0,C1", [(32, 0), (64, 64), (96, 0), (32, 32)])
def test_groupnorm_silu_prologue(hc, C0, C1):
    """GN32 (+concat) + SiLU fused into a 1x1x1 identity conv == F.silu(F.group_norm(cat))."""
    import guided_diffusion._hip as H
    xs = [rnd(2, C0, 3, 8, 8, seed=1) * 2 + 0.5]
    if C1:
        xs.append(rnd(2, C1, 3, 8, 8, seed=2) * 0.7 - 1.0)
    Cn = C0 + C1
    gamma, beta = 1 + 0.1 * rnd(Cn, seed=3), 0.1 * rnd(Cn, seed=4)
    ref = F.silu(F.group_norm(torch.cat(xs, 1), 32, gamma, beta, 1e-5))
    A, B = _gn_affine(hc, xs, gamma, beta)
    w = torch.eye(Cn).reshape(Cn, Cn, 1, 1, 1).contiguous()
    out, _, _ = hc.conv3d([hc.to_ndhwc(x).cuda() for x in xs], w.cuda(), torch.zeros(Cn).cuda(), (3, 8, 8),
                          aff=(A, B), act=H.ACT_SILU)
    assert rel_err(hc.to_ncdhw(out.cpu()).numpy(), ref.numpy()) < 1e-5


def test_groupnorm_film(hc):
    """out_norm(h) * (1 + scale) + shift then SiLU (unet.py:248-252)."""
    import guided_diffusion._hip as H
    x = rnd(2, 64, 2, 8, 8, seed=1)
    gamma, beta = 1 + 0.1 * rnd(64, seed=3), 0.1 * rnd(64, seed=4)
    film = rnd(2, 128, seed=5, scale=0.5)
    scale, shift = film[:, :64, None, None, None], film[:, 64:, None, None, None]
    ref = F.silu(F.group_norm(x, 32, gamma, beta, 1e-5) * (1 + scale) + shift)
    A, B = _gn_affine(hc, [x], gamma, beta, film)
    w = torch.eye(64).reshape(64, 64, 1, 1, 1).contiguous()
    out, _, _ = hc.conv3d([hc.to_ndhwc(x).cuda()], w.cuda(), torch.zeros(64).cuda(), (2, 8, 8),
                          aff=(A, B), act=H.ACT_SILU)
    assert rel_err(hc.to_ncdhw(out.cpu()).numpy(), ref.numpy()) < 1e-5


@pytest.mark.parametrize("ratio", [30.0, 300.0])
def test_groupnorm_with_large_mean(hc, ratio):
    """un-normalised PET counts (scripts/test.py:201-203 feeds them raw) put tensors
    whose |mean| is tens to hundreds of standard deviations in front of a GroupNorm.  With fp32 partial
    sums E[x^2] - mean^2 loses ratio^2 x 1e-7 of the variance to cancellation (1e-2 at ratio 300); the
    partial sums are fp64 (conv3d_params.h: struct GnAcc -- pivoted fp32 accumulation, one fp64 conversion).  Both producers of statistics -- the
    stand-alone pass and a conv epilogue -- then finalize -> the next conv's prologue, against
    group_norm evaluated in fp64 (nn.py:93-100), at the 1e-4 bar on the NORMALISED output's scale."""
    import guided_diffusion._hip as H
    C = 64
    gamma, beta = 1 + 0.1 * rnd(C, seed=3), 0.1 * rnd(C, seed=4)
    # (a) stand-alone statistics: per-group offsets of +-ratio standard deviations
    x = rnd(2, C, 3, 8, 8, seed=1)
    off = (ratio * torch.sign(rnd(2, 32, seed=7)))[:, :, None].expand(2, 32, C // 32).reshape(2, C)
    x = x + off[:, :, None, None, None]
    ref = F.silu(F.group_norm(x.double(), 32, gamma.double(), beta.double(), 1e-5)).float()
    A, B = _gn_affine(hc, [x], gamma, beta)
    w = torch.eye(C).reshape(C, C, 1, 1, 1).contiguous()
    out, _, _ = hc.conv3d([hc.to_ndhwc(x).cuda()], w.cuda(), torch.zeros(C).cuda(), (3, 8, 8),
                          aff=(A, B), act=H.ACT_SILU)
    assert rel_err(hc.to_ncdhw(out.cpu()).numpy(), ref.numpy()) < 1e-4
    # (b) statistics from a conv epilogue whose output carries the offset (a large bias), in the exact
    # and the default arithmetic, direct and Winograd-D kernels
    for co, precision in ((64, 0), (64, 1), (128, 3)):
        xi = rnd(1, 32, 6, 16, 16, seed=11)
        wc = rnd(co, 32, 3, 3, 3, seed=12, scale=0.05)
        y0 = F.conv3d(xi, wc, None, padding=1)
        g = y0.reshape(1, 32, -1)
        bias = (ratio * g.std(dim=2)[0] * torch.sign(rnd(32, seed=13)))[:, None].expand(32, co // 32).reshape(co)
        gm, bt = 1 + 0.1 * rnd(co, seed=14), 0.1 * rnd(co, seed=15)
        out, stats, _ = hc.conv3d([hc.to_ndhwc(xi).cuda()], wc.cuda(), bias.cuda(), (6, 16, 16), precision=precision)
        y = hc.to_ncdhw(out.cpu())
        refn = F.group_norm(y.double(), 32, gm.double(), bt.double(), 1e-5)
        A, B = hc.gn_finalize([stats], 6 * 16 * 16, gm.cuda(), bt.cuda())
        got = y.double() * A.cpu().double()[:, :, None, None, None] + B.cpu().double()[:, :, None, None, None]
        assert rel_err(got.numpy(), refn.numpy()) < 1e-4, (ratio, precision)


def test_conv_stats_feed_groupnorm(hc):
    """Statistics from a conv epilogue, folded, reproduce group_norm of its output."""
    x = rnd(1, 32, 5, 12, 12, seed=1)
    w = rnd(64, 32, 3, 3, 3, seed=2, scale=0.05)
    b = rnd(64, seed=3)
    y = F.conv3d(x, w, b, padding=1)
    gamma, beta = 1 + 0.1 * rnd(64, seed=4), 0.1 * rnd(64, seed=5)
    ref = F.group_norm(y, 32, gamma, beta, 1e-5)
    out, stats, _ = hc.conv3d([hc.to_ndhwc(x).cuda()], w.cuda(), b.cuda(), (5, 12, 12))
    A, B = hc.gn_finalize([stats], 5 * 12 * 12, gamma.cuda(), beta.cuda())
    got = hc.to_ncdhw(out.cpu()) * A.cpu()[:, :, None, None, None] + B.cpu()[:, :, None, None, None]
    assert rel_err(got.numpy(), ref.numpy()) < 1e-5


def test_conv3d_downsample_block_paths(hc):
    """conv(avgpool(SiLU(GN(x)))) + avgpool(x): the `down` ResBlock's two pooled paths."""
    import guided_diffusion._hip as H
    x = rnd(1, 32, 3, 16, 12, seed=1)
    gamma, beta = 1 + 0.1 * rnd(32, seed=3), 0.1 * rnd(32, seed=4)
    w = rnd(32, 32, 3, 3, 3, seed=2, scale=0.05)
    b = rnd(32, seed=5)
    pool = lambda t: F.avg_pool3d(t, (1, 2, 2), (1, 2, 2))
    ref = F.conv3d(pool(F.silu(F.group_norm(x, 32, gamma, beta, 1e-5))), w, b, padding=1) + pool(x)
    xd = hc.to_ndhwc(x).cuda()
    A, B = _gn_affine(hc, [x], gamma, beta)
    out, stats, _ = hc.conv3d([xd], w.cuda(), b.cuda(), (3, 8, 6), in_mode=H.IN_POOL, aff=(A, B),
                              act=H.ACT_SILU, res=xd, res_mode=H.RES_POOL)
    assert rel_err(hc.to_ncdhw(out.cpu()).numpy(), ref.numpy()) < TOL
    check_stats(stats, ref)


def test_conv3d_splitk_with_pooled_paths(hc):
    """Split-K path with the prologue pool, the pooled residual and statistics from the reduce kernel."""
    import guided_diffusion._hip as H
    lib = H.load()
    assert lib.ddpm3d_conv_workspace_bytes(1, 12, 4, 4, 128, 128, 3, H.PREC_F16X3) > 0   # this shape does split
    x = rnd(1, 128, 12, 8, 8, seed=1)
    gamma, beta = 1 + 0.1 * rnd(128, seed=3), 0.1 * rnd(128, seed=4)
    w = rnd(128, 128, 3, 3, 3, seed=2, scale=0.03)
    b = rnd(128, seed=5)
    pool = lambda t: F.avg_pool3d(t, (1, 2, 2), (1, 2, 2))
    ref = F.conv3d(pool(F.silu(F.group_norm(x, 32, gamma, beta, 1e-5))), w, b, padding=1) + pool(x)
    xd = hc.to_ndhwc(x).cuda()
    A, B = _gn_affine(hc, [x], gamma, beta)
    out, stats, _ = hc.conv3d([xd], w.cuda(), b.cuda(), (12, 4, 4), in_mode=H.IN_POOL, aff=(A, B),
                              act=H.ACT_SILU, res=xd, res_mode=H.RES_POOL)
    assert rel_err(hc.to_ncdhw(out.cpu()).numpy(), ref.numpy()) < TOL
    check_stats(stats, ref)


def test_conv3d_upsample_block_paths(hc):
    import guided_diffusion._hip as H
    x = rnd(1, 32, 3, 4, 6, seed=1)
    gamma, beta = 1 + 0.1 * rnd(32, seed=3), 0.1 * rnd(32, seed=4)
    w = rnd(32, 32, 3, 3, 3, seed=2, scale=0.05)
    b = rnd(32, seed=5)
    up = lambda t: F.interpolate(t, (t.shape[2], t.shape[3] * 2, t.shape[4] * 2), mode="nearest")
    ref = F.conv3d(up(F.silu(F.group_norm(x, 32, gamma, beta, 1e-5))), w, b, padding=1) + up(x)
    xd = hc.to_ndhwc(x).cuda()
    A, B = _gn_affine(hc, [x], gamma, beta)
    out, _, _ = hc.conv3d([xd], w.cuda(), b.cuda(), (3, 8, 12), in_mode=H.IN_UP, aff=(A, B),
                          act=H.ACT_SILU, res=xd, res_mode=H.RES_UP)
    assert rel_err(hc.to_ncdhw(out.cpu()).numpy(), ref.numpy()) < TOL


def test_conv3d_residual_in_place(hc):
    """out = conv(h) + out (the 1x1 skip result already sitting in the output buffer)."""
    import ctypes as C
    import guided_diffusion._hip as H
    h = rnd(1, 32, 2, 8, 8, seed=1)
    r = rnd(1, 48, 2, 8, 8, seed=2)
    w = rnd(48, 32, 3, 3, 3, seed=3, scale=0.05)
    b = rnd(48, seed=4)
    ref = F.conv3d(h, w, b, padding=1) + r
    buf = hc.to_ndhwc(r).cuda()
    lib = H.load()
    d = H.ConvDesc()
    d.N, d.D, d.H, d.W, d.Cin, d.Cout, d.ksize, d.in_mode = 1, 2, 8, 8, 32, 48, 3, H.IN_SAME
    hd = hc.to_ndhwc(h).cuda()
    wp, bd = hc.pack(w.cuda()), b.cuda()
    d.src0, d.C0, d.w_packed, d.bias = H.ptr(hd), 32, H.ptr(wp), H.ptr(bd)
    d.res_mode, d.res, d.out = H.RES_SAME, H.ptr(buf), H.ptr(buf)
    H.check(lib.ddpm3d_conv3d(C.byref(d), H.stream()))
    torch.cuda.synchronize()
    assert rel_err(hc.to_ncdhw(buf.cpu()).numpy(), ref.numpy()) < TOL


def test_conv3d_rejects_bad_descriptors(hc):
    import ctypes as C
    import guided_diffusion._hip as H
    lib = H.load()
    d = H.ConvDesc()
    assert lib.ddpm3d_conv3d(C.byref(d), 0) == -1          # all-zero descriptor
    assert b"shape" in lib.ddpm3d_last_error()
    x = torch.zeros(1, 2, 8, 8, 24, device="cuda")
    with pytest.raises(RuntimeError, match="multiple of 16"):
        hc.conv3d([x], torch.zeros(32, 24, 3, 3, 3, device="cuda"), torch.zeros(32, device="cuda"), (2, 8, 8))


def test_linear_and_timestep_embedding(hc):
    import guided_diffusion._hip as H
    lib = H.load()
    t = torch.tensor([0.0, 4.0, 499.0, 999.0, 123.0])
    from oracle import unet_ref
    ref = unet_ref.timestep_embedding(t, 128)
    import math
    freqs = torch.exp(-math.log(10000) * torch.arange(0, 64, dtype=torch.float32) / 64).cuda()
    td, out = t.cuda(), torch.empty(5, 128, device="cuda")
    H.check(lib.ddpm3d_timestep_embedding(H.ptr(td), 5, 128, H.ptr(freqs), H.ptr(out), H.stream()))
    assert torch.allclose(out.cpu(), ref, atol=2e-7, rtol=0)
    for rows, K, O, silu in [(5, 128, 512, 0), (11, 512, 70, 1), (1, 96, 3, 1)]:
        x, w, b = rnd(rows, K, seed=1), rnd(O, K, seed=2, scale=0.05), rnd(O, seed=3)
        ref = F.linear(F.silu(x) if silu else x, w, b)
        o = torch.empty(rows, O, device="cuda")
        xd, wd, bd = x.cuda(), w.cuda(), b.cuda()   # keep the device buffers alive across the launch
        H.check(lib.ddpm3d_linear(H.ptr(xd), rows, K, H.ptr(wd), H.ptr(bd), O, silu, H.ptr(o), O, H.stream()))
        assert rel_err(o.cpu().numpy(), ref.numpy()) < 1e-5


@pytest.mark.parametrize("precision", [0, 1])
@pytest.mark.parametrize("N,T,heads,ch,scale", [(1, 128, 2, 32, 1.0), (2, 200, 3, 64, 1.0), (1, 77, 1, 32, 1.0),
                                                (1, 512, 2, 64, 1.0), (1, 300, 2, 128, 1.0),
                                                (1, 256, 1, 64, 3.0),      # peaked softmax (|scores| ~ 70)
                                                (1, 160, 2, 32, 0.02)])    # tiny activations (lo parts small)
def test_attention_core_vs_legacy_reference(hc, N, T, heads, ch, scale, precision):
    """Streaming attention vs the materialised softmax of QKVAttentionLegacy (unet.py:337-354),
    including T not a multiple of the 32-key tile or the 128-query block, in both arithmetic
    modes of the two products (0: exact fp32 MFMA, 1: f16x3) -- same bar."""
    import math
    import guided_diffusion._hip as H
    lib = H.load()
    qkv = rnd(N, heads * 3 * ch, T, seed=1, scale=scale)         # reference layout (N, H*3*C, T)
    q, k, v = qkv.reshape(N * heads, ch * 3, T).split(ch, dim=1)
    s = 1 / math.sqrt(math.sqrt(ch))
    w = torch.softmax(torch.einsum("bct,bcs->bts", q * s, k * s).float(), dim=-1)
    ref = torch.einsum("bts,bcs->bct", w, v).reshape(N, -1, T)    # (N, H*C, T)
    qd = qkv.permute(0, 2, 1).contiguous().cuda()                 # channels-last (N, T, H*3*C)
    out = torch.full((N, T, heads * ch), float("nan"), device="cuda")
    qb = qd.abs().reshape(N, -1).amax(dim=1).contiguous()          # per-sample range of q, k, v
    H.check(lib.ddpm3d_attention_p(H.ptr(qd), N, T, heads, ch, precision, H.ptr(qb), 1, 1, H.ptr(out), H.stream()))
    torch.cuda.synchronize()
    assert rel_err(out.permute(0, 2, 1).cpu().numpy(), ref.double().numpy()) < 1e-5
    if precision == 0:      # the two-argument entry point is the exact mode
        out0 = torch.full_like(out, float("nan"))
        H.check(lib.ddpm3d_attention(H.ptr(qd), N, T, heads, ch, H.ptr(out0), H.stream()))
        torch.cuda.synchronize()
        assert torch.equal(out0, out)


@pytest.mark.parametrize("precision", [0, 1])
@pytest.mark.parametrize("T,heads", [(4096, 3), (32768, 1)])
def test_attention_core_long_sequences_vs_blocked_cpu_softmax(hc, T, heads, precision):
    """The streaming kernel at the sequence lengths of the published network's attention levels -- T = 4 096 and
    BASELINE config 5's T = 32 768 (1 024 key tiles per query block, with the running-max rescale skipped where
    no lane needs it) -- against an INDEPENDENT evaluation: the oracle's query-blocked softmax in fp64 on the CPU
    (oracle/unet_ref.py: qkv_attention, pinned to the reference's attention outputs in test_oracle_golden.py).
    ch = 64 as in the published configuration; both arithmetics of the two products; same 1e-5 bar as at T <= 512."""
    import guided_diffusion._hip as H
    from oracle import unet_ref
    lib = H.load()
    ch, N = 64, 1
    qkv = rnd(N, heads * 3 * ch, T, seed=11)                        # reference layout (N, H*3*C, T)
    q, k, v = qkv.reshape(N * heads, ch * 3, T).split(ch, dim=1)
    ref = unet_ref.qkv_attention(q, k, v, block=1024, dtype=torch.float64).reshape(N, -1, T)
    qd = qkv.permute(0, 2, 1).contiguous().cuda()
    out = torch.full((N, T, heads * ch), float("nan"), device="cuda")
    qb = qd.abs().reshape(N, -1).amax(dim=1).contiguous()
    H.check(lib.ddpm3d_attention_p(H.ptr(qd), N, T, heads, ch, precision, H.ptr(qb), 1, 1, H.ptr(out), H.stream()))
    torch.cuda.synchronize()
    got = out.permute(0, 2, 1).cpu().numpy()
    assert np.isfinite(got).all()
    assert rel_err(got, ref.numpy()) < 1e-5
    # per query position too: at this T the averaged-out values are small against the global maximum
    err = np.abs(got - ref.numpy()).max(axis=1) / np.abs(ref.numpy()).max(axis=1)
    assert err.max() < 1e-4


def test_attention_rejects_unsupported_head_width(hc):
    import guided_diffusion._hip as H
    lib = H.load()
    x = torch.zeros(1, 16, 3 * 8, device="cuda")
    assert lib.ddpm3d_attention(H.ptr(x), 1, 16, 1, 8, H.ptr(x), H.stream()) == -3
    assert b"channels per head" in lib.ddpm3d_last_error()


def test_layout_round_trip(hc):
    import guided_diffusion._hip as H
    lib = H.load()
    x = rnd(2, 37, 3, 5, 7, seed=1)
    xd = x.cuda()
    y = torch.empty(2, 3, 5, 7, 37, device="cuda")
    H.check(lib.ddpm3d_ncdhw_to_ndhwc(H.ptr(xd), 2, 37, 105, H.ptr(y), H.stream()))
    assert torch.equal(y.cpu(), hc.to_ndhwc(x))
    z = torch.empty_like(xd)
    H.check(lib.ddpm3d_ndhwc_to_ncdhw(H.ptr(y), 2, 37, 105, H.ptr(z), H.stream()))
    assert torch.equal(z.cpu(), x)


@pytest.mark.parametrize("learn,xstart,clip", [(True, False, True), (False, False, True),
                                               (True, True, False), (True, False, False)])
def test_sampler_update_kernels(hc, learn, xstart, clip):
    import guided_diffusion._hip as H
    from guided_diffusion import script_util as su
    from oracle import sampler_ref, schedule_ref
    diff = su.create_gaussian_diffusion(steps=1000, learn_sigma=learn, predict_xstart=xstart,
                                        timestep_respacing="10")
    _, tb = schedule_ref.spaced_schedule(1000, "linear", "10")
    N, vox = 3, 1000
    x, z = rnd(N, 1, 10, 10, 10, seed=1), rnd(N, 1, 10, 10, 10, seed=2)
    mo = rnd(N, 2 if learn else 1, 10, 10, 10, seed=3)
    for i in (9, 4, 0):
        t = torch.full((N,), i, dtype=torch.long)
        mean, logvar, x0 = sampler_ref.mean_variance(tb, mo, x, i, learn, xstart, clip)
        ref = mean + (0.0 if i == 0 else 1.0) * torch.exp(0.5 * logvar) * z
        got = diff._update("ddpm", mo.cuda(), x.cuda(), t, z.cuda(), clip)
        assert rel_err(got["sample"].cpu().numpy(), ref.numpy()) < 2e-6
        assert rel_err(got["pred_xstart"].cpu().numpy(), x0.numpy()) < 2e-6
        for eta in (0.0, 0.7):
            got2 = diff._update("ddim", mo.cuda(), x.cuda(), t, z.cuda(), clip, eta)
            # single-step DDIM reference, written out (gaussian_diffusion.py:566-584)
            c = lambda k: float(np.float32(tb[k][i]))
            eps = (c("sqrt_recip_alphas_cumprod") * x - x0) / c("sqrt_recipm1_alphas_cumprod")
            ab, abp = torch.tensor(c("alphas_cumprod")), torch.tensor(c("alphas_cumprod_prev"))
            sig = eta * torch.sqrt((1 - abp) / (1 - ab)) * torch.sqrt(1 - ab / abp)
            ref2 = x0 * torch.sqrt(abp) + torch.sqrt(1 - abp - sig ** 2) * eps + (0.0 if i == 0 else 1.0) * sig * z
            assert rel_err(got2["sample"].cpu().numpy(), ref2.numpy()) < 2e-6


@pytest.mark.parametrize("N,D,Hh,W,ci,co,precision", [
    (1, 4, 16, 16, 32, 32, 0),       # tiny net's widths, one 8x8 output tile
    (2, 5, 24, 12, 32, 48, 1),       # ragged output tiles (12x6), batch 2, partial cout tile
    (1, 9, 8, 8, 64, 160, 1),        # 4x4 output tiles, two cout blocks
    (1, 4, 32, 32, 128, 128, 1),     # published widths
    (1, 4, 32, 32, 128, 128, 5),     # bf16 arithmetic
])
def test_conv3d_strided_downsample(hc, N, D, Hh, W, ci, co, precision):
    """DDPM3D_IN_STRIDE2: Downsample(use_conv=True) = Conv3d(stride=(1,2,2), padding=1) (unet.py:129-133)
    in ONE launch on the half-resolution output grid (r01 ran the full-resolution conv and kept a
    quarter of it), with the GroupNorm statistics of the result from the same epilogue."""
    import guided_diffusion._hip as H
    x = rnd(N, ci, D, Hh, W, seed=71)
    w = rnd(co, ci, 3, 3, 3, seed=72, scale=0.05)
    b = rnd(co, seed=73)
    ref = F.conv3d(x, w, b, stride=(1, 2, 2), padding=1)
    assert ref.shape[-2:] == (Hh // 2, W // 2)
    out, stats, _ = hc.conv3d([hc.to_ndhwc(x).cuda()], w.cuda(), b.cuda(), (D, Hh // 2, W // 2),
                              in_mode=H.IN_STRIDE2, precision=precision)
    got = hc.to_ncdhw(out.cpu())
    assert rel_err(got.numpy(), ref.numpy()) < (1e-2 if precision == 5 else TOL)
    check_stats(stats, got)


def test_strided_downsample_conv_as_conv_plus_subsample(hc):
    """Downsample(use_conv=True) (unet.py:129-133): Conv3d(stride=(1,2,2), padding=1) equals the
    stride-1 conv kept at the even (y, x) -- conv3d + ddpm3d_subsample_hw2, then ddpm3d_gn_stats."""
    from guided_diffusion import _hip as H
    lib = H.load()
    N, D, Hh, W, ci, co = 2, 3, 12, 8, 32, 48
    x = rnd(N, ci, D, Hh, W, seed=1)
    w = rnd(co, ci, 3, 3, 3, seed=2, scale=0.05)
    b = rnd(co, seed=3)
    ref = F.conv3d(x, w, b, stride=(1, 2, 2), padding=1)
    full, _, _ = hc.conv3d([hc.to_ndhwc(x).cuda()], w.cuda(), b.cuda(), (D, Hh, W), precision=1)
    out = torch.full((N, D, Hh // 2, W // 2, co), float("nan"), dtype=torch.float32, device="cuda")
    H.check(lib.ddpm3d_subsample_hw2(H.ptr(full), N, D, Hh, W, co, H.ptr(out), H.stream()))
    vox = D * (Hh // 2) * (W // 2)
    rows = lib.ddpm3d_gn_stats_rows(vox)
    stats = torch.zeros(N, co, rows, 2, dtype=torch.float64, device="cuda")
    H.check(lib.ddpm3d_gn_stats(H.ptr(out), N, vox, co, H.ptr(stats), H.stream()))
    torch.cuda.synchronize()
    assert rel_err(hc.to_ncdhw(out.cpu()).numpy(), ref.numpy()) < TOL
    check_stats(stats, ref)
    # odd extents / unaligned channel counts are refused, not mis-copied
    assert lib.ddpm3d_subsample_hw2(H.ptr(full), N, D, Hh + 1, W, co, H.ptr(out), H.stream()) == -1  # DDPM3D_EINVAL
    assert lib.ddpm3d_subsample_hw2(H.ptr(full), N, D, Hh, W, co + 2, H.ptr(out), H.stream()) == -1


def test_conv3d_launch_orders_agree_bitwise(hc):
    """Both workgroup -> XCD orders (tiles fastest / cout blocks and K splits fastest, chosen per call
    through ddpm3d_conv_desc.kernel_hint) run the SAME arithmetic per output element: outputs and
    GroupNorm partial sums must be bit-identical, in the direct and in the Winograd-D form."""
    import guided_diffusion._hip as H
    cases = [
        (1, 8, 16, 16, 64, 128),     # four z-pairs
        (2, 4, 8, 24, 32, 256),      # batch 2, two cout blocks
        (1, 64, 8, 8, 256, 384),     # split-K + reduce kernel, weights outweigh activations
        (1, 5, 16, 24, 32, 128),     # odd D: three z-pairs, the last half valid
        (1, 1, 9, 12, 16, 128),      # D = 1, ragged H / W: edge tiles take the general epilogue
    ]
    for i, (N, D, Hh, W, ci, co) in enumerate(cases):
        x = hc.to_ndhwc(rnd(N, ci, D, Hh, W, seed=10 + i)).cuda()
        w = rnd(co, ci, 3, 3, 3, seed=20 + i, scale=0.05).cuda()
        b = rnd(co, seed=30 + i).cuda()
        A = (1.0 + 0.1 * rnd(N, ci, seed=40 + i)).cuda()
        B = (0.1 * rnd(N, ci, seed=50 + i)).cuda()
        res = hc.to_ndhwc(rnd(N, co, D, Hh, W, seed=60 + i)).cuda()
        for precision in (1, 3):
            ref_o = ref_s = None
            for hint in (0, H.HINT_WSTAT_ON, H.HINT_WSTAT_OFF):
                out, stats, _ = hc.conv3d([x], w, b, (D, Hh, W), aff=(A, B), act=H.ACT_SILU, res=res,
                                          res_mode=H.RES_SAME, precision=precision, hint=hint)
                o, s_ = out.cpu().numpy(), stats.cpu().numpy()
                assert np.isfinite(o).all()
                if ref_o is None:
                    ref_o, ref_s = o, s_
                else:
                    assert np.array_equal(o, ref_o), (i, precision, hint)
                    assert np.array_equal(s_, ref_s), (i, precision, hint)


@pytest.mark.parametrize("case", [
    # (N, D, H, W, Cin, Cout, ksize, precision, forced split (0 = the rule's), residual)
    (1, 64, 4, 4, 512, 512, 3, 3, 0, "same"),       # the published 4x4 level: 4x4x8 Winograd tiles, 16-way
    (1, 24, 4, 4, 384, 384, 3, 3, 0, "none"),       # three cout blocks
    (1, 8, 8, 8, 256, 256, 3, 3, 0, "same"),        # 8x4x4 tiles (H % 8 == 0, D % 4 == 0)
    (2, 6, 8, 16, 128, 256, 3, 3, 3, "same"),       # 8x8x2 tiles (D % 4 != 0), batch 2, forced 3-way
    (1, 5, 9, 11, 64, 128, 3, 3, 2, "same"),        # ragged tiles: the general slab stores
    (1, 16, 4, 4, 128, 128, 3, 1, 0, "same"),       # direct f16x3 kernel on 4x4 tiles
    (1, 9, 8, 8, 96, 128, 3, 0, 2, "same"),         # exact fp32 mode
    (1, 12, 4, 4, 256, 128, 3, 6, 0, "same"),       # bf16 Winograd form
    (1, 64, 8, 8, 256, 384, 3, 3, 0, "same"),       # 768 workgroups: more than are resident at once
    (1, 16, 4, 4, 1024, 512, 1, 1, 0, "none"),      # the 1x1 skip-connection kernel, f16x3
    (1, 16, 8, 8, 768, 256, 1, 5, 4, "none"),       # ... bf16, forced 4-way
])
def test_conv3d_splitk_slabs_in_every_kernel_family(hc, case):
    """Split-over-Cin launches of every kernel family (r04: their raw slabs leave the f16x3 Winograd kernel through
    the 16-byte epilogue too; forced split factors may now carry statistics, sized by ddpm3d_conv_plan): against
    torch on the CPU, with statistics, bitwise repeatable, no slab element left unwritten (the workspace starts
    out as NaN).  (The in-launch combine of r04 -- last workgroup to arrive sums the slabs -- measured 7-12 % SLOWER
    per forward than this reduce launch and is not in the product: profiles/r04_ab_splitk_in_launch_*.txt.)"""
    import guided_diffusion._hip as H
    N, D, Hh, W, ci, co, k, precision, forced, resm = case
    x = rnd(N, ci, D, Hh, W, seed=3)
    w = rnd(co, ci, k, k, k, seed=4, scale=0.03)
    b = rnd(co, seed=5)
    res = rnd(N, co, D, Hh, W, seed=6)
    kw = dict(precision=precision)
    ref = F.conv3d(x, w, b, padding=k // 2)
    if k == 3:
        A, B = 1.0 + 0.1 * rnd(N, ci, seed=7), 0.1 * rnd(N, ci, seed=8)
        ref = F.conv3d(F.silu(x * A[:, :, None, None, None] + B[:, :, None, None, None]), w, b, padding=1)
        kw.update(aff=(A.cuda(), B.cuda()), act=H.ACT_SILU)
    else:
        kw.update(want_stats=False)
    if resm == "same":
        ref = ref + res
        kw.update(res=hc.to_ndhwc(res).cuda(), res_mode=H.RES_SAME)
    hint = forced << H.HINT_SPLITK_SHIFT
    xd, wd, bd = hc.to_ndhwc(x).cuda(), w.cuda(), b.cuda()
    out, stats, rows = hc.conv3d([xd], wd, bd, (D, Hh, W), hint=hint, **kw)
    assert hc.LAST_PLAN["split"] > 1 and (not forced or hc.LAST_PLAN["split"] == forced)
    o = out.cpu().numpy()
    assert np.isfinite(o).all()
    tol = {0: 2e-5, 1: 2e-5, 3: 2e-5}.get(precision, 4e-3)
    assert rel_err(hc.to_ncdhw(out.cpu()).numpy(), ref.numpy()) < tol
    if stats is not None:
        # the sums are of the values the kernel stored: against the CPU result in the fp32-grade modes, against the
        # kernel's own output in the 16-bit-operand modes
        check_stats(stats, ref if precision in (0, 1, 3) else hc.to_ncdhw(out.cpu().float()))
    for _ in range(3):
        o3, s3, _ = hc.conv3d([xd], wd, bd, (D, Hh, W), hint=hint, **kw)
        assert np.array_equal(o3.cpu().numpy(), o)
        if stats is not None:
            assert torch.equal(s3, stats)


@pytest.mark.parametrize("precision", [1, 2, 5])
@pytest.mark.parametrize("N,D,Hh,W,ci,co,layout", [
    (1, 6, 16, 24, 32, 2, "ncdhw"),     # the network's last layer: GroupNorm + SiLU prologue, NCDHW fp32 output
    (2, 5, 9, 12, 128, 1, "ndhwc"),     # learn_sigma off: one output channel; batch 2, ragged H / W, odd D
    (1, 1, 8, 8, 48, 2, "ncdhw"),       # D = 1 (the 2-D model's last conv shape class)
    (1, 19, 8, 16, 16, 2, "ndhwc"),     # several depth segments with a ragged last one
])
def test_conv3d_skinny_last_layer(hc, precision, N, D, Hh, W, ci, co, layout):
    """3x3x3 convs with one or two output channels run on their own kernel (conv3d_skinny.hip: per input
    voxel a 27*Cout-column GEMM, gathered along the three axes) -- vs F.conv3d, with the prologue
    (affine + SiLU), both output layouts, in the f16x3 / f16 / bf16 arithmetic, bf16 source tensor,
    and bit-exact on small-integer data."""
    import guided_diffusion._hip as H
    x = rnd(N, ci, D, Hh, W, seed=71)
    w = rnd(co, ci, 3, 3, 3, seed=72, scale=0.05)
    b = rnd(co, seed=73)
    A = 1.0 + 0.1 * rnd(N, ci, seed=74)
    B = 0.1 * rnd(N, ci, seed=75)
    xs = hc.to_ndhwc(x)
    if precision == 5:
        xs = xs.to(torch.bfloat16)
        x = hc.to_ncdhw(xs.float())
    ref = F.conv3d(F.silu(x * A[:, :, None, None, None] + B[:, :, None, None, None]), w, b, padding=1)
    out, _, _ = hc.conv3d([xs.cuda()], w.cuda(), b.cuda(), (D, Hh, W), aff=(A.cuda(), B.cuda()), act=H.ACT_SILU,
                          out_layout=H.OUT_NCDHW if layout == "ncdhw" else H.OUT_NDHWC, want_stats=False,
                          precision=precision)
    got = out.cpu() if layout == "ncdhw" else hc.to_ncdhw(out.cpu())
    assert torch.isfinite(got).all()
    e = rel_err_per_channel(got.numpy(), ref.numpy())
    assert e < {1: TOL, 2: 2e-3, 5: 1e-2}[precision], e
    # no prologue, integer data: exact in every arithmetic
    g = np.random.default_rng(9)
    xi = torch.from_numpy(g.integers(-3, 4, (N, ci, D, Hh, W)).astype(np.float32))
    wi = torch.from_numpy((2 * g.integers(-2, 3, (co, ci, 3, 3, 3))).astype(np.float32))
    bi = torch.from_numpy(g.integers(-5, 6, (co,)).astype(np.float32))
    out, _, _ = hc.conv3d([hc.to_ndhwc(xi).cuda()], wi.cuda(), bi.cuda(), (D, Hh, W), want_stats=False,
                          precision=precision)
    assert torch.equal(hc.to_ncdhw(out.cpu()), F.conv3d(xi, wi, bi, padding=1))


@pytest.mark.parametrize("precision", [1, 2, 5])
@pytest.mark.parametrize("case", [
    # N, D, H, W, C0, C1, Cout, residual
    (1, 4, 16, 16, 128, 128, 128, False),    # decoder skip connection: virtual concat, 8x8 tiles
    (2, 3, 9, 11, 64, 0, 128, True),         # ragged extents (general epilogue), two samples, residual
    (1, 16, 4, 4, 512, 512, 384, False),     # 4x4 tiles, split over Cin (slabs + reduce kernel)
    (1, 8, 8, 8, 96, 32, 256, True),         # two cout blocks, 4 blocks of K (3 + 1 across the concat)
    (1, 2, 8, 8, 32, 0, 128, False),         # ONE block of K (the ring's other slots multiply zeros)
])
def test_conv1x1_skip_connection_kernel(hc, precision, case):
    """conv1x1.hip (r03): the ResBlock skip connections (unet.py:173-186) -- 1x1x1 convs on raw inputs -- run
    a register-fed GEMM that walks K in permuted 32-channel blocks.  Against torch's conv on the same
    operands, per output channel: fp32 bar for the split-f16 mode, on bf16- / f16-rounded operands for
    the one-MFMA modes; fp32, bf16 and f16 storage of the sources; small-integer data exactly."""
    import guided_diffusion._hip as H
    N, D, Hh, W, c0, c1, co, with_res = case
    ci = c0 + c1
    x = rnd(N, ci, D, Hh, W, seed=71) * 2.0
    w = rnd(co, ci, 1, 1, 1, seed=72, scale=0.05)
    b = rnd(co, seed=73)
    res = rnd(N, co, D, Hh, W, seed=74) if with_res else None
    for store in (torch.float32, torch.bfloat16, torch.float16):
        if store == torch.bfloat16 and precision != 5 or store == torch.float16 and precision == 5:
            continue          # the engine's plans: bf16 storage with bf16 arithmetic, f16 with the f16 modes
        xs = x.to(store)
        xv = xs.float()
        if precision == 5:
            xo, wo = xv.bfloat16().float(), w.bfloat16().float()
        elif precision == 2:
            # (the kernel rounds the SCALED operands to f16; with power-of-two scales that is the rounding of
            # the operands themselves except in the subnormal range, which these magnitudes do not reach)
            xo, wo = xv, w
        else:
            xo, wo = xv, w
        ref = F.conv3d(xo.double(), wo.double(), b.double()).float()
        if with_res:
            ref = ref + res
        srcs = [hc.to_ndhwc(xs[:, :c0]).cuda()] + ([hc.to_ndhwc(xs[:, c0:]).cuda()] if c1 else [])
        out, _, _ = hc.conv3d(srcs, w.cuda(), b.cuda(), (D, Hh, W), precision=precision, want_stats=False,
                              res=hc.to_ndhwc(res).cuda() if with_res else None,
                              res_mode=H.RES_SAME if with_res else H.RES_NONE)
        got = hc.to_ncdhw(out.cpu())
        assert torch.isfinite(got).all()
        tol = {1: 5e-6, 2: 2e-3, 5: 1e-5}[precision]
        assert rel_err_per_channel(got.numpy(), ref.numpy()) < tol, (store, precision)
    # small integers: every product and sum exact in every mode
    g = np.random.default_rng(75)
    xi = torch.from_numpy(g.integers(-3, 4, (N, ci, D, Hh, W)).astype(np.float32))
    wi = torch.from_numpy((2 * g.integers(-2, 3, (co, ci, 1, 1, 1))).astype(np.float32))
    bi = torch.from_numpy(g.integers(-5, 6, (co,)).astype(np.float32))
    srcs = [hc.to_ndhwc(xi[:, :c0]).cuda()] + ([hc.to_ndhwc(xi[:, c0:]).cuda()] if c1 else [])
    out, _, _ = hc.conv3d(srcs, wi.cuda(), bi.cuda(), (D, Hh, W), precision=precision, want_stats=False)
    assert torch.equal(hc.to_ncdhw(out.cpu()), F.conv3d(xi, wi, bi))


@pytest.mark.parametrize("store", ["f32", "bf16", "f16"])
def test_pool_act_prepass(hc, store):
    """ddpm3d_pool_act (ABI 11): AvgPool3d((1,2,2)) of SiLU(A x + B) -- the down ResBlock's
    h_upd(in_rest(x)) (unet.py:194-195, :238-242) -- against torch, for fp32 / bf16 / f16 sources and
    outputs, exact and fast SiLU; and against the conv's own IN_POOL prologue: an identity 1x1 conv
    on the pooled input must give the SAME tensor bit for bit (exact mode, same window order)."""
    import ctypes as C
    import guided_diffusion._hip as H
    lib = H.load()
    N, D, Hh, W, Cn = 2, 3, 6, 10, 32
    x = rnd(N, Cn, D, 2 * Hh, 2 * W, seed=81) * 2.0
    A = 1.0 + 0.1 * rnd(N, Cn, seed=82)
    B = 0.1 * rnd(N, Cn, seed=83)
    dt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[store]
    xs = hc.to_ndhwc(x).to(dt).cuda()
    xv = hc.to_ncdhw(xs.float().cpu())
    ref = F.avg_pool3d(F.silu(xv * A[:, :, None, None, None] + B[:, :, None, None, None]), (1, 2, 2))
    io_in = 0 if store == "f32" else (H.IO_SRC0_BF16 | (H.IO_HALF_IS_F16 if store == "f16" else 0))
    Ad, Bd = A.cuda(), B.cuda()
    for fast in (0, 1):
        for out_half in ((False,) if store == "f32" else (False, True)):
            io = io_in | ((H.IO_OUT_BF16 | (H.IO_HALF_IS_F16 if store == "f16" else 0)) if out_half else 0)
            out = torch.full((N, D, Hh, W, Cn), float("nan"), dtype=dt if out_half else torch.float32, device="cuda")
            H.check(lib.ddpm3d_pool_act(H.ptr(xs), H.ptr(Ad), H.ptr(Bd), H.ACT_SILU, fast, N, D, Hh, W, Cn, H.ptr(out),
                                        io, H.stream()))
            torch.cuda.synchronize()
            got = hc.to_ncdhw(out.float().cpu())
            tol = (2.0 ** -8 if store == "bf16" else 2.0 ** -10) if out_half else (2e-6 if fast else 5e-7)
            assert rel_err(got.numpy(), ref.numpy()) < tol, (store, fast, out_half)
    # no affine, no activation: the plain average, exactly
    out = torch.empty(N, D, Hh, W, Cn, dtype=torch.float32, device="cuda")
    H.check(lib.ddpm3d_pool_act(H.ptr(xs), 0, 0, H.ACT_NONE, 0, N, D, Hh, W, Cn, H.ptr(out), io_in, H.stream()))
    v = xs.float().cpu()
    plain = (((v[:, :, 0::2, 0::2] + v[:, :, 0::2, 1::2]) + v[:, :, 1::2, 0::2]) + v[:, :, 1::2, 1::2]) * 0.25
    assert torch.equal(out.cpu(), plain)
    # the conv's own pooled prologue evaluates the same numbers: identity 1x1 conv, exact mode
    if store == "f32":
        w = torch.eye(Cn).reshape(Cn, Cn, 1, 1, 1)
        b = torch.zeros(Cn)
        fused, _, _ = hc.conv3d([xs], w.cuda(), b.cuda(), (D, Hh, W), in_mode=H.IN_POOL, aff=(Ad, Bd),
                                act=H.ACT_SILU, precision=0, want_stats=False)
        H.check(lib.ddpm3d_pool_act(H.ptr(xs), H.ptr(Ad), H.ptr(Bd), H.ACT_SILU, 0, N, D, Hh, W, Cn, H.ptr(out), 0,
                                    H.stream()))
        torch.cuda.synchronize()
        assert torch.equal(out.cpu(), fused.cpu())
    # rejected: C not a multiple of 4, an activation without the affine
    assert lib.ddpm3d_pool_act(H.ptr(xs), 0, 0, H.ACT_NONE, 0, N, D, Hh, W, 30, H.ptr(out), io_in, H.stream()) == H.E_INVAL
    assert lib.ddpm3d_pool_act(H.ptr(xs), 0, 0, H.ACT_SILU, 0, N, D, Hh, W, Cn, H.ptr(out), io_in, H.stream()) == H.E_INVAL


def test_absmax_is_exact_and_order_free(hc):
    """ddpm3d_absmax (the range bound of tensors no conv epilogue produced: the network's two input volumes):
    max |x| per sample and tensor, exactly, from 32 workgroups per tensor folded by an atomic max on the bit
    patterns -- including sizes that are not a multiple of the vector width or of the workgroup slices, an
    unaligned base pointer, and a sample whose maximum sits in the last element."""
    import guided_diffusion._hip as H
    lib = H.load()
    for N, per in ((1, 64 ** 3), (3, 40 * 33 * 17), (2, 5), (2, 4099)):
        g = torch.Generator().manual_seed(per)
        a = (torch.randn(N, per, generator=g) * 3).cuda()
        b = (torch.randn(N, per + 1, generator=g) * 0.01).cuda()[:, 1:].contiguous()
        a[-1, -1] = -1234.5
        bound = torch.full((N, 2), float("nan"), device="cuda")
        for _ in range(2):      # a second call must not see the first one's result
            H.check(lib.ddpm3d_absmax(H.ptr(a), H.ptr(b), N, per, H.ptr(bound), H.stream()))
        torch.cuda.synchronize()
        want = torch.stack([a.abs().amax(dim=1), b.abs().amax(dim=1)], dim=1)
        assert torch.equal(bound, want)
        # one tensor, unaligned base (the scalar path)
        flat = torch.randn(N * per + 1, generator=g).cuda()
        u = flat[1:]
        b1 = torch.full((N,), float("nan"), device="cuda")
        H.check(lib.ddpm3d_absmax(H.ptr(u), 0, N, per, H.ptr(b1), H.stream()))
        torch.cuda.synchronize()
        assert torch.equal(b1, u.reshape(N, per).abs().amax(dim=1))


@pytest.mark.parametrize("case", ["bf16 concat 8-byte stores", "f16x3 split slabs", "bf16 1x1 concat", "f16 residual in place"])
def test_conv3d_wide_epilogue_stores_are_repeatable(hc, case):
    """r04: gfx950 reads the first data register of an 8- / 16-byte buffer store late for the last lanes of a row; a
    VALU write scheduled shortly behind the store can overtake it (profiles/r04_store_data_hazard_plain.txt: the lean
    epilogue made three launches of the bf16 network differ from run to run in 25-1850 bytes, 1-2 launches of 13, with
    every accuracy test green).  Every epilogue store now pins its data registers (conv3d_epilogue.h epi_store_*);
    this holds the affected launch shapes to bitwise repeatability over 40 launches each, output and statistics."""
    import guided_diffusion._hip as H
    if case == "bf16 concat 8-byte stores":         # the decoder's 256 -> 128 conv at the top level of the 8x32x32 network
        srcs = [hc.to_ndhwc(rnd(1, 128, 8, 32, 32, seed=71)).bfloat16().cuda(), hc.to_ndhwc(rnd(1, 128, 8, 32, 32, seed=72)).bfloat16().cuda()]
        w, kw, dhw = rnd(128, 256, 3, 3, 3, seed=73, scale=0.03), dict(precision=6, out_bf16=True), (8, 32, 32)
    elif case == "f16x3 split slabs":               # 4x4x8 tiles, 16-way split: 16-byte slab stores + reduce
        srcs = [hc.to_ndhwc(rnd(1, 512, 16, 4, 4, seed=74)).cuda()]
        w, kw, dhw = rnd(512, 512, 3, 3, 3, seed=75, scale=0.02), dict(precision=3), (16, 4, 4)
    elif case == "bf16 1x1 concat":                 # conv1x1.hip, lean epilogue over four cout blocks
        srcs = [hc.to_ndhwc(rnd(1, 128, 8, 32, 32, seed=76)).bfloat16().cuda(), hc.to_ndhwc(rnd(1, 128, 8, 32, 32, seed=77)).bfloat16().cuda()]
        w, kw, dhw = rnd(128, 256, 1, 1, 1, seed=78, scale=0.05), dict(precision=5, out_bf16=True, want_stats=False), (8, 32, 32)
    else:                                           # f16 tensors, same-shape residual read behind the same kind of loads
        srcs = [hc.to_ndhwc(rnd(1, 128, 8, 32, 32, seed=79)).half().cuda()]
        res = hc.to_ndhwc(rnd(1, 128, 8, 32, 32, seed=80)).half().cuda()
        w = rnd(128, 128, 3, 3, 3, seed=81, scale=0.03)
        kw, dhw = dict(precision=4, out_f16=True, res=res, res_mode=H.RES_SAME), (8, 32, 32)
    b = rnd(w.shape[0], seed=82).cuda()
    A = (1.0 + 0.1 * rnd(1, w.shape[1], seed=83)).cuda()
    B = (0.1 * rnd(1, w.shape[1], seed=84)).cuda()
    if w.shape[2] == 3:
        kw.update(aff=(A, B), act=H.ACT_SILU)
    first = None
    for it in range(40):
        out, stats, _ = hc.conv3d(srcs, w.cuda(), b, dhw, **kw)
        assert torch.isfinite(out.float()).all()
        got = (out.clone(), None if stats is None else stats.clone())
        if first is None:
            first = got
            if case == "f16x3 split slabs":
                assert hc.LAST_PLAN["split"] >= 8
        else:
            assert torch.equal(got[0], first[0]), (case, it)
            assert first[1] is None or torch.equal(got[1], first[1]), (case, it)
