"""
BASELINE.json's configurations at (or as near as a test can get to) their stated workloads,
on the GPU, through the HIP engine:

  C2  published architecture, 250 dependent DDPM steps -- pinned to the REFERENCE's own
      250-step run (tests/golden/sampler_published250.npz, 1x1x8x32x32) at the north_star
      bar (1e-3), in the exact-fp32 and in the default f16x3 arithmetic;
  C3  one GPU's share of the 8-GPU run: 8 volumes of 64^3 in one launch train, each equal to
      its batch-1 result;
  C4  50-step DDIM (respace.py "ddim50"), published architecture, 1x64^3, in the bf16 mode the
      config names (convert_to_bf16()) and in the reference's own reduced precision (convert_to_fp16());
  C5  1x128^3 with attention at ds 8 (T = 32 768 tokens): full-size properties AND one forward against
      the CPU oracle (query-blocked attention, pinned to the reference's attention outputs);
  and the 64^3-level Winograd layers (128->128, 256->128 concat, upsampled input)
  checked DIRECTLY against F.conv3d on the CPU.

A 250-step volume takes the CPU an hour, so the full-size 250-step case is compared with a fixture:
the REFERENCE's own run of config 2 (tests/golden/sampler250_64.npz, generated once in the build
container by tests/golden/make_golden.py sampler250_64).  Single forwards at full size are compared with
the CPU oracle inside the tests (64^3: seconds; 128^3 with attention: about two minutes).
"""

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err, rel_err_per_channel
from guided_diffusion import synth
from test_gpu_model import PUBLISHED, build, inputs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_c2_full_size_250_steps_vs_reference_golden(golden, precision):
    """BASELINE config 2 ITSELF -- published architecture, 1x1x64^3, timestep_respacing "250", all 250 dependent
    steps -- against the REFERENCE's own run of it on the same weights, conditioning volume and injected noise
    (tests/golden/sampler250_64.npz: about one CPU-hour of /root/reference in the build container,
    make_golden.py sampler250_64), at the north_star bar of 1e-3, in the exact-fp32 and in the default f16x3
    arithmetic.  The samples after 1, 50 and 150 steps show how the difference grows along the chain."""
    g = golden("sampler250_64.npz")
    model, diff = build(PUBLISHED, "250", precision=precision)
    shape = (1, 1, 64, 64, 64)
    T = diff.num_timesteps
    assert T == 250
    draws = [torch.from_numpy(a).cuda() for a in synth.synth_noise(shape, T + 1, seed=10)]
    lr = torch.from_numpy(synth.synth_low_res(shape, seed=1234)).cuda()
    trace, errs = [], {}
    for k, o in enumerate(diff.p_sample_loop_progressive(model, shape, draws[0], model_kwargs={"low_res": lr},
                                                         step_noise=draws[1:])):
        s = o["sample"]
        trace.append((float(s.mean()), float(o["pred_xstart"].mean()), float(s.std())))
        key = "after%d" % (k + 1)
        if key in g.files:
            errs[key] = rel_err(s.cpu().numpy(), g[key])
        last = o
    errs["sample"] = rel_err(last["sample"].cpu().numpy(), g["sample"])
    print("config 2 at full size, 250 steps (%s): rel err along the chain %s" % (precision, errs))
    assert set(errs) == {"after1", "after50", "after150", "sample"}
    assert max(errs.values()) < 1e-3, errs
    assert np.allclose(np.array(trace), g["trace"], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_c2_published_250_steps_vs_reference_golden(golden, precision):
    """The north_star bar as stated: 1e-3 relative after 250 DEPENDENT steps on the published
    architecture.  Intermediate samples show how the difference grows along the chain."""
    g = golden("sampler_published250.npz")
    model, diff = build(PUBLISHED, "250", precision=precision)
    shape = (1, 1, 8, 32, 32)
    T = diff.num_timesteps
    assert T == 250
    draws = [torch.from_numpy(a).cuda() for a in synth.synth_noise(shape, T + 1, seed=10)]
    lr = torch.from_numpy(synth.synth_low_res(shape, seed=1234)).cuda()
    trace, errs = [], {}
    for k, o in enumerate(diff.p_sample_loop_progressive(model, shape, draws[0], model_kwargs={"low_res": lr},
                                                         step_noise=draws[1:])):
        s = o["sample"]
        trace.append((float(s.mean()), float(o["pred_xstart"].mean()), float(s.std())))
        key = "after%d" % (k + 1)
        if key in g.files:
            errs[key] = rel_err(s.cpu().numpy(), g[key])
        last = o
    errs["sample"] = rel_err(last["sample"].cpu().numpy(), g["sample"])
    errs["pred_xstart"] = rel_err(last["pred_xstart"].cpu().numpy(), g["pred_xstart"])
    print("250-step published (%s): rel err along the chain %s" % (precision, errs))
    assert max(errs.values()) < 1e-3, errs
    assert np.allclose(np.array(trace), g["trace"], rtol=1e-3, atol=1e-4)


def test_c3_batch8_per_gpu_share_matches_batch1():
    """BASELINE config 3 gives each of 8 GPUs 8 volumes of 64^3.  One forward of that batch:
    every volume must equal its own batch-1 result (other tile counts / split-K factors change
    summation orders only)."""
    B = 8
    shape1 = (1, 1, 64, 64, 64)
    xs = [torch.from_numpy(synth.synth_noise(shape1, 1, seed=20 + b)[0]) for b in range(B)]
    lrs = [torch.from_numpy(synth.synth_low_res(shape1, seed=300 + b)) for b in range(B)]
    ts = [617, 41, 999, 0, 5, 250, 761, 333]
    model, _ = build(PUBLISHED)
    with torch.no_grad():
        yb = model(torch.cat(xs).cuda(), torch.tensor(ts).cuda(), low_res=torch.cat(lrs).cuda()).cpu().numpy()
        assert yb.shape == (B, 2, 64, 64, 64) and np.isfinite(yb).all()
        for b in range(B):
            y1 = model(xs[b].cuda(), torch.tensor(ts[b:b + 1]).cuda(), low_res=lrs[b].cuda()).cpu().numpy()
            assert rel_err_per_channel(yb[b:b + 1], y1) < 2e-5, b


def test_c4_ddim50_published_fp16_mode_psnr():
    """BASELINE config 4's run: published architecture, 1x64^3, "ddim50", eta 0, with
    model.convert_to_fp16() (scripts/test.py:33-34).  There is no reduced-precision CPU oracle
    (the reference's fp16 path does not run on CPU), so the judge is PSNR against the
    fp32-grade default mode on the same noise.  Bar: operands rounded to f16 carry 2^-11 relative
    error each; over a K = 3456 reduction the output error is ~2^-11 / sqrt(K) * |w||x| ~ 1e-4
    relative per layer, and 50 clipped DDIM steps measured 55-70 dB on the tiny network -- 45 dB
    (error 5.6e-3 of the [-1, 1] range) leaves room for the deeper network without admitting a
    wrong kernel (a dropped tap or channel costs > 20 dB)."""
    shape = (1, 1, 64, 64, 64)
    draws = [torch.from_numpy(a).cuda() for a in synth.synth_noise(shape, 51, seed=10)]
    lr = torch.from_numpy(synth.synth_low_res(shape, seed=1234)).cuda()
    model, diff = build(PUBLISHED, "ddim50")
    assert diff.num_timesteps == 50 and list(diff.timestep_map[:3]) == [0, 20, 40]
    ref = diff.ddim_sample_loop(model, shape, draws[0], model_kwargs={"low_res": lr}, step_noise=draws[1:])
    model.convert_to_fp16()
    assert model.dtype == torch.float16
    out = diff.ddim_sample_loop(model, shape, draws[0], model_kwargs={"low_res": lr}, step_noise=draws[1:])
    assert torch.isfinite(out).all() and float(out.abs().max()) <= 1.0 + 1e-6   # last step is clipped x0
    # the 50-step chain is repeatable bit for bit in this mode too (its kernels are not the default mode's)
    again = diff.ddim_sample_loop(model, shape, draws[0], model_kwargs={"low_res": lr}, step_noise=draws[1:])
    assert torch.equal(out, again)
    mse = float(((out - ref) ** 2).mean())
    psnr = 10 * np.log10(4.0 / max(mse, 1e-30))
    print("config 4 (ddim50, f16 operands) PSNR vs f16x3: %.1f dB" % psnr)
    assert psnr > 45.0, psnr


def test_c4_ddim50_published_bf16_mode_psnr():
    """BASELINE config 4 AS WRITTEN: published architecture, 1x64^3, "ddim50", eta 0, bf16 -- operands
    rounded to bf16 on the matrix cores and the residual stream stored in bf16
    (model.convert_to_bf16()).  PSNR against the fp32-grade default mode on the same noise; bf16 has
    8 significant bits, so the bar is lower than the fp16 mode's."""
    shape = (1, 1, 64, 64, 64)
    draws = [torch.from_numpy(a).cuda() for a in synth.synth_noise(shape, 51, seed=10)]
    lr = torch.from_numpy(synth.synth_low_res(shape, seed=1234)).cuda()
    model, diff = build(PUBLISHED, "ddim50")
    ref = diff.ddim_sample_loop(model, shape, draws[0], model_kwargs={"low_res": lr}, step_noise=draws[1:])
    model.convert_to_bf16()
    out = diff.ddim_sample_loop(model, shape, draws[0], model_kwargs={"low_res": lr}, step_noise=draws[1:])
    assert torch.isfinite(out).all() and float(out.abs().max()) <= 1.0 + 1e-6
    again = diff.ddim_sample_loop(model, shape, draws[0], model_kwargs={"low_res": lr}, step_noise=draws[1:])
    assert torch.equal(out, again)                     # bitwise repeatable over the 50-step chain
    mse = float(((out - ref) ** 2).mean())
    psnr = 10 * np.log10(4.0 / max(mse, 1e-30))
    print("config 4 (ddim50, bf16 operands + bf16 residual stream) PSNR vs f16x3: %.1f dB" % psnr)
    # PARITY UNPINNED against the reference (it has fp16 only; no fixture covers bf16): the bar is this
    # package's own fp32-grade run.  Measured 36.6 dB; 34 dB catches a regression of a few dB.
    assert psnr > 34.0, psnr


def test_c5_full_size_attention_forward_properties():
    """BASELINE config 5 at FULL size: large_size=128, attention_resolutions="16" (attention at
    ds 8: T = 128*16*16 = 32 768 tokens, 5 blocks), 1x128^3.  Default arithmetic vs the exact-fp32
    mode, bitwise repeatability, finiteness."""
    arch = dict(PUBLISHED, large_size=128, small_size=128, attention_resolutions="16")
    shape = (1, 1, 128, 128, 128)
    x, lr = inputs(shape)
    t = torch.tensor([444])
    model, _ = build(arch)
    assert sum(1 for l in model.topology.all_layers() if l.kind == "attn") == 5
    with torch.no_grad():
        y = model(x.cuda(), t.cuda(), low_res=lr.cuda())
        y_again = model(x.cuda(), t.cuda(), low_res=lr.cuda())
    assert tuple(y.shape) == (1, 2, 128, 128, 128) and torch.isfinite(y).all()
    assert torch.equal(y, y_again)
    y_def = y.cpu().numpy()
    del model, y, y_again
    torch.cuda.empty_cache()
    exact, _ = build(arch, precision="f32")
    with torch.no_grad():
        y_exact = exact(x.cuda(), t.cuda(), low_res=lr.cuda()).cpu().numpy()
    del exact
    torch.cuda.empty_cache()
    assert rel_err_per_channel(y_def, y_exact) < 1e-4
    # ... and against the CPU oracle (r04), so that config 5 no longer rests on the library's own exact mode: ONE
    # forward of the same network on the host, its five attention blocks through the query-blocked softmax that
    # tests/test_oracle_golden.py pins to the reference's attention outputs (54 TFLOP: about two minutes on the
    # box's 16 granted cores).  Same single-forward bar, per output channel, for both arithmetics.
    from oracle import unet_ref
    cfg = unet_ref.sr_config(**arch)
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(unet_ref.param_shapes(cfg), 0).items()}
    torch.set_num_threads(min(16, torch.get_num_threads() or 16))
    with torch.no_grad():
        ref = unet_ref.unet_forward(sd, cfg, x, t, lr).numpy()
    assert rel_err_per_channel(y_def, ref) < 1e-4
    assert rel_err_per_channel(y_exact, ref) < 1e-4


@pytest.mark.parametrize("ci,upsampled", [(128, False), (256, False), (128, True)])
def test_full_size_winograd_kernel_vs_cpu_conv(ci, upsampled):
    """The kernel the 64^3 level auto-selects (>= 2048 workgroups: the wave-specialised, persistent
    Winograd-D form) compared DIRECTLY with F.conv3d on the CPU: ci -> 128 @ 64x64x64, GroupNorm-style
    affine + SiLU prologue, residual and statistics, precision 3 -- the layer shapes that carry
    60 % of the forward.  256 = the decoder's virtual concat of two 128-channel tensors; upsampled =
    the up-ResBlock's nearest-neighbour input mode."""
    import hipcall as hc
    import guided_diffusion._hip as H
    from test_gpu_ops import TOL, check_stats, rnd
    D = Hh = W = 64
    hs, ws = (Hh // 2, W // 2) if upsampled else (Hh, W)
    x = rnd(1, ci, D, hs, ws, seed=1)
    w = rnd(128, ci, 3, 3, 3, seed=2, scale=0.03)
    b = rnd(128, seed=3)
    a = 1.0 + 0.1 * rnd(1, ci, seed=4)
    bb = 0.1 * rnd(1, ci, seed=5)
    res = rnd(1, 128, D, Hh, W, seed=6)
    xin = F.silu(x * a[:, :, None, None, None] + bb[:, :, None, None, None])
    if upsampled:
        xin = F.interpolate(xin, (D, Hh, W), mode="nearest")
    torch.set_num_threads(16)
    ref = F.conv3d(xin, w, b, padding=1) + res
    xd = hc.to_ndhwc(x).cuda()
    srcs = [xd] if ci == 128 else [xd[..., :128].contiguous(), xd[..., 128:].contiguous()]
    out, stats, _ = hc.conv3d(srcs, w.cuda(), b.cuda(), (D, Hh, W), aff=(a.cuda(), bb.cuda()), act=H.ACT_SILU,
                              in_mode=H.IN_UP if upsampled else H.IN_SAME,
                              res=hc.to_ndhwc(res).cuda(), res_mode=H.RES_SAME, precision=3)
    got = hc.to_ncdhw(out.cpu())
    assert rel_err_per_channel(got.numpy(), ref.numpy()) < TOL
    check_stats(stats, ref)


@pytest.mark.parametrize("precision,dims", [(1, (64, 32, 32)), (5, (64, 64, 64)), (1, (64, 64, 64))])
def test_full_size_skip_connection_conv_vs_cpu(precision, dims):
    """The decoder's 1x1 skip conv at its published shapes -- virtual concat 128 + 128 -> 128 -- against
    F.conv3d on the CPU: the register-fed kernel (conv1x1.hip) at 64x32x32 in the default arithmetic and at
    64^3 in the bf16 mode (2048 workgroups, bf16 tensors), and the 64^3 layer in the default arithmetic (402 MB of
    fp32 tensors; on the general kernel until r04's lean epilogue).  Residual accumulated in place, as the
    engine does (the conv2 of the block adds onto the skip result)."""
    import hipcall as hc
    import guided_diffusion._hip as H
    from test_gpu_ops import rnd
    D, Hh, W = dims
    half = precision == 5
    x = rnd(1, 256, D, Hh, W, seed=11) * 2.0
    w = rnd(128, 256, 1, 1, 1, seed=12, scale=0.05)
    b = rnd(128, seed=13)
    xs = x.bfloat16() if half else x
    torch.set_num_threads(16)
    if half:
        ref = F.conv3d(xs.float().double(), w.bfloat16().double(), b.double()).float()
    else:
        ref = F.conv3d(x.double(), w.double(), b.double()).float()
    xd = hc.to_ndhwc(xs).cuda()
    srcs = [xd[..., :128].contiguous(), xd[..., 128:].contiguous()]
    out, _, _ = hc.conv3d(srcs, w.cuda(), b.cuda(), (D, Hh, W), precision=precision, want_stats=False)
    got = hc.to_ncdhw(out.float().cpu())
    assert rel_err_per_channel(got.numpy(), ref.numpy()) < (1e-5 if half else 3e-6)


def test_full_size_down_block_prepass_and_small_level_winograd():
    """Two r03 paths at the published shapes.  (1) ddpm3d_pool_act on the 64^3 -> 64x32x32 down ResBlock's input
    (GroupNorm-style affine + SiLU + AvgPool3d((1,2,2))) followed by the Winograd-D conv on the plain pooled
    tensor, against avg_pool3d + conv3d on the CPU.  (2) 512 -> 512 @ 64x4x4: the Winograd-D form on 4x4x8
    tiles with its 16-way split over Cin, residual and statistics, against F.conv3d."""
    import hipcall as hc
    import guided_diffusion._hip as H
    from test_gpu_ops import TOL, check_stats, rnd
    lib = H.load()
    torch.set_num_threads(16)
    D, Hh, W, Cn = 64, 32, 32, 128
    x = rnd(1, Cn, D, 2 * Hh, 2 * W, seed=21)
    a = 1.0 + 0.1 * rnd(1, Cn, seed=22)
    bb = 0.1 * rnd(1, Cn, seed=23)
    w = rnd(128, Cn, 3, 3, 3, seed=24, scale=0.03)
    b = rnd(128, seed=25)
    pooled_ref = F.avg_pool3d(F.silu(x * a[:, :, None, None, None] + bb[:, :, None, None, None]), (1, 2, 2))
    ref = F.conv3d(pooled_ref, w, b, padding=1)
    xd = hc.to_ndhwc(x).cuda()
    pooled = torch.empty(1, D, Hh, W, Cn, dtype=torch.float32, device="cuda")
    ad, bd = a.cuda(), bb.cuda()
    H.check(lib.ddpm3d_pool_act(H.ptr(xd), H.ptr(ad), H.ptr(bd), H.ACT_SILU, 1, 1, D, Hh, W, Cn, H.ptr(pooled), 0,
                                H.stream()))
    torch.cuda.synchronize()
    assert rel_err(hc.to_ncdhw(pooled.cpu()).numpy(), pooled_ref.numpy()) < 2e-6
    bound = pooled_ref.abs().amax().reshape(1, 1).cuda()
    out, stats, _ = hc.conv3d([pooled], w.cuda(), b.cuda(), (D, Hh, W), precision=3, bound=bound)
    assert rel_err_per_channel(hc.to_ncdhw(out.cpu()).numpy(), ref.numpy()) < TOL
    check_stats(stats, ref)
    # (2)
    x4 = rnd(1, 512, 64, 4, 4, seed=31)
    w4 = rnd(512, 512, 3, 3, 3, seed=32, scale=0.02)
    b4 = rnd(512, seed=33)
    a4 = 1.0 + 0.1 * rnd(1, 512, seed=34)
    bb4 = 0.1 * rnd(1, 512, seed=35)
    r4 = rnd(1, 512, 64, 4, 4, seed=36)
    ref4 = F.conv3d(F.silu(x4 * a4[:, :, None, None, None] + bb4[:, :, None, None, None]), w4, b4, padding=1) + r4
    out, stats, _ = hc.conv3d([hc.to_ndhwc(x4).cuda()], w4.cuda(), b4.cuda(), (64, 4, 4), aff=(a4.cuda(), bb4.cuda()),
                              act=H.ACT_SILU, res=hc.to_ndhwc(r4).cuda(), res_mode=H.RES_SAME, precision=3)
    assert rel_err_per_channel(hc.to_ncdhw(out.cpu()).numpy(), ref4.numpy()) < TOL
    check_stats(stats, ref4)


def test_script_patch_shape_96_cubed():
    """The shape the reference's launcher really runs (test_DDPM_3d_mpi.sh / README.md: --large_size 96,
    patches of 96^3, published architecture): resolutions 96 / 48 / 24 / 12 / 6, i.e. tile grids that
    are NOT powers of two -- ragged 4x4 tiles at the 6x6 level (general epilogue), 12x12 = 1.5 8x8
    tiles, odd workgroup counts for the XCD orders and the split-K model.  Default arithmetic vs the
    exact-fp32 mode per output channel, bitwise repeatability, and one DDPM step through the fused
    update."""
    arch = dict(PUBLISHED, large_size=96, small_size=96)
    shape = (1, 1, 96, 96, 96)
    x, lr = inputs(shape)
    t = torch.tensor([444])
    model, diff = build(arch, "250")
    with torch.no_grad():
        y = model(x.cuda(), t.cuda(), low_res=lr.cuda())
        y_again = model(x.cuda(), t.cuda(), low_res=lr.cuda())
    assert tuple(y.shape) == (1, 2, 96, 96, 96) and torch.isfinite(y).all()
    assert torch.equal(y, y_again)
    out = diff.p_sample(model, x.cuda(), torch.tensor([249]).cuda(), model_kwargs={"low_res": lr.cuda()},
                        noise=torch.from_numpy(synth.synth_noise(shape, 1, seed=5)[0]).cuda())
    assert torch.isfinite(out["sample"]).all() and float(out["pred_xstart"].abs().max()) <= 1.0
    y_def = y.cpu().numpy()
    del model, y, y_again, out
    torch.cuda.empty_cache()
    exact, _ = build(arch, precision="f32")
    with torch.no_grad():
        y_exact = exact(x.cuda(), t.cuda(), low_res=lr.cuda()).cpu().numpy()
    del exact
    torch.cuda.empty_cache()
    assert rel_err_per_channel(y_def, y_exact) < 1e-4


def test_c2_full_size_250_steps_default_vs_exact():
    """BASELINE config 2 exactly as benchmarked -- published architecture, 1x64^3, all 250 dependent DDPM
    steps -- in the default f16x3 arithmetic against the exact-fp32 mode on the same noise (the CPU
    oracle needs an hour for this volume; the exact mode is pinned to the reference by the 8x32x32
    chain above).  The north_star bar at the benchmark's own size."""
    shape = (1, 1, 64, 64, 64)
    draws = [torch.from_numpy(a).cuda() for a in synth.synth_noise(shape, 251, seed=10)]
    lr = torch.from_numpy(synth.synth_low_res(shape, seed=1234)).cuda()
    outs = []
    for precision in ("f16x3", "f32"):
        model, diff = build(PUBLISHED, "250", precision=precision)
        outs.append(diff.p_sample_loop(model, shape, draws[0], model_kwargs={"low_res": lr},
                                       step_noise=draws[1:]).cpu().numpy())
        del model
        torch.cuda.empty_cache()
    assert np.isfinite(outs[0]).all()
    err = rel_err(outs[0], outs[1])
    print("config 2 at full size, 250 steps: f16x3 vs exact fp32 rel err %.2e" % err)
    assert err < 1e-3, err
