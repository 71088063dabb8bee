"""
Range safety of the split-f16 arithmetic (f16x3 / f16 modes).

The fp32 operands of every product are scaled by a power of two before the f16 hi/lo split.  Up to
r01 that scale was a constant (x8) followed by a clamp to +-60000: |activation| > 7500 was clipped
silently.  Now the scale is chosen per launch from an upper bound of the input's magnitude
(ddpm3d_conv_desc.in_bound: from the GroupNorm partial sums via ddpm3d_gn_finalize, or
ddpm3d_absmax), nothing is clamped, and tensors of very small magnitude keep their low bits.

Each case compares the f16x3 arithmetic with an fp64 reference PER OUTPUT CHANNEL and with the
exact-fp32 mode's own error on the same data, for inputs from 1e-6 to 1e5 through the paths that
read un-normalised tensors: the 1x1 skip conv, the first (planar) conv, the Winograd-D conv behind a
large affine, and attention.
"""

import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err_per_channel
from test_gpu_model import TINY, build, inputs
from test_gpu_ops import rnd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hc():
    import hipcall
    return hipcall


def chan_err(got_ncdhw, ref_ncdhw):
    return rel_err_per_channel(got_ncdhw.double().numpy(), ref_ncdhw.double().numpy())


@pytest.mark.parametrize("mag", [1e-6, 1e-3, 1.0, 7.5e3, 1e5])
def test_skip_conv_1x1_any_magnitude(hc, mag):
    """The ResBlock's 1x1 skip conv reads the raw residual stream (unet.py:215-222, 256)."""
    x = rnd(1, 256, 4, 16, 16, seed=1) * mag
    x[:, :64] *= 1e-3                                   # a thousand-fold range between channels
    w = rnd(128, 256, 1, 1, 1, seed=2, scale=0.05)
    b = rnd(128, seed=3) * mag
    ref = F.conv3d(x.double(), w.double(), b.double())
    errs = []
    for precision in (0, 1):
        out, _, _ = hc.conv3d([hc.to_ndhwc(x).cuda()], w.cuda(), b.cuda(), (4, 16, 16), precision=precision)
        assert torch.isfinite(out).all()
        errs.append(chan_err(hc.to_ncdhw(out.cpu()), ref))
    assert errs[0] < 2e-6 and errs[1] < 2e-6, errs
    assert errs[1] < 2 * errs[0] + 2e-7, errs


@pytest.mark.parametrize("mag", [1e-5, 1.0, 1e5])
def test_first_conv_planar_any_magnitude(hc, mag):
    """The first conv reads the two input volumes as they come: scripts/test.py feeds PET data
    without normalisation (x: the diffusion state, low_res: the measured volume)."""
    xs = rnd(1, 1, 6, 16, 24, seed=4) * 3.0
    lr = rnd(1, 1, 6, 16, 24, seed=5).abs() * mag
    w = rnd(128, 2, 3, 3, 3, seed=6, scale=0.2)
    b = rnd(128, seed=7)
    ref = F.conv3d(torch.cat([xs, lr], 1).double(), w.double(), b.double(), padding=1)
    errs = []
    for precision in (0, 1):
        out, _, _ = hc.conv3d([xs.cuda(), lr.cuda()], w.cuda(), b.cuda(), (6, 16, 24), planar=True,
                              precision=precision)
        assert torch.isfinite(out).all()
        errs.append(chan_err(hc.to_ncdhw(out.cpu()), ref))
    assert errs[0] < 2e-6 and errs[1] < 2e-6, errs
    assert errs[1] < 2 * errs[0] + 2e-7, errs


@pytest.mark.parametrize("gain", [1e-4, 1.0, 3e3])
def test_winograd_conv_behind_large_affine(hc, gain):
    """GroupNorm + FiLM can scale the normalised tensor by any factor (unet.py:248-252); the
    Winograd-D input transform adds two planes on top (its own factor 2 is part of the scale)."""
    import guided_diffusion._hip as H
    x = rnd(1, 64, 6, 16, 16, seed=8)
    A = (1.0 + 0.1 * rnd(1, 64, seed=9)) * gain
    B = 0.1 * rnd(1, 64, seed=10) * gain
    w = rnd(128, 64, 3, 3, 3, seed=11, scale=0.03)
    b = rnd(128, seed=12)
    xin = F.silu(x.double() * A.double()[:, :, None, None, None] + B.double()[:, :, None, None, None])
    ref = F.conv3d(xin, w.double(), b.double(), padding=1)
    errs = []
    for precision in (0, 3):
        out, _, _ = hc.conv3d([hc.to_ndhwc(x).cuda()], w.cuda(), b.cuda(), (6, 16, 16), aff=(A.cuda(), B.cuda()),
                              act=H.ACT_SILU, precision=precision)
        assert torch.isfinite(out).all()
        errs.append(chan_err(hc.to_ncdhw(out.cpu()), ref))
    # the fast SiLU of the split modes (v_exp / v_rcp, ~2 ulp) sits on top of the product error
    assert errs[0] < 4e-6 and errs[1] < 8e-6, errs


def test_bound_too_small_is_loud_not_wrong(hc):
    """in_bound is a contract: an understated bound overflows f16 and the output is non-finite --
    never a silently clipped result."""
    x = rnd(1, 32, 2, 8, 8, seed=13) * 1e4
    w = rnd(32, 32, 1, 1, 1, seed=14, scale=0.05)
    b = rnd(32, seed=15)
    lie = torch.full((1, 1), 1.0, device="cuda")
    out, _, _ = hc.conv3d([hc.to_ndhwc(x).cuda()], w.cuda(), b.cuda(), (2, 8, 8), precision=1, bound=lie)
    assert not torch.isfinite(out).all()
    ok, _, _ = hc.conv3d([hc.to_ndhwc(x).cuda()], w.cuda(), b.cuda(), (2, 8, 8), precision=1)
    ref = F.conv3d(x.double(), w.double(), b.double())
    assert chan_err(hc.to_ncdhw(ok.cpu()), ref) < 2e-6


def test_split_modes_require_a_bound(hc):
    import guided_diffusion._hip as H
    lib = H.load()
    x = torch.zeros(1, 2, 8, 8, 16, device="cuda")
    w = torch.zeros(32, 16, 1, 1, 1, device="cuda")
    wp = hc.pack(w, 1)
    out = torch.empty(1, 2, 8, 8, 32, device="cuda")
    d = H.ConvDesc()
    d.N, d.D, d.H, d.W, d.Cin, d.Cout, d.ksize, d.C0 = 1, 2, 8, 8, 16, 32, 1, 16
    d.src0, d.w_packed, d.bias, d.out = H.ptr(x), H.ptr(wp), H.ptr(torch.zeros(32, device="cuda")), H.ptr(out)
    d.precision = 1
    assert lib.ddpm3d_conv3d(C.byref(d), H.stream()) == -1            # DDPM3D_EINVAL
    assert b"in_bound" in lib.ddpm3d_last_error()


def test_gn_finalize_bounds_are_upper_bounds(hc):
    """ddpm3d_gn_finalize's bound output really bounds |act(A x + B)| and |x| per (sample, group), and
    is not uselessly loose (within sqrt(rows per statistics row) of the truth)."""
    import guided_diffusion._hip as H
    x = rnd(2, 64, 5, 12, 12, seed=16) * 37.0
    x[1] *= 1e-3
    w = torch.zeros(64, 64, 1, 1, 1)
    for c in range(64):
        w[c, c] = 1.0                                   # identity 1x1 conv: its epilogue emits x's statistics
    out, stats, _ = hc.conv3d([hc.to_ndhwc(x).cuda()], w.cuda(), torch.zeros(64).cuda(), (5, 12, 12), precision=0)
    gamma, beta = 1 + 0.1 * rnd(64, seed=17), 0.1 * rnd(64, seed=18)
    A, B, bound = hc.gn_finalize([stats], 5 * 12 * 12, gamma.cuda(), beta.cuda(), with_bound=True)
    A, B, bound = A.cpu(), B.cpu(), bound.cpu()
    y = x * A[:, :, None, None, None] + B[:, :, None, None, None]
    for n in range(2):
        for g in range(32):
            cs = slice(2 * g, 2 * g + 2)
            true_y = float(y[n, cs].abs().max())
            true_x = float(x[n, cs].abs().max())
            assert bound[n, g, 0] >= true_y * (1 - 1e-6) and bound[n, g, 1] >= true_x * (1 - 1e-6)
            assert bound[n, g, 1] < 40 * true_x           # rows of <= 128 voxels x 2 channels
    # bounds only (tensor consumed without a GroupNorm)
    lib = H.load()
    b2 = torch.full((2, 32, 2), float("nan"), device="cuda")
    H.check(lib.ddpm3d_gn_finalize(H.ptr(stats), 64, stats.shape[2], 0, 0, 0, 2, 32, float(5 * 12 * 12), 1e-5,
                                   0, 0, 0, 0, 0, 0, 0, H.ptr(b2), H.stream()))
    torch.cuda.synchronize()
    assert torch.equal(b2[..., 0].cpu(), bound[..., 1]) and torch.equal(b2[..., 1].cpu(), bound[..., 1])


@pytest.mark.parametrize("mag", [1e-3, 1.0, 200.0])
def test_attention_any_magnitude(hc, mag):
    """q, k, v of any common magnitude (the qkv conv's output is not normalised), f16x3 products vs
    the reference's materialised softmax evaluated in fp64.  One scale serves q, k and v, so full
    22-bit operands need them within 2^18 of each other (f16's normal range minus the split) -- the
    attention block's qkv come out of one GroupNorm-fed conv and are."""
    import guided_diffusion._hip as H
    lib = H.load()
    N, T, heads, ch = 1, 160, 2, 64
    qkv = rnd(N, T, heads * 3 * ch, seed=19) * mag
    # keep the logits moderate (softmax of huge logits is a one-hot either way): scale q down
    q = qkv.reshape(N, T, heads, 3, ch)
    q[:, :, :, 0] *= 1.0 / max(mag * mag, 1e-6) if mag > 1 else 1.0
    qkv = q.reshape(N, T, heads * 3 * ch).contiguous()
    qd = qkv.cuda()
    qb = qd.abs().reshape(N, -1).amax(dim=1).contiguous()
    out = torch.empty(N, T, heads * ch, device="cuda")
    H.check(lib.ddpm3d_attention_p(H.ptr(qd), N, T, heads, ch, H.PREC_F16X3, H.ptr(qb), 1, 1, H.ptr(out), H.stream()))
    torch.cuda.synchronize()
    x = qkv.double().reshape(N, T, heads, 3, ch)
    qq, kk, vv = x[:, :, :, 0], x[:, :, :, 1], x[:, :, :, 2]
    s = 1.0 / np.sqrt(np.sqrt(ch))
    wgt = torch.softmax(torch.einsum("nthc,nshc->nhts", qq * s, kk * s), dim=-1)
    ref = torch.einsum("nhts,nshc->nthc", wgt, vv).reshape(N, T, heads * ch)
    assert torch.isfinite(out).all()
    err = float((out.cpu().double() - ref).abs().max() / ref.abs().max())
    assert err < 1e-5, err


def test_network_with_unnormalised_input_volume():
    """Whole engine: low_res up to 1e4 (PET counts without normalisation) in the default f16x3
    arithmetic against the exact-fp32 mode, per output channel; and a tiny-magnitude volume."""
    shape = (1, 1, 8, 32, 32)
    x, lr = inputs(shape)
    t = torch.tensor([321])
    for mag in (1e4, 1e-5):
        ys = []
        for precision in ("f16x3", "f32"):
            model, _ = build(TINY, precision=precision)
            with torch.no_grad():
                ys.append(model(x.cuda(), t.cuda(), low_res=(lr * mag).cuda()).cpu().numpy())
        assert np.isfinite(ys[0]).all()
        assert rel_err_per_channel(ys[0], ys[1]) < 1e-4, mag
