#!/usr/bin/env python3
"""
Randomised differential run of the round-3 kernels against torch on the CPU (one-off confidence run; the
fixed cases live in tests/): conv1x1.hip (skip-connection 1x1 convs), the Winograd-D form on 4x4x8 tiles,
ddpm3d_pool_act.  Shapes, splits over the concat, residuals, storage types and precisions are drawn at random;
every case must meet the bar of its arithmetic mode per output channel.

    python tools/stress_new_paths.py [--cases 150] [--seed 0] > gpurun_out/stress.txt
"""

import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "3d-denoising-diffusion-model_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import hipcall as hc  # noqa: E402
from conftest import rel_err_per_channel  # noqa: E402
from guided_diffusion import _hip as H  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=150)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    g = torch.Generator().manual_seed(a.seed)
    lib = H.load()
    worst = {}
    torch.set_num_threads(16)

    def rnd(*shape, scale=1.0):
        return torch.randn(*shape, generator=g) * scale

    def note(kind, e, tol, desc):
        w = worst.get(kind, (0.0, tol, ""))
        if e / tol > w[0] / w[1]:
            worst[kind] = (e, tol, desc)
        if not e < tol:
            print("FAIL %s: %.3e >= %.1e  %s" % (kind, e, tol, desc))
            sys.exit(1)

    for case in range(a.cases):
        which = case % 3
        if which == 0:      # ---- 1x1 skip conv
            prec = int(rng.choice([1, 2, 5]))
            N = int(rng.integers(1, 4))
            D, Hh, W = int(rng.integers(1, 13)), int(rng.integers(1, 19)), int(rng.integers(1, 19))
            c0 = 32 * int(rng.integers(1, 9))
            c1 = 32 * int(rng.integers(0, 5))
            co = 128 * int(rng.integers(1, 4))
            store = {1: rng.choice(["f32", "f16"]), 2: rng.choice(["f32", "f16"]), 5: rng.choice(["f32", "bf16"])}[prec]
            dt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[store]
            with_res = bool(rng.integers(0, 2))
            x = (rnd(N, c0 + c1, D, Hh, W) * 2.0).to(dt)
            w = rnd(co, c0 + c1, 1, 1, 1, scale=0.05)
            b = rnd(co)
            res = rnd(N, co, D, Hh, W) if with_res else None
            xv = x.float()
            if prec == 5:
                ref = F.conv3d(xv.bfloat16().double(), w.bfloat16().double(), b.double()).float()
            else:
                ref = F.conv3d(xv.double(), w.double(), b.double()).float()
            if with_res:
                ref = ref + res
            srcs = [hc.to_ndhwc(x[:, :c0]).cuda()] + ([hc.to_ndhwc(x[:, c0:]).cuda()] if c1 else [])
            out, _, _ = hc.conv3d(srcs, w.cuda(), b.cuda(), (D, Hh, W), precision=prec, want_stats=False,
                                  res=hc.to_ndhwc(res).cuda() if with_res else None,
                                  res_mode=H.RES_SAME if with_res else H.RES_NONE)
            e = rel_err_per_channel(hc.to_ncdhw(out.cpu()).numpy(), ref.numpy())
            note("conv1x1 p%d" % prec, e, {1: 5e-6, 2: 2e-3, 5: 1e-5}[prec],
                 "N=%d %dx%dx%d %d+%d->%d %s res=%d" % (N, D, Hh, W, c0, c1, co, store, with_res))
        elif which == 1:    # ---- Winograd-D on 4x4x8 tiles (H or W < 8)
            prec = int(rng.choice([3, 4, 6]))
            N = int(rng.integers(1, 3))
            D = int(rng.integers(1, 28))
            Hh, W = int(rng.integers(1, 8)), int(rng.integers(1, 12))
            ci = 16 * int(rng.integers(1, 9))
            co = 128 * int(rng.integers(1, 3))
            x = rnd(N, ci, D, Hh, W)
            w = rnd(co, ci, 3, 3, 3, scale=0.04)
            b = rnd(co)
            A = 1.0 + 0.1 * rnd(N, ci)
            B = 0.1 * rnd(N, ci)
            res = rnd(N, co, D, Hh, W)
            ref = F.conv3d(F.silu(x * A[:, :, None, None, None] + B[:, :, None, None, None]), w, b, padding=1) + res
            out, stats, _ = hc.conv3d([hc.to_ndhwc(x).cuda()], w.cuda(), b.cuda(), (D, Hh, W), aff=(A.cuda(), B.cuda()),
                                      act=H.ACT_SILU, res=hc.to_ndhwc(res).cuda(), res_mode=H.RES_SAME, precision=prec)
            got = hc.to_ncdhw(out.cpu())
            e = rel_err_per_channel(got.numpy(), ref.numpy())
            note("winograd p%d" % prec, e, {3: 6e-6, 4: 4e-3, 6: 2e-2}[prec],
                 "N=%d %dx%dx%d %d->%d" % (N, D, Hh, W, ci, co))
            s = stats.cpu()
            cnt = float(D * Hh * W)
            mean = s[..., 0].sum(-1) / cnt
            e2 = float((mean - got.double().mean(dim=(2, 3, 4))).abs().max() / (got.abs().max() + 1e-30))
            note("winograd stats", e2, 1e-5, "N=%d %dx%dx%d %d->%d" % (N, D, Hh, W, ci, co))
        else:               # ---- pool pre-pass
            N = int(rng.integers(1, 3))
            D, Hh, W = int(rng.integers(1, 9)), int(rng.integers(1, 13)), int(rng.integers(1, 13))
            Cn = 4 * int(rng.integers(1, 33))
            x = rnd(N, Cn, D, 2 * Hh, 2 * W) * 2.0
            A = 1.0 + 0.1 * rnd(N, Cn)
            B = 0.1 * rnd(N, Cn)
            ref = F.avg_pool3d(F.silu(x * A[:, :, None, None, None] + B[:, :, None, None, None]), (1, 2, 2))
            xd = hc.to_ndhwc(x).cuda()
            out = torch.empty(N, D, Hh, W, Cn, dtype=torch.float32, device="cuda")
            Ad, Bd = A.cuda(), B.cuda()
            H.check(lib.ddpm3d_pool_act(H.ptr(xd), H.ptr(Ad), H.ptr(Bd), H.ACT_SILU, 1, N, D, Hh, W, Cn, H.ptr(out), 0,
                                        H.stream()))
            torch.cuda.synchronize()
            e = rel_err_per_channel(hc.to_ncdhw(out.cpu()).numpy(), ref.numpy())
            note("pool_act", e, 3e-6, "N=%d %dx%dx%d C=%d" % (N, D, Hh, W, Cn))
    print("# %d random cases, seed %d: all within their bars.  Worst per kind (error / bar):" % (a.cases, a.seed))
    for k in sorted(worst):
        e, tol, desc = worst[k]
        print("%-16s %.3e / %.1e   %s" % (k, e, tol, desc))


if __name__ == "__main__":
    main()
