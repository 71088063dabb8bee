#!/usr/bin/env python3
"""
A/B of kernel forms / launch orders of identical arithmetic, layer by layer, in ONE process
(ddpm3d_conv_desc.kernel_hint): every distinct Winograd-eligible layer shape of the published
architecture, variants interleaved over --rounds rounds.

    python tools/layer_ab.py [--size 64] [--precision f16x3] [--only64] > gpurun_out/layer_ab.txt
"""

import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "3d-denoising-diffusion-model_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

import bench  # noqa: E402
from guided_diffusion import _hip as H  # noqa: E402
from guided_diffusion import synth  # noqa: E402

IN_MODES = {H.IN_SAME: "same", H.IN_POOL: "pool", H.IN_UP: "up", H.IN_PLANAR2: "planar"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "f16", "bf16"])
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--only64", action="store_true", help="only the full-resolution layers")
    ap.add_argument("--what", default="order", choices=["order", "issue"],
                    help="order: workgroup -> XCD orders; issue: issue orders of a tap (f16x3 Winograd-D kernel)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model, _, _ = bench.build_model(bench.PUBLISHED, "250", dev)
    model.conv_precision = a.precision
    S, B = a.size, a.batch
    shape = (B, 1, S, S, S)
    x = torch.from_numpy(synth.synth_noise(shape, 1, seed=3)[0]).to(dev)
    lr = torch.from_numpy(synth.synth_low_res(shape, seed=1234)).to(dev)
    t = torch.full((B,), 617, dtype=torch.long, device=dev)
    lib = H.load()
    with torch.no_grad():
        model(x, t, low_res=lr)                       # builds the plan, fills every buffer
    torch.cuda.synchronize()
    plan = model.engine().plan(B, S, S, S)
    ap_what = a.what
    variants = {"order": [("default", 0), ("wstat_off", H.HINT_WSTAT_OFF), ("wstat_on", H.HINT_WSTAT_ON)],
                # issue orders of a tap in the f16x3 Winograd-D kernel (conv3d_wz.h: IL)
                "issue": [("default", 0)] + [("il%d" % il, (il + 1) << H.HINT_WZ_ORDER_SHIFT) for il in (0, 1, 2, 4)]}[ap_what]
    seen = {}
    print("# published architecture, %dx1x%d^3, %s; ms per launch (median of %d rounds x %d launches)"
          % (B, S, a.precision, a.rounds, a.iters))
    print("%-6s %-12s %-12s %2s | %s" % ("input", "Cin->Cout", "DxHxW", "S", "  ".join("%9s" % n for n, _ in variants)))
    for i in sorted(plan.conv_meta):
        tag, fl = plan.conv_meta[i]
        if not any(("_p%d_" % k) in tag for k in (3, 4, 6)):
            continue
        d = plan.steps[i][1][0]._obj
        if a.only64 and d.H != S:
            continue
        key = (d.in_mode, d.Cin, d.Cout, d.D, d.H, d.W, d.res_mode)
        if key in seen:
            continue
        seen[key] = True
        st = H.stream()
        times = {n: [] for n, _ in variants}
        for _ in range(a.rounds):
            for n, hint in variants:
                d.kernel_hint = hint
                H.check(lib.ddpm3d_conv3d(C.byref(d), st))     # warm
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _k in range(a.iters):
                    H.check(lib.ddpm3d_conv3d(C.byref(d), st))
                e1.record()
                torch.cuda.synchronize()
                times[n].append(e0.elapsed_time(e1) / a.iters)
        d.kernel_hint = 0
        med = {n: sorted(v)[len(v) // 2] for n, v in times.items()}
        ws = lib.ddpm3d_conv_workspace_bytes(d.N, d.D, d.H, d.W, d.Cin, d.Cout, d.ksize, d.precision)
        split = ws // (d.N * d.D * d.H * d.W * d.Cout * 4) if ws else 1
        best = min(med, key=med.get)
        print("%-6s %-12s %-12s %2d | %s   best %s (%.0f TFLOP/s)" % (
            IN_MODES.get(d.in_mode, "?"), "%d->%d" % (d.Cin, d.Cout), "%dx%dx%d" % (d.D, d.H, d.W), split,
            "  ".join("%9.4f" % med[n] for n, _ in variants), best, fl / med[best] / 1e9))


if __name__ == "__main__":
    main()
