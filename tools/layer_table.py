#!/usr/bin/env python3
"""
Per-launch table of one UNet forward of the published architecture (BASELINE config 2 shape):
layer shape, input mode, kernel family, split-K factor, time (HIP events, averaged over --reps
instrumented forwards) and algorithmic TFLOP/s.  Run on the GPU box:

    python tools/layer_table.py [--size 64] [--batch 1] [--precision f16x3] > gpurun_out/layers.txt

The events bracket each conv3d / attention call of the launch plan (a split conv's time includes
its reduce kernel); GroupNorm finalizes and the sampler update are not listed.
"""

import argparse
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
    ap.add_argument("--precision", default="f16x3", choices=["f32", "f16x3", "f16", "bf16"])
    ap.add_argument("--reps", type=int, default=5)
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
        model(x, t, low_res=lr)                       # builds the plan, warms up
        plan = model.engine().plan(B, S, S, S)
        runs = []
        for _ in range(a.reps):
            plan.timing = []
            model(x, t, low_res=lr)
            torch.cuda.synchronize()
            runs.append([e0.elapsed_time(e1) for _, _, e0, e1 in plan.timing])
            plan.timing = None
    steps = sorted(plan.conv_meta)
    assert all(len(r) == len(steps) for r in runs)
    ms = [sum(r[i] for r in runs) / len(runs) for i in range(len(steps))]
    print("# published architecture, %dx1x%d^3, conv arithmetic %s, %d instrumented forwards" % (B, S, a.precision, a.reps))
    print("%3s  %-24s %-6s %-12s %-14s %2s  %8s %8s" % ("#", "kernel family", "input", "Cin->Cout", "D x H x W", "S", "ms", "TFLOP/s"))
    tot = 0.0
    for j, i in enumerate(steps):
        tag, fl = plan.conv_meta[i]
        fn, args = plan.steps[i]
        tot += ms[j]
        if tag == "pool_act":                                  # ddpm3d_pool_act(src, A, B, act, fast, N, D, H, W, C, ...)
            print("%3d  %-24s %-6s %-12s %-14s %2s  %8.3f %8s" % (j, tag, "pool", "%d" % args[9], "%dx%dx%d" % (args[6], args[7], args[8]),
                                                                  "-", ms[j], "-"))
            continue
        if tag.startswith("attention"):
            print("%3d  %-24s %-6s %-12s %-14s %2s  %8.3f %8.1f" % (j, tag, "-", "-", "T=%d" % args[2], "-", ms[j], fl / ms[j] / 1e9))
            continue
        d = args[0]._obj
        vox = d.D * d.H * d.W
        ws = lib.ddpm3d_conv_workspace_bytes(d.N, d.D, d.H, d.W, d.Cin, d.Cout, d.ksize, d.precision)
        split = ws // (d.N * vox * d.Cout * 4) if ws else 1
        print("%3d  %-24s %-6s %-12s %-14s %2d  %8.3f %8.1f" % (
            j, tag, IN_MODES.get(d.in_mode, "?"), "%d->%d" % (d.Cin, d.Cout), "%dx%dx%d" % (d.D, d.H, d.W),
            split, ms[j], fl / ms[j] / 1e9))
    fl_tot = sum(f for _, f in plan.conv_meta.values())
    print("# total %.3f ms, %.1f GFLOP, %.1f TFLOP/s" % (tot, fl_tot / 1e9, fl_tot / tot / 1e9))


if __name__ == "__main__":
    main()
