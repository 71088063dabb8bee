#!/usr/bin/env python3
"""
Time UNet forwards of the published architecture with whatever library DDPM3D_LIB names (same-box
A/B of two builds: run alternately, tools/lib_ab.sh).  Prints one JSON line: ms per forward (HIP
events around K back-to-back forwards) and the per-kernel-family table of one instrumented forward.

    DDPM3D_LIB=scratch/prevlib/libddpm3d.so python tools/forward_time.py --precision f16x3 --iters 30
"""

import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "3d-denoising-diffusion-model_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

import bench  # noqa: E402
from guided_diffusion import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="f16x3")
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--tag", default="")
    ap.add_argument("--graph", action="store_true", help="replay the captured hipGraph of the forward (engine step_graph)")
    ap.add_argument("--native", action="store_true", help="run on the C-level plan (ddpm3d_unet_forward)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model, _, _ = bench.build_model(bench.PUBLISHED, "250", dev)
    model.conv_precision = a.precision
    model.step_graph = a.graph
    model.native_plan = a.native
    S, B = a.size, a.batch
    shape = (B, 1, S, S, S)
    x = torch.from_numpy(synth.synth_noise(shape, 1, seed=3)[0]).to(dev)
    lr = torch.from_numpy(synth.synth_low_res(shape, seed=1234)).to(dev)
    t = torch.full((B,), 617, dtype=torch.long, device=dev)
    eng = model.engine()
    rows = eng.film_rows(t.float())
    with torch.no_grad():
        for _ in range(a.warmup):
            eng.forward(x, lr, rows, eng.film_total)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.iters):
            eng.forward(x, lr, rows, eng.film_total)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        eng.native_plan = False                    # the per-family table comes from the instrumented Python plan
        plan = eng.plan(B, S, S, S)
        fam = {}
        for _ in range(3):
            plan.timing = []
            eng.forward(x, lr, rows, eng.film_total)
            torch.cuda.synchronize()
            for tag, fl, a0, a1 in plan.timing:
                f = fam.setdefault(tag, [0, 0.0])
                f[0] += 1
                f[1] += a0.elapsed_time(a1)
            plan.timing = None
    print(json.dumps({"tag": a.tag, "lib": os.environ.get("DDPM3D_LIB", "in-tree"), "precision": a.precision, "graph": a.graph, "native": a.native,
                      "ms_per_forward": round(ms, 4),
                      "families_ms": {k: round(v[1] / 3, 4) for k, v in sorted(fam.items())}}))


if __name__ == "__main__":
    main()
