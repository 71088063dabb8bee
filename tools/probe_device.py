#!/usr/bin/env python3
"""
What this device sustains on the matrix pipes (ddpm3d_mfma_probe, register-only loops on
pseudo-random operands): every MFMA shape the kernels issue or could issue, one and two waves
per SIMD, variants interleaved over rounds in one process (boards differ; rank shapes on ONE).

    python tools/probe_device.py [--ms 60] [--rounds 3] > gpurun_out/probe.txt
"""

import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "3d-denoising-diffusion-model_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

from guided_diffusion import _hip as H  # noqa: E402

KINDS = [("f16 32x32x16", H.PROBE_F16_32X32X16), ("f16 16x16x32", H.PROBE_F16_16X16X32),
         ("bf16 32x32x16", H.PROBE_BF16_32X32X16), ("bf16 16x16x32", H.PROBE_BF16_16X16X32),
         ("f32 32x32x2", H.PROBE_F32_32X32X2)]


def probe(lib, kind, blocks, target_ms, dev):
    """-> (TFLOP/s, in-kernel GHz) of one timed launch of about target_ms."""
    out = torch.empty(blocks * 256, dtype=torch.float32, device=dev)
    clk = torch.zeros(blocks * 2, dtype=torch.int64, device=dev)
    st = H.stream()
    fl_iter = lib.ddpm3d_mfma_probe_flops_per_iter(kind)
    # size the loop from a short calibration launch
    iters = 2000
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    H.check(lib.ddpm3d_mfma_probe(kind, iters, blocks, H.ptr(out), H.ptr(clk), st))
    e0.record()
    H.check(lib.ddpm3d_mfma_probe(kind, iters, blocks, H.ptr(out), H.ptr(clk), st))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    iters = max(1000, int(iters * target_ms / max(ms, 1e-3)))
    e0.record()
    H.check(lib.ddpm3d_mfma_probe(kind, iters, blocks, H.ptr(out), H.ptr(clk), st))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    c = clk.view(blocks, 2).double().cpu()
    ghz = float((c[:, 0] / c[:, 1]).median()) * 0.1
    return fl_iter * iters * blocks / (ms * 1e-3) / 1e12, ghz


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", type=float, default=60.0)
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = H.load()
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    print("# %s, %d CUs; register-only MFMA loops, %g ms each, median of %d interleaved rounds"
          % (torch.cuda.get_device_name(dev), cus, a.ms, a.rounds))
    res = {}
    for _ in range(a.rounds):
        for name, kind in KINDS:
            for wps in (1, 2):
                res.setdefault((name, wps), []).append(probe(lib, kind, cus * wps, a.ms, dev))
    for (name, wps), v in res.items():
        tf = sorted(x[0] for x in v)[len(v) // 2]
        ghz = sorted(x[1] for x in v)[len(v) // 2]
        print("%-14s %d wave(s)/SIMD: %8.1f TFLOP/s   in-kernel clock %.3f GHz" % (name, wps, tf, ghz))


if __name__ == "__main__":
    main()
