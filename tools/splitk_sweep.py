#!/usr/bin/env python3
"""
Calibration of the split-over-Cin rule (conv3d_params.h: ddpm3d_conv_cfg): every low-resolution
3x3x3 layer shape of the published network timed at forced split factors (kernel_hint bits 16..21;
no statistics), conv + reduce launches together, in the f16x3 Winograd-D arithmetic.

    python tools/splitk_sweep.py > gpurun_out/splitk_sweep.txt
    python tools/splitk_sweep.py --ksize 1 [--precision 1|2|5] > gpurun_out/splitk_sweep_1x1.txt
        the skip connections' 1x1 convs (raw input, conv1x1.hip) -- every shape the network holds
"""

import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-denoising-diffusion-model_amd"))
import torch  # noqa: E402

from guided_diffusion import _hip as H  # noqa: E402

SHAPES = [  # (Cin, Cout, D, H, W) of the published network's split levels
    (128, 128, 64, 32, 32), (256, 128, 64, 32, 32),
    (128, 128, 64, 16, 16), (128, 256, 64, 16, 16), (256, 256, 64, 16, 16), (512, 256, 64, 16, 16),
    (512, 128, 64, 16, 16), (256, 128, 64, 16, 16),
    (256, 256, 64, 8, 8), (256, 384, 64, 8, 8), (384, 384, 64, 8, 8), (768, 384, 64, 8, 8), (768, 256, 64, 8, 8),
    (512, 256, 64, 8, 8),
    # the 64x4x4 level (4x4x8 tiles since r03)
    (384, 384, 64, 4, 4), (384, 512, 64, 4, 4), (512, 512, 64, 4, 4), (1024, 512, 64, 4, 4), (1024, 384, 64, 4, 4),
    (768, 384, 64, 4, 4),
]
SHAPES_1X1 = [  # the skip connections (unet.py:173-186) of the published network
    (256, 128, 64, 64, 64), (256, 128, 64, 32, 32),
    (128, 256, 64, 16, 16), (512, 256, 64, 16, 16), (512, 128, 64, 16, 16), (256, 128, 64, 16, 16),
    (256, 384, 64, 8, 8), (768, 384, 64, 8, 8), (768, 256, 64, 8, 8), (512, 256, 64, 8, 8),
    (384, 512, 64, 4, 4), (1024, 512, 64, 4, 4), (1024, 384, 64, 4, 4), (768, 384, 64, 4, 4),
]
SPLITS = [1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 24, 32]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ksize", type=int, default=3)
    ap.add_argument("--precision", type=int, default=None, help="C ABI precision code (default 3 for ksize 3, 1 for ksize 1)")
    a = ap.parse_args()
    ks = a.ksize
    prec = a.precision if a.precision is not None else (3 if ks == 3 else 1)
    lib = H.load()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    print("# %s conv (precision code %d) + split-K reduce, ms per layer at forced split factors (median of 5 x 10 launches);"
          % ("f16x3 Winograd-D" if ks == 3 and prec == 3 else "%dx%dx%d" % (ks, ks, ks), prec))
    print("# 'auto' = the library's choice with the same descriptor%s" % (" (with GroupNorm partial sums)" if ks == 3 else ""))
    print("%-22s %5s | %s" % ("Cin->Cout @ DxHxW", "auto", "  ".join("S=%-5d" % s for s in SPLITS)))
    for ci, co, D, Hh, W in (SHAPES if ks == 3 else SHAPES_1X1):
        x = torch.randn(1, D, Hh, W, ci, device=dev, generator=g)
        w = torch.randn(co, ci, ks, ks, ks, device=dev, generator=g) * 0.02
        b = torch.randn(co, device=dev, generator=g) * 0.02
        A = 1 + 0.1 * torch.randn(1, ci, device=dev, generator=g)
        B = 0.1 * torch.randn(1, ci, device=dev, generator=g)
        wp = torch.empty(lib.ddpm3d_packed_weight_bytes(co, ci, ks, prec), dtype=torch.uint8, device=dev)
        H.check(lib.ddpm3d_pack_conv_weight(H.ptr(w), co, ci, ks, prec, H.ptr(wp), H.stream()))
        out = torch.empty(1, D, Hh, W, co, device=dev)
        ws = torch.empty(max(SPLITS) * out.numel() * 4, dtype=torch.uint8, device=dev)
        stats = torch.empty(co * (D * Hh * W // 4 + 1) * 2, dtype=torch.float64, device=dev) if ks == 3 else None
        bound = torch.full((1, 1), 8.0, device=dev)
        d = H.ConvDesc()
        d.N, d.D, d.H, d.W, d.Cin, d.Cout, d.ksize, d.in_mode = 1, D, Hh, W, ci, co, ks, H.IN_SAME
        d.src0, d.C0 = H.ptr(x), ci
        if ks == 3:
            d.aff_a, d.aff_b, d.act = H.ptr(A), H.ptr(B), H.ACT_SILU
        d.precision = prec
        d.w_packed, d.bias, d.out = H.ptr(wp), H.ptr(b), H.ptr(out)
        d.in_bound, d.in_bound_count, d.in_bound_stride = H.ptr(bound), 1, 1
        d.workspace, d.workspace_bytes = H.ptr(ws), ws.numel()

        def run(hint):
            d.kernel_hint = hint
            d.stats, d.stats_rows = 0, 0
            if stats is not None:       # the network's 3x3x3 convs all emit GroupNorm partial sums
                d.stats, d.stats_rows = H.ptr(stats), H.conv_plan(d)[0]
            for _ in range(3):
                H.check(lib.ddpm3d_conv3d(C.byref(d), H.stream()))
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _k in range(10):
                    H.check(lib.ddpm3d_conv3d(C.byref(d), H.stream()))
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 10)
            return sorted(ts)[2]

        auto = run(0)
        nch = ci // 16
        cells = []
        for s in SPLITS:
            cells.append("%7.4f" % run(s << H.HINT_SPLITK_SHIFT) if s <= nch and (s == 1 or (nch + s - 1) // s >= 1) else "      -")
        d.kernel_hint = 0
        d.stats, d.stats_rows = 0, 0
        s_auto = H.conv_plan(d)[2]
        print("%-22s %5.4f (S=%d) | %s" % ("%d->%d @ %dx%dx%d" % (ci, co, D, Hh, W), auto, s_auto, "  ".join(cells)))


if __name__ == "__main__":
    main()
