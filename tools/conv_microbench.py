#!/usr/bin/env python3
"""
Single-layer microbenchmark of the fused conv3d kernel (for rocprofv3 PMC runs
and A/B comparisons of kernel variants in ONE process).

    python tools/conv_microbench.py --shape 64,64,64 --cin 128 --cout 128 --precision 1 --iters 20

Synthetic data; GroupNorm+SiLU prologue and statistics epilogue enabled, like
the network's 3x3x3 convs.
"""

import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-denoising-diffusion-model_amd"))
import torch  # noqa: E402

from guided_diffusion import _hip as H  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="64,64,64")
    ap.add_argument("--n", type=int, default=1)
    ap.add_argument("--cin", type=int, default=128)
    ap.add_argument("--cout", type=int, default=128)
    ap.add_argument("--k", type=int, default=3)
    ap.add_argument("--precision", type=int, default=1)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-prologue", action="store_true")
    ap.add_argument("--no-stats", action="store_true", help="no GroupNorm partial sums (the network's last conv)")
    a = ap.parse_args()
    D, Hh, W = [int(v) for v in a.shape.split(",")]
    lib = H.load()
    dev = "cuda"
    N = a.n
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(N, D, Hh, W, a.cin, device=dev, generator=g)
    w = torch.randn(a.cout, a.cin, a.k, a.k, a.k, device=dev, generator=g) * 0.02
    b = torch.randn(a.cout, device=dev, generator=g) * 0.02
    A = 1 + 0.1 * torch.randn(N, a.cin, device=dev, generator=g)
    B = 0.1 * torch.randn(N, a.cin, device=dev, generator=g)
    wp = torch.empty(lib.ddpm3d_packed_weight_bytes(a.cout, a.cin, a.k, a.precision), dtype=torch.uint8, device=dev)
    H.check(lib.ddpm3d_pack_conv_weight(H.ptr(w), a.cout, a.cin, a.k, a.precision, H.ptr(wp), H.stream()))
    out = torch.empty(N, D, Hh, W, a.cout, device=dev)
    rows = lib.ddpm3d_conv_stats_rows(N, D, Hh, W, a.cin, a.cout, a.k, a.precision)
    stats = torch.empty(N, a.cout, rows, 2, dtype=torch.float64, device=dev)
    need = lib.ddpm3d_conv_workspace_bytes(N, D, Hh, W, a.cin, a.cout, a.k, a.precision)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dev)
    d = H.ConvDesc()
    d.N, d.D, d.H, d.W, d.Cin, d.Cout, d.ksize, d.in_mode = N, D, Hh, W, a.cin, a.cout, a.k, H.IN_SAME
    d.src0, d.C0 = H.ptr(x), a.cin
    if not a.no_prologue:
        d.aff_a, d.aff_b, d.act = H.ptr(A), H.ptr(B), H.ACT_SILU
    d.precision = a.precision
    d.w_packed, d.bias = H.ptr(wp), H.ptr(b)
    d.out = H.ptr(out)
    if not a.no_stats:
        d.stats, d.stats_rows = H.ptr(stats), rows
    # upper bound of the input as the matrix cores see it (ddpm3d_conv_desc.in_bound)
    xin = x if a.no_prologue else torch.nn.functional.silu(x * A[:, None, None, None, :] + B[:, None, None, None, :])
    bound = xin.abs().reshape(N, -1).amax(dim=1, keepdim=True).contiguous()
    d.in_bound, d.in_bound_count, d.in_bound_stride = H.ptr(bound), 1, 1
    if need:
        d.workspace, d.workspace_bytes = H.ptr(ws), need
    for _ in range(a.warmup):
        H.check(lib.ddpm3d_conv3d(C.byref(d), H.stream()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.iters):
        H.check(lib.ddpm3d_conv3d(C.byref(d), H.stream()))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    fl = 2.0 * N * D * Hh * W * a.cout * a.cin * a.k ** 3
    print("conv %dx%dx%d n=%d %d->%d k%d prec%d split_ws=%d: %.3f ms  %.1f TFLOP/s (algorithmic)"
          % (D, Hh, W, N, a.cin, a.cout, a.k, a.precision, need, ms, fl / ms / 1e9))


if __name__ == "__main__":
    main()
