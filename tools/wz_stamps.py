#!/usr/bin/env python3
"""
Where the cycles of the dominant kernel go: per-wave phase stamps (s_memtime) of conv3d_wz_kernel
on one layer.  Needs a MEASUREMENT build of the library (never the shipped one):

    mkdir -p scratch/stamps && cp 3d-denoising-diffusion-model_amd/csrc/*.h* 3d-denoising-diffusion-model_amd/csrc/Makefile scratch/stamps/
    make -C scratch/stamps -j8 EXTRA=-DDDPM3D_WZ_STAMPS INC="-I../../include -I."
    DDPM3D_LIB=scratch/stamps/libddpm3d.so python tools/wz_stamps.py > gpurun_out/stamps.txt

Stamps per wave (conv3d_wz.h): 0 entry, 1 prologue done, per chunk c: 2+5c at barrier 1, 3+5c past it,
4+5c staging written, 5+5c past barrier 2, 6+5c tap loop done; 42 output transform, 43 epilogue done;
44 HW_ID, 45 XCC_ID, 46 realtime.
"""

import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-denoising-diffusion-model_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from guided_diffusion import _hip as H  # noqa: E402

NS = 48


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="64,64,64")
    ap.add_argument("--cin", type=int, default=128)
    ap.add_argument("--cout", type=int, default=128)
    ap.add_argument("--precision", type=int, default=3)
    ap.add_argument("--res", action="store_true", help="with a same-shape residual (ResBlock conv2)")
    a = ap.parse_args()
    D, Hh, W = [int(v) for v in a.shape.split(",")]
    lib = H.load()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(1, D, Hh, W, a.cin, device=dev, generator=g)
    w = torch.randn(a.cout, a.cin, 3, 3, 3, device=dev, generator=g) * 0.02
    b = torch.randn(a.cout, device=dev, generator=g) * 0.02
    A = 1 + 0.1 * torch.randn(1, a.cin, device=dev, generator=g)
    B = 0.1 * torch.randn(1, a.cin, device=dev, generator=g)
    wp = torch.empty(lib.ddpm3d_packed_weight_bytes(a.cout, a.cin, 3, a.precision), dtype=torch.uint8, device=dev)
    H.check(lib.ddpm3d_pack_conv_weight(H.ptr(w), a.cout, a.cin, 3, a.precision, H.ptr(wp), H.stream()))
    out = torch.empty(1, D, Hh, W, a.cout, device=dev)
    res = torch.randn(1, D, Hh, W, a.cout, device=dev, generator=g)
    rows = lib.ddpm3d_conv_stats_rows(1, D, Hh, W, a.cin, a.cout, 3, a.precision)
    stats = torch.empty(1, a.cout, rows, 2, dtype=torch.float64, device=dev)
    # a split launch keeps its slabs at the front of the workspace; the kernel dumps the stamps behind them
    need = lib.ddpm3d_conv_workspace_bytes(1, D, Hh, W, a.cin, a.cout, 3, a.precision)
    S = max(1, need // (4 * D * Hh * W * a.cout))
    if Hh >= 8 and W >= 8:
        tiles = ((D + 1) // 2) * ((Hh + 7) // 8) * ((W + 7) // 8)      # 8x8x2 (8x4x4 tiles the same count)
    else:
        tiles = ((D + 7) // 8) * ((Hh + 3) // 4) * ((W + 3) // 4)      # 4x4x8
    nwg = tiles * (a.cout // 128) * S
    assert -(-(a.cin // 16) // S) <= 8, "at most 8 chunks per workgroup fit the stamp record"
    ws = torch.zeros(need // 8 + nwg * 4 * NS, dtype=torch.int64, device=dev)
    stamps = ws[need // 8:]
    d = H.ConvDesc()
    d.N, d.D, d.H, d.W, d.Cin, d.Cout, d.ksize, d.in_mode = 1, D, Hh, W, a.cin, a.cout, 3, H.IN_SAME
    d.src0, d.C0 = H.ptr(x), a.cin
    d.aff_a, d.aff_b, d.act = H.ptr(A), H.ptr(B), H.ACT_SILU
    d.precision = a.precision
    d.w_packed, d.bias = H.ptr(wp), H.ptr(b)
    d.out = H.ptr(out)
    d.stats, d.stats_rows = H.ptr(stats), rows
    if a.res:
        d.res, d.res_mode = H.ptr(res), H.RES_SAME
    xin = torch.nn.functional.silu(x * A[:, None, None, None, :] + B[:, None, None, None, :])
    bound = xin.abs().reshape(1, -1).amax(dim=1, keepdim=True).contiguous()
    d.in_bound, d.in_bound_count, d.in_bound_stride = H.ptr(bound), 1, 1
    d.workspace, d.workspace_bytes = H.ptr(ws), ws.numel() * 8             # slabs (if any) + the stamp dump
    for _ in range(20):                                                    # reach the sustained clock
        H.check(lib.ddpm3d_conv3d(C.byref(d), H.stream()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    H.check(lib.ddpm3d_conv3d(C.byref(d), H.stream()))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    s = stamps.view(nwg, 4, NS).cpu().numpy().astype(np.int64)
    if not s[:, :, 43].all():
        print("no stamps: this library was not built with -DDDPM3D_WZ_STAMPS")
        return
    nch = -(-(a.cin // 16) // S)      # chunks per workgroup
    life = s[:, :, 43] - s[:, :, 0]
    ghz = None
    print("# conv3d_wz_kernel stamps, %d->%d @ %dx%dx%d, precision %d%s: %.3f ms (instrumented; conv + reduce if split), "
          "%d workgroups, split %d, %d chunks each"
          % (a.cin, a.cout, D, Hh, W, a.precision, ", residual" if a.res else "", ms, nwg, S, nch))
    def stat(name, v):
        v = np.asarray(v).ravel()
        print("%-44s median %8.0f  mean %8.0f  p10 %8.0f  p90 %8.0f" % (name, np.median(v), v.mean(),
                                                                        np.percentile(v, 10), np.percentile(v, 90)))
    stat("wave lifetime", life)
    # (s_memtime is per XCD: differences are meaningful within one wave / one XCD only, never across the launch)
    stat("prologue (entry -> first stage issued)", s[:, :, 1] - s[:, :, 0])
    c = np.arange(nch)
    b1 = s[:, :, 3 + 5 * c] - s[:, :, 2 + 5 * c]
    sw = s[:, :, 4 + 5 * c] - s[:, :, 3 + 5 * c]
    b2 = s[:, :, 5 + 5 * c] - s[:, :, 4 + 5 * c]
    tl = s[:, :, 6 + 5 * c] - s[:, :, 5 + 5 * c]
    stat("barrier 1 wait (tile free), chunks >= 1", b1[:, :, 1:])
    stat("barrier 1 wait, chunk 0 (first loads land)", b1[:, :, 0])
    stat("stage_write, waves 0-2 (two slots)", sw[:, :3])
    stat("stage_write, wave 3 (one slot)", sw[:, 3])
    stat("barrier 2 wait (image complete)", b2)
    stat("tap loop (216 MFMAs = 6912 pipe cycles alone)", tl)
    stat("epilogue (transform, residual, stores, sums)", s[:, :, 43] - s[:, :, 42])
    per_chunk = (s[:, :, 6 + 5 * (nch - 1)] - s[:, :, 2]) / nch
    stat("chunk period", per_chunk)
    tot = life.mean()
    parts = {"prologue": (s[:, :, 1] - s[:, :, 0]).mean(), "barrier1": b1.sum(-1).mean(), "stage_write": sw.sum(-1).mean(),
             "barrier2": b2.sum(-1).mean(), "taps": tl.sum(-1).mean(), "epilogue": (s[:, :, 43] - s[:, :, 42]).mean()}
    print("share of a wave's life: " + ", ".join("%s %.1f %%" % (k, 100 * v / tot) for k, v in parts.items()))
    print("MFMA pipe time per wave = %d cycles = %.1f %% of its life (two waves share a SIMD: 50 %% = saturated)"
          % (nch * 216 * 32, 100.0 * nch * 216 * 32 / tot))

    # partner analysis: waves that shared a SIMD (same XCC / SE / CU / SIMD ids, overlapping lifetimes)
    hw = s[:, :, 44]
    xcc = s[:, :, 45] & 0xf
    key = (xcc << 32) | (hw & 0xfff0)        # drop the wave slot bits [3:0]
    flat = [(int(key[i, j]), int(s[i, j, 0]), int(s[i, j, 43]), i, j) for i in range(nwg) for j in range(4)]
    flat.sort()
    both_stage = both_tap = one_tap = 0
    total = 0
    from itertools import groupby
    for k, grp in groupby(flat, key=lambda t: t[0]):
        grp = list(grp)
        ev = []   # (time, kind, +1/-1): kind 0 = in tap loop, 1 = alive
        for _, t0, t1, i, j in grp:
            ev.append((t0, 1, 1)); ev.append((t1, 1, -1))
            for cc in range(nch):
                ev.append((int(s[i, j, 5 + 5 * cc]), 0, 1)); ev.append((int(s[i, j, 6 + 5 * cc]), 0, -1))
        ev.sort()
        ntap = nalive = 0
        last = ev[0][0]
        for t, kind, dlt in ev:
            dt = t - last
            if nalive > 0:
                total += dt
                if ntap >= 2:
                    both_tap += dt
                elif ntap == 1:
                    one_tap += dt
                else:
                    both_stage += dt
            last = t
            if kind == 0:
                ntap += dlt
            else:
                nalive += dlt
    if total:
        print("per SIMD, while at least one wave is resident: >= 2 waves in their tap loops %.1f %%, exactly one %.1f %%, "
              "none (pipe has no MFMA work) %.1f %%" % (100 * both_tap / total, 100 * one_tap / total, 100 * both_stage / total))


if __name__ == "__main__":
    main()
