#!/usr/bin/env python3
"""
Phase stamps (s_memtime) of the 1x1x1 conv kernel (conv1x1.hip) on one unsplit layer.  Needs a MEASUREMENT build:

    make -C scratch/stamps -j8 EXTRA="-DDDPM3D_WZ_STAMPS -DDDPM3D_PW_STAMPS" INC="-I../../include -I."
    DDPM3D_LIB=scratch/stamps/libddpm3d.so python tools/pw_stamps.py --shape 64,16,16 --cin 256 --cout 128

Stamps per wave: 0 entry, 1 first blocks' loads issued, 2 first block group done, 3 K loop done, 4 epilogue done.
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="64,16,16")
    ap.add_argument("--cin", type=int, default=256)
    ap.add_argument("--cout", type=int, default=128)
    ap.add_argument("--precision", type=int, default=1)
    ap.add_argument("--half", action="store_true", help="16-bit source / output tensors (the bf16 mode's residual stream)")
    a = ap.parse_args()
    D, Hh, W = [int(v) for v in a.shape.split(",")]
    lib = H.load()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(1, D, Hh, W, a.cin, device=dev, generator=g)
    w = torch.randn(a.cout, a.cin, 1, 1, 1, device=dev, generator=g) * 0.05
    b = torch.randn(a.cout, device=dev, generator=g) * 0.02
    wp = torch.empty(lib.ddpm3d_packed_weight_bytes(a.cout, a.cin, 1, a.precision), dtype=torch.uint8, device=dev)
    H.check(lib.ddpm3d_pack_conv_weight(H.ptr(w), a.cout, a.cin, 1, a.precision, H.ptr(wp), H.stream()))
    odt = torch.bfloat16 if a.half else torch.float32
    xs = x.to(odt).contiguous()
    out = torch.empty(1, D, Hh, W, a.cout, device=dev, dtype=odt)
    tiles = (((D + 1) // 2) * ((Hh + 7) // 8) * ((W + 7) // 8)) if (Hh >= 8 and W >= 8) else (((D + 7) // 8) * ((Hh + 3) // 4) * ((W + 3) // 4))
    nwg = tiles * (a.cout // 128)
    ws = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device=dev)
    d = H.ConvDesc()
    d.N, d.D, d.H, d.W, d.Cin, d.Cout, d.ksize, d.in_mode = 1, D, Hh, W, a.cin, a.cout, 1, H.IN_SAME
    d.src0, d.C0 = H.ptr(xs), a.cin
    d.precision = a.precision
    d.w_packed, d.bias = H.ptr(wp), H.ptr(b)
    d.out = H.ptr(out)
    if a.half:
        d.io_dtype = H.IO_SRC0_BF16 | H.IO_OUT_BF16
    bound = x.abs().reshape(1, -1).amax(dim=1, keepdim=True).contiguous()
    d.in_bound, d.in_bound_count, d.in_bound_stride = H.ptr(bound), 1, 1
    d.kernel_hint = 1 << 16                                               # forced split factor 1 (DDPM3D_HINT_SPLITK_SHIFT)
    d.workspace, d.workspace_bytes = H.ptr(ws), ws.numel() * 8
    name = C.create_string_buffer(64)
    H.check(lib.ddpm3d_conv_kernel_family(C.byref(d), name, 64))
    for _ in range(20):
        H.check(lib.ddpm3d_conv3d(C.byref(d), H.stream()))
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        H.check(lib.ddpm3d_conv3d(C.byref(d), H.stream()))
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    s = ws.view(nwg, 4, 8).cpu().numpy().astype(np.int64)
    if not s[:, :, 4].all():
        print("no stamps: this library was not built with -DDDPM3D_PW_STAMPS (family %s)" % name.value.decode())
        return
    print("# %s stamps, %d->%d @ %dx%dx%d%s: %.1f us between events (median of 10), %d workgroups, %d blocks of 32 channels"
          % (name.value.decode(), a.cin, a.cout, D, Hh, W, ", 16-bit tensors" if a.half else "", 1e3 * float(np.median(ts)), nwg, a.cin // 32))

    def stat(nm, v):
        v = np.asarray(v).ravel()
        print("%-44s median %8.0f  mean %8.0f  p10 %8.0f  p90 %8.0f" % (nm, np.median(v), v.mean(), np.percentile(v, 10), np.percentile(v, 90)))
    stat("wave lifetime", s[:, :, 4] - s[:, :, 0])
    stat("prologue (entry -> first loads issued)", s[:, :, 1] - s[:, :, 0])
    stat("first block group (loads land, MFMAs)", s[:, :, 2] - s[:, :, 1])
    stat("remaining blocks", s[:, :, 3] - s[:, :, 2])
    stat("epilogue", s[:, :, 4] - s[:, :, 3])
    rt = s[:, :, 5]
    print("launch span (s_memrealtime, 100 MHz): first wave end -> last wave end %.2f us; entry ticks are per XCD"
          % ((rt.max() - rt.min()) / 100.0))
    life_us = None
    # s_memtime runs at the constant 100 MHz reference on gfx950 when read this way? report the ratio to realtime if sensible
    print("wave lifetime / kernel time between events: %.3f ticks per ns" % (np.median(s[:, :, 4] - s[:, :, 0]) / (1e6 * float(np.median(ts)))))


if __name__ == "__main__":
    main()
