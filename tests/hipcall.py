"""Thin test helpers that call the C ABI (include/ddpm3d.h) on torch device tensors."""

import ctypes as C

import torch

from guided_diffusion import _hip as H


def to_ndhwc(x):
    """(N,C,D,H,W) -> contiguous (N,D,H,W,C)"""
    return x.permute(0, 2, 3, 4, 1).contiguous()


def to_ncdhw(x):
    return x.permute(0, 4, 1, 2, 3).contiguous()


def pack(w, precision=0):
    lib = H.load()
    co, ci, k = w.shape[0], w.shape[1], w.shape[2]
    out = torch.empty(lib.ddpm3d_packed_weight_bytes(co, ci, k, precision), dtype=torch.uint8, device=w.device)
    wc = w.contiguous()
    H.check(lib.ddpm3d_pack_conv_weight(H.ptr(wc), co, ci, k, precision, H.ptr(out), H.stream()))
    torch.cuda.synchronize()
    return out


def conv3d(srcs, w, b, out_dhw, in_mode=H.IN_SAME, aff=None, act=H.ACT_NONE, res=None, res_mode=H.RES_NONE,
           out_layout=H.OUT_NDHWC, want_stats=True, planar=False, bias_stride_n=0, precision=0, hint=0,
           bound=None, out_bf16=False, out_f16=False):
    """srcs: list of NDHWC device tensors (or two (N,1,D,H,W) volumes when planar).
    bound: [N, k] device tensor of upper bounds of the input as the matrix cores see it
    (ddpm3d_conv_desc.in_bound); default = the exact maximum of |act(A*x + B)| per sample.
    Returns (out, stats, rows)."""
    lib = H.load()
    dev = w.device
    co, ci, k = w.shape[0], w.shape[1], w.shape[2]
    N = srcs[0].shape[0]
    D, Hh, W = out_dhw
    d = H.ConvDesc()
    d.N, d.D, d.H, d.W, d.Cout, d.ksize = N, D, Hh, W, co, k
    if planar:
        d.in_mode, d.Cin, d.C0, d.C1 = H.IN_PLANAR2, 2, 1, 1
        d.src0, d.src1 = H.ptr(srcs[0]), H.ptr(srcs[1])
    else:
        d.in_mode = in_mode
        d.src0, d.C0 = H.ptr(srcs[0]), srcs[0].shape[-1]
        if len(srcs) > 1:
            d.src1, d.C1 = H.ptr(srcs[1]), srcs[1].shape[-1]
        d.Cin = d.C0 + d.C1
        # 16-bit tensors are recognised by their dtype (ddpm3d_conv_desc.io_dtype); f16 ones set the
        # descriptor-wide IO_HALF_IS_F16 (a call mixes fp32 with ONE 16-bit type)
        half = (torch.bfloat16, torch.float16)
        d.io_dtype |= H.IO_SRC0_BF16 if srcs[0].dtype in half else 0
        d.io_dtype |= H.IO_SRC1_BF16 if (len(srcs) > 1 and srcs[1].dtype in half) else 0
        if any(s_.dtype == torch.float16 for s_ in srcs):
            d.io_dtype |= H.IO_HALF_IS_F16
    assert d.Cin == ci
    if aff is not None:
        d.aff_a, d.aff_b = H.ptr(aff[0]), H.ptr(aff[1])
    d.act = act
    d.precision = precision
    d.kernel_hint = hint
    if bound is None:
        with torch.no_grad():
            if planar:
                xin = torch.stack([srcs[0].reshape(N, -1), srcs[1].reshape(N, -1)], dim=-1)
            else:
                xin = torch.cat([s.float().reshape(N, -1, s.shape[-1]) for s in srcs], dim=-1)
            if aff is not None:
                xin = xin * aff[0].reshape(N, 1, -1) + aff[1].reshape(N, 1, -1)
                if act == H.ACT_SILU:
                    xin = torch.nn.functional.silu(xin)
            bound = xin.abs().reshape(N, -1).amax(dim=1, keepdim=True).contiguous()
    bound = bound.float().contiguous()
    d.in_bound, d.in_bound_count, d.in_bound_stride = H.ptr(bound), bound.shape[1], 1
    wp = pack(w, precision)
    d.w_packed, d.bias, d.bias_stride_n = H.ptr(wp), H.ptr(b), bias_stride_n
    d.res_mode, d.res = res_mode, H.ptr(res)
    if res is not None and res.dtype in (torch.bfloat16, torch.float16):
        d.io_dtype |= H.IO_RES_BF16 | (H.IO_HALF_IS_F16 if res.dtype == torch.float16 else 0)
    if out_bf16 or out_f16:
        d.io_dtype |= H.IO_OUT_BF16 | (H.IO_HALF_IS_F16 if out_f16 else 0)
    if out_layout == H.OUT_NDHWC:
        odt = torch.bfloat16 if out_bf16 else (torch.float16 if out_f16 else torch.float32)
        out = torch.full((N, D, Hh, W, co), float("nan"), dtype=odt, device=dev)
    else:
        out = torch.full((N, co, D, Hh, W), float("nan"), dtype=torch.float32, device=dev)
    d.out, d.out_layout = H.ptr(out), out_layout
    # how the library will run THIS descriptor (hints included): statistics rows, scratch, split over Cin
    rows, need, split = H.conv_plan(d)
    if not hint and not planar:
        assert rows == lib.ddpm3d_conv_stats_rows(N, D, Hh, W, ci, co, k, d.precision)
        assert need == lib.ddpm3d_conv_workspace_bytes(N, D, Hh, W, ci, co, k, d.precision)
    assert need == (split * N * D * Hh * W * co * 4 if split > 1 else 0)
    # the slabs start out poisoned: one that was never written shows up as NaN in the output
    ws = torch.full((max(need, 16) // 4,), POISON, dtype=torch.float32, device=dev)
    if need:
        d.workspace, d.workspace_bytes = H.ptr(ws), need
    stats = None
    if want_stats and out_layout == H.OUT_NDHWC:
        stats = torch.full((N, co, rows, 2), float("nan"), dtype=torch.float64, device=dev)
        d.stats, d.stats_rows = H.ptr(stats), rows
    H.check(lib.ddpm3d_conv3d(C.byref(d), H.stream()))
    torch.cuda.synchronize()
    LAST_PLAN.update(split=split, rows=rows)
    return out, stats, rows


POISON = float("nan")   # what a split conv's workspace holds before the launch
LAST_PLAN = {}     # what the last conv3d() call ran as (tests assert that a path was really taken)


def gn_finalize(stats_list, count, gamma, beta, film=None, film_stride=0, film_off=0, groups=32,
                with_bound=False):
    lib = H.load()
    s0 = stats_list[0]
    s1 = stats_list[1] if len(stats_list) > 1 else None
    N = s0.shape[0]
    Cn = s0.shape[1] + (s1.shape[1] if s1 is not None else 0)
    A = torch.empty(N, Cn, dtype=torch.float32, device=s0.device)
    B = torch.empty(N, Cn, dtype=torch.float32, device=s0.device)
    bound = torch.full((N, groups, 2), float("nan"), dtype=torch.float32, device=s0.device)
    H.check(lib.ddpm3d_gn_finalize(H.ptr(s0), s0.shape[1], s0.shape[2],
                                   H.ptr(s1), s1.shape[1] if s1 is not None else 0,
                                   s1.shape[2] if s1 is not None else 0,
                                   N, groups, float(count), 1e-5, H.ptr(gamma), H.ptr(beta),
                                   H.ptr(film), film_stride, film_off, H.ptr(A), H.ptr(B), H.ptr(bound),
                                   H.stream()))
    torch.cuda.synchronize()
    return (A, B, bound) if with_bound else (A, B)


def gn_stats(x_ndhwc):
    lib = H.load()
    N = x_ndhwc.shape[0]
    Cn = x_ndhwc.shape[-1]
    vox = x_ndhwc[0].numel() // Cn
    rows = lib.ddpm3d_gn_stats_rows(vox)
    st = torch.full((N, Cn, rows, 2), float("nan"), dtype=torch.float64, device=x_ndhwc.device)
    H.check(lib.ddpm3d_gn_stats(H.ptr(x_ndhwc), N, vox, Cn, H.ptr(st), H.stream()))
    torch.cuda.synchronize()
    return st
