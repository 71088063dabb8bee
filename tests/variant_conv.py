"""
Helper of test_gpu_ops.py::test_conv3d_variants_agree_bitwise: run as a subprocess so that the
library's launch-time switches (DDPM3D_WZ2, DDPM3D_WZ_DB, DDPM3D_WSTAT -- read once per process)
can differ between runs.  Prints one line per case: sha256 of the output bytes and of the
GroupNorm partial sums (rows sorted per (n, c): variants may number their rows differently).
"""

import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "3d-denoising-diffusion-model_amd"))

import hipcall as hc  # noqa: E402
from guided_diffusion import _hip as H  # noqa: E402


def rnd(*shape, seed=0, scale=1.0):
    g = np.random.default_rng(seed)
    return torch.from_numpy((g.standard_normal(shape) * scale).astype(np.float32))


def main():
    cases = [
        (1, 8, 16, 16, 64, 128),     # D % 4 == 0: the 8x8x4-tile kernel is eligible
        (2, 4, 8, 24, 32, 256),      # batch 2, two cout blocks
        (1, 64, 8, 8, 256, 384),     # split-K + reduce kernel, weights outweigh activations
        (1, 5, 16, 24, 32, 128),     # odd D: three z-pairs, the last half valid; z-walk groups of 2 + 1
        (1, 1, 9, 12, 16, 128),      # D = 1, ragged H / W: edge tiles take the general epilogue
        (2, 10, 8, 8, 48, 128),      # five z-pairs in one tile column, batch 2
    ]
    for i, (N, D, Hh, W, ci, co) in enumerate(cases):
        x = hc.to_ndhwc(rnd(N, ci, D, Hh, W, seed=10 + i)).cuda()
        w = rnd(co, ci, 3, 3, 3, seed=20 + i, scale=0.05).cuda()
        b = rnd(co, seed=30 + i).cuda()
        A = (1.0 + 0.1 * rnd(N, ci, seed=40 + i)).cuda()
        B = (0.1 * rnd(N, ci, seed=50 + i)).cuda()
        res = hc.to_ndhwc(rnd(N, co, D, Hh, W, seed=60 + i)).cuda()
        out, stats, _ = hc.conv3d([x], w, b, (D, Hh, W), aff=(A, B), act=H.ACT_SILU, res=res,
                                  res_mode=H.RES_SAME, precision=3)
        torch.cuda.synchronize()
        o = out.cpu().numpy()
        s = np.sort(stats.cpu().numpy(), axis=2)
        assert np.isfinite(o).all()
        print("case%d %s %s" % (i, hashlib.sha256(o.tobytes()).hexdigest(), hashlib.sha256(s.tobytes()).hexdigest()))


if __name__ == "__main__":
    main()
