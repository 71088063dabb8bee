#!/usr/bin/env python3
"""
bench.py's cpu_baseline at several thread counts on this box's host cores (the box shows 256
logical CPUs and grants a share of them): one warm-up + N timed oracle steps per count, with what the cgroup grants.

    python tools/cpu_threads_sweep.py --threads 8,16,32,64 --steps 2 > gpurun_out/cpu_threads_sweep.txt
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
from guided_diffusion import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", default="8,16,32,64")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--size", type=int, default=64)
    a = ap.parse_args()
    arch = dict(bench.PUBLISHED)
    from guided_diffusion import script_util as su
    flags = su.sr_model_and_diffusion_defaults()
    flags.update(arch, timestep_respacing="250")
    # parameter shapes without touching a GPU: the host-side module tree only
    model, _ = su.sr_create_model_and_diffusion(**flags)
    sd = {k: torch.from_numpy(synth.synth_param(k, tuple(v.shape), 0)) for k, v in model.state_dict().items()}
    model_s, phys, aff = bench.host_cpu()
    print("# CPU oracle (oracle/, torch fp32) on %s: %d physical cores, %d visible, cgroup quota %s cores"
          % (model_s, phys, aff, bench.cpu_quota_cores()))
    print("# threads   s/step   volumes/s (x250 steps)")
    for t in [int(v) for v in a.threads.split(",")]:
        rec, _ = bench.cpu_baseline(arch, sd, a.size, "250", a.steps, t)
        per_step = 1.0 / (rec["value"] * 250)
        print("%8d %8.2f   %.3e" % (t, per_step, rec["value"]), flush=True)


if __name__ == "__main__":
    main()
