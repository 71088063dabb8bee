"""
CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

A plain restatement (numpy fp64 for the schedule tables, torch-CPU fp32
functional ops for the network and the sampler) of the reference's sampling
hot path, written from the behaviour of

    guided_diffusion/gaussian_diffusion.py:18-42, 118-169, 232-333, 395-439,
        441-535, 537-585, 625-707, 897-910
    guided_diffusion/respace.py:7-60, 72-86, 123-128
    guided_diffusion/unet.py:81-140, 143-256, 259-354, 720-1044, 1676-1694
    guided_diffusion/nn.py:17-19, 93-121

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package, and only as the checker / the timed CPU baseline.  The
product path (3d-denoising-diffusion-model_amd/) never imports it and has no
CPU fallback.

Parity pin: the reference holds no tests, golden vectors or fixtures of its
own for this path, so the oracle is pinned against
outputs of the reference itself, produced in the build container by
tests/golden/make_golden.py (which imports /root/reference) and committed as
tests/golden/*.npz / *.json.  tests/test_oracle_golden.py checks every one of
them.
"""
