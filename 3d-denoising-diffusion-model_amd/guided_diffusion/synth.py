"""
Synthetic weights and inputs for benchmarks and parity tests.

The reference ships no checkpoint and no sample data (README.md:57-58), and a
freshly constructed reference network outputs exactly zero because every
ResBlock's second conv, every attention proj_out and the final conv are
zero-initialised (nn.py:68-74, unet.py:210-212, :294, :996).  Every parity
test and every benchmark therefore re-initialises ALL parameters from a fixed
recipe.  The recipe is pure numpy (PCG64, stable across numpy versions) and is
keyed by the state_dict key, so the golden-vector generator, the CPU oracle and
the HIP engine all rebuild bit-identical weights from (key, shape, seed) alone
and no weight file has to be committed.
"""

import zlib

import numpy as np


def _rng(key, seed):
    return np.random.default_rng([zlib.crc32(key.encode("utf-8")), int(seed)])


def synth_param(key, shape, seed=0):
    """One parameter as float32 numpy, from its state_dict key and shape.

    GroupNorm affine (the 1-D ``in_layers.0`` / ``out_layers.0`` / ``out.0`` /
    ``norm`` tensors): gamma = 1 + 0.1 n, beta = 0.1 n.  Everything else
    (conv / linear weights AND biases, including the reference's
    zero-initialised ones): 0.02 n for weights, 0.02 n for biases.
    """
    g = _rng(key, seed)
    n = g.standard_normal(tuple(shape), dtype=np.float32)
    leaf = key.rsplit(".", 1)[-1]
    parent = key.rsplit(".", 1)[0]
    is_norm = len(shape) == 1 and (
        parent.endswith("in_layers.0")
        or parent.endswith("out_layers.0")
        or parent.endswith(".norm")
        or parent == "out.0"
    )
    if is_norm:
        return (1.0 + 0.1 * n if leaf == "weight" else 0.1 * n).astype(np.float32)
    return (0.02 * n).astype(np.float32)


def synth_state_dict(keys_and_shapes, seed=0):
    """dict key -> float32 numpy array for an iterable of (key, shape)."""
    return {k: synth_param(k, s, seed) for k, s in keys_and_shapes}


def synth_low_res(shape, seed=1234):
    """Conditioning volume: uniform [0, 1), the reference's post-normalisation
    range (image_datasets.py:292)."""
    g = np.random.default_rng(int(seed))
    return g.random(tuple(shape), dtype=np.float32)


def synth_noise(shape, count, seed=10):
    """``count`` standard-normal tensors of ``shape`` (initial noise followed
    by one draw per sampler step, the reference's consumption order:
    scripts/test.py:62 then gaussian_diffusion.py:430 once per step)."""
    g = np.random.default_rng(int(seed))
    return [g.standard_normal(tuple(shape), dtype=np.float32) for _ in range(count)]
