"""
Timestep respacing (reference respace.py:7-128): which steps of the base
process are kept, the betas of the shortened process, and the step-index ->
original-timestep mapping the network is conditioned on.
"""

import numpy as np
import torch as th

from .gaussian_diffusion import GaussianDiffusion


def space_timesteps(num_timesteps, section_counts):
    """respace.py:7-60.  "ddimN" = fixed integer stride giving exactly N steps;
    otherwise per-section counts with fractional strides rounded by Python's
    round() (half to even)."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[len("ddim"):])
            for stride in range(1, num_timesteps):
                steps = range(0, num_timesteps, stride)
                if len(steps) == want:
                    return set(steps)
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    per, extra = divmod(num_timesteps, len(section_counts))
    kept, start = [], 0
    for i, count in enumerate(section_counts):
        size = per + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        pos = 0.0
        for _ in range(count):
            kept.append(start + round(pos))
            pos += stride
        start += size
    return set(kept)


class SpacedDiffusion(GaussianDiffusion):
    """respace.py:63-113: a diffusion over a subset of the base timesteps."""

    def __init__(self, use_timesteps, **kwargs):
        self.use_timesteps = set(use_timesteps)
        self.original_num_steps = len(kwargs["betas"])
        base = GaussianDiffusion(**kwargs)
        self.timestep_map = [i for i in range(self.original_num_steps) if i in self.use_timesteps]
        # beta_k = 1 - acp[i_k] / acp[i_{k-1}]   (respace.py:79-83)
        prev, new_betas = 1.0, []
        for i in self.timestep_map:
            a = base.alphas_cumprod[i]
            new_betas.append(1 - a / prev)
            prev = a
        kwargs["betas"] = np.array(new_betas)
        super().__init__(**kwargs)

    def _scale_timesteps(self, t):
        return t  # scaling is applied together with the index mapping below

    def _model_timesteps(self, t):
        # respace.py:123-128
        table = th.tensor(self.timestep_map, device=t.device, dtype=t.dtype)
        mapped = table[t]
        if self.rescale_timesteps:
            mapped = mapped.float() * (1000.0 / self.original_num_steps)
        return mapped

    def _wrap_model(self, model):
        if isinstance(model, _WrappedModel):
            return model
        return _WrappedModel(model, self.timestep_map, self.rescale_timesteps, self.original_num_steps)


class _WrappedModel:
    """respace.py:116-128 (kept for callers that wrap models themselves)."""

    def __init__(self, model, timestep_map, rescale_timesteps, original_num_steps):
        self.model = model
        self.timestep_map = timestep_map
        self.rescale_timesteps = rescale_timesteps
        self.original_num_steps = original_num_steps

    def __call__(self, x, ts, **kwargs):
        table = th.tensor(self.timestep_map, device=ts.device, dtype=ts.dtype)
        new_ts = table[ts]
        if self.rescale_timesteps:
            new_ts = new_ts.float() * (1000.0 / self.original_num_steps)
        return self.model(x, new_ts, **kwargs)
