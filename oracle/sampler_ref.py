"""
ORACLE (test infrastructure): DDPM ancestral and DDIM sampling loops, torch
CPU fp32 with fp64-computed / fp32-applied coefficients.

Follows gaussian_diffusion.py:232-333 (p_mean_variance for the
EPSILON / START_X mean types and LEARNED_RANGE / FIXED_LARGE / FIXED_SMALL
variance types reachable from script_util.py:602-613), :395-439 (p_sample),
:487-535 (loop), :537-585 (ddim_sample), :659-707 (ddim loop), :897-910
(table lookup: fp64 table -> .float() scalar) and respace.py:123-128
(step index -> original timestep before the network call).

``model_fn(x, t_original, low_res) -> (N, C_out, D, H, W)`` is any callable;
noise is INJECTED (one tensor per step, drawn by the caller in the reference's
order) so the CPU oracle and the GPU path consume identical randomness.
"""

import numpy as np
import torch


def _coef(table, i):
    # _extract_into_tensor: fp64 table entry cast to fp32 (gaussian_diffusion.py:907)
    return float(np.float32(table[i]))


def mean_variance(tb, model_out, x, i, learn_sigma=True, predict_xstart=False,
                  clip_denoised=True, sigma_small=False):
    """gaussian_diffusion.py:262-326 for one (batch-uniform) step index i."""
    C = x.shape[1]
    if learn_sigma:
        eps, v = torch.split(model_out, C, dim=1)
        min_log = _coef(tb["posterior_log_variance_clipped"], i)
        max_log = _coef(np.log(tb["betas"]), i)
        frac = (v + 1) / 2
        log_var = frac * max_log + (1 - frac) * min_log
    else:
        eps = model_out
        if sigma_small:
            lv = tb["posterior_log_variance_clipped"]
        else:
            lv = np.log(np.append(tb["posterior_variance"][1], tb["betas"][1:]))
        log_var = torch.full_like(x, _coef(lv, i))
    if predict_xstart:
        x0 = eps
    else:
        x0 = _coef(tb["sqrt_recip_alphas_cumprod"], i) * x \
            - _coef(tb["sqrt_recipm1_alphas_cumprod"], i) * eps
    if clip_denoised:
        x0 = x0.clamp(-1, 1)
    mean = _coef(tb["posterior_mean_coef1"], i) * x0 + _coef(tb["posterior_mean_coef2"], i) * x
    return mean, log_var, x0


def p_sample_loop(model_fn, tmap, tb, noise0, step_noise, low_res, learn_sigma=True,
                  predict_xstart=False, clip_denoised=True, trace=None):
    """Ancestral sampling; step_noise[k] is the k-th randn_like draw (the
    reference draws one every step, including the masked last one)."""
    img = noise0
    T = len(tmap)
    N = img.shape[0]
    for k, i in enumerate(range(T - 1, -1, -1)):
        t_orig = torch.full((N,), tmap[i], dtype=torch.long)
        out = model_fn(img, t_orig, low_res)
        mean, log_var, x0 = mean_variance(tb, out, img, i, learn_sigma, predict_xstart,
                                          clip_denoised)
        mask = 0.0 if i == 0 else 1.0
        img = mean + mask * torch.exp(0.5 * log_var) * step_noise[k]
        if trace is not None:
            trace.append((float(img.mean()), float(x0.mean())))
    return img


def ddim_sample_loop(model_fn, tmap, tb, noise0, step_noise, low_res, eta=0.0,
                     learn_sigma=True, predict_xstart=False, clip_denoised=True, trace=None):
    """gaussian_diffusion.py:537-585."""
    img = noise0
    T = len(tmap)
    N = img.shape[0]
    for k, i in enumerate(range(T - 1, -1, -1)):
        t_orig = torch.full((N,), tmap[i], dtype=torch.long)
        out = model_fn(img, t_orig, low_res)
        _, _, x0 = mean_variance(tb, out, img, i, learn_sigma, predict_xstart, clip_denoised)
        # eps re-derived from the (clipped) x0 (:566, :345-349)
        eps = (_coef(tb["sqrt_recip_alphas_cumprod"], i) * img - x0) \
            / _coef(tb["sqrt_recipm1_alphas_cumprod"], i)
        ab = torch.tensor(_coef(tb["alphas_cumprod"], i))
        ab_prev = torch.tensor(_coef(tb["alphas_cumprod_prev"], i))
        sigma = eta * torch.sqrt((1 - ab_prev) / (1 - ab)) * torch.sqrt(1 - ab / ab_prev)
        mean_pred = x0 * torch.sqrt(ab_prev) + torch.sqrt(1 - ab_prev - sigma ** 2) * eps
        mask = 0.0 if i == 0 else 1.0
        img = mean_pred + mask * sigma * step_noise[k]
        if trace is not None:
            trace.append((float(img.mean()), float(x0.mean())))
    return img
