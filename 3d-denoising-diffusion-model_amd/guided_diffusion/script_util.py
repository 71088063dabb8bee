"""
Factory / flag surface of the reference (script_util.py:11-65, 269-331,
334-450, 578-644) so that scripts written against it keep working: the same
function names, keyword sets, defaults, flag parsing rules and
(model, diffusion) return convention -- with the model and diffusion objects
being the HIP-backed ones of this package.
"""

import argparse
import inspect

from . import gaussian_diffusion as gd
from .respace import SpacedDiffusion, space_timesteps
from .unet import SuperResModel_noatt, UNetModel

NUM_CLASSES = 1000

_DIFFUSION_DEFAULTS = (
    ("learn_sigma", False), ("diffusion_steps", 1000), ("noise_schedule", "linear"),
    ("timestep_respacing", ""), ("use_kl", False), ("predict_xstart", False),
    ("rescale_timesteps", False), ("rescale_learned_sigmas", False),
)
_MODEL_DEFAULTS = (
    ("image_size", 64), ("num_channels", 128), ("num_res_blocks", 2), ("num_heads", 4),
    ("num_heads_upsample", -1), ("num_head_channels", -1), ("attention_resolutions", "16,8"),
    ("channel_mult", ""), ("dropout", 0.0), ("class_cond", False), ("use_checkpoint", False),
    ("use_scale_shift_norm", True), ("resblock_updown", False), ("use_fp16", False),
    ("use_new_attention_order", False),
)


def diffusion_defaults():
    return dict(_DIFFUSION_DEFAULTS)


def model_and_diffusion_defaults():
    res = dict(_MODEL_DEFAULTS)
    res.update(diffusion_defaults())
    return res


def sr_model_and_diffusion_defaults():
    """script_util.py:269-277: image defaults + large/small size, restricted to
    the keyword set of sr_create_model_and_diffusion."""
    res = model_and_diffusion_defaults()
    res["large_size"] = 256
    res["small_size"] = 64
    accepted = inspect.getfullargspec(sr_create_model_and_diffusion)[0]
    return {k: v for k, v in res.items() if k in accepted}


def _diffusion_from(kw):
    return create_gaussian_diffusion(
        steps=kw["diffusion_steps"], learn_sigma=kw["learn_sigma"], noise_schedule=kw["noise_schedule"],
        use_kl=kw["use_kl"], predict_xstart=kw["predict_xstart"], rescale_timesteps=kw["rescale_timesteps"],
        rescale_learned_sigmas=kw["rescale_learned_sigmas"], timestep_respacing=kw["timestep_respacing"])


def sr_create_model_and_diffusion(large_size, small_size, class_cond, learn_sigma, num_channels,
                                  num_res_blocks, num_heads, num_head_channels, num_heads_upsample,
                                  attention_resolutions, dropout, diffusion_steps, noise_schedule,
                                  timestep_respacing, use_kl, predict_xstart, rescale_timesteps,
                                  rescale_learned_sigmas, use_checkpoint, use_scale_shift_norm,
                                  resblock_updown, use_fp16):
    kw = dict(locals())
    model = sr_create_model(
        large_size, small_size, num_channels, num_res_blocks, learn_sigma=learn_sigma, class_cond=class_cond,
        use_checkpoint=use_checkpoint, attention_resolutions=attention_resolutions, num_heads=num_heads,
        num_head_channels=num_head_channels, num_heads_upsample=num_heads_upsample,
        use_scale_shift_norm=use_scale_shift_norm, dropout=dropout, resblock_updown=resblock_updown,
        use_fp16=use_fp16)
    return model, _diffusion_from(kw)


def _sr_channel_mult(large_size):
    # script_util.py:353-361: anything that is not 512 / 256 / 64 (96, 128, 32 ...) takes
    # the five-level multiplier.
    if large_size in (512, 256):
        return (1, 1, 2, 2, 4, 4)
    if large_size == 64:
        return (1, 2, 3, 4)
    return (1, 1, 2, 3, 4)


def sr_create_model(large_size, small_size, num_channels, num_res_blocks, learn_sigma, class_cond,
                    use_checkpoint, attention_resolutions, num_heads, num_head_channels, num_heads_upsample,
                    use_scale_shift_norm, dropout, resblock_updown, use_fp16):
    """script_util.py:334-450; the live return (:432-450) builds SuperResModel_noatt
    with one input channel (doubled by the low_res concat) and dims=3."""
    del small_size
    attention_ds = tuple(large_size // int(res) for res in attention_resolutions.split(","))
    return SuperResModel_noatt(
        image_size=large_size, in_channels=1, model_channels=num_channels,
        out_channels=(2 if learn_sigma else 1), num_res_blocks=num_res_blocks,
        attention_resolutions=attention_ds, dropout=dropout, channel_mult=_sr_channel_mult(large_size),
        dims=3, num_classes=(NUM_CLASSES if class_cond else None), use_checkpoint=use_checkpoint,
        num_heads=num_heads, num_head_channels=num_head_channels, num_heads_upsample=num_heads_upsample,
        use_scale_shift_norm=use_scale_shift_norm, resblock_updown=resblock_updown, use_fp16=use_fp16)


def create_model_and_diffusion(image_size, class_cond, learn_sigma, num_channels, num_res_blocks,
                               channel_mult, num_heads, num_head_channels, num_heads_upsample,
                               attention_resolutions, dropout, diffusion_steps, noise_schedule,
                               timestep_respacing, use_kl, predict_xstart, rescale_timesteps,
                               rescale_learned_sigmas, use_checkpoint, use_scale_shift_norm, resblock_updown,
                               use_fp16, use_new_attention_order):
    """script_util.py:74-127: the 2-D RGB UNetModel (dims=2) + its diffusion.  The 2-D convs run as
    depth-1 3-D convs on the same engine (engine.py)."""
    kw = dict(locals())
    model = create_model(image_size, num_channels, num_res_blocks, channel_mult=channel_mult,
                         learn_sigma=learn_sigma, class_cond=class_cond, use_checkpoint=use_checkpoint,
                         attention_resolutions=attention_resolutions, num_heads=num_heads,
                         num_head_channels=num_head_channels, num_heads_upsample=num_heads_upsample,
                         use_scale_shift_norm=use_scale_shift_norm, dropout=dropout,
                         resblock_updown=resblock_updown, use_fp16=use_fp16,
                         use_new_attention_order=use_new_attention_order)
    return model, _diffusion_from(kw)


def create_model(image_size, num_channels, num_res_blocks, channel_mult="", learn_sigma=False,
                 class_cond=False, use_checkpoint=False, attention_resolutions="16", num_heads=1,
                 num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False, dropout=0,
                 resblock_updown=False, use_fp16=False, use_new_attention_order=False):
    """script_util.py:130-184."""
    if channel_mult == "":
        table = {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4)}
        if image_size not in table:
            raise ValueError(f"unsupported image size: {image_size}")
        channel_mult = table[image_size]
    else:
        channel_mult = tuple(int(m) for m in channel_mult.split(","))
    attention_ds = tuple(image_size // int(res) for res in attention_resolutions.split(","))
    return UNetModel(
        image_size=image_size, in_channels=3, model_channels=num_channels,
        out_channels=(3 if not learn_sigma else 6), num_res_blocks=num_res_blocks,
        attention_resolutions=attention_ds, dropout=dropout, channel_mult=channel_mult,
        num_classes=(NUM_CLASSES if class_cond else None), use_checkpoint=use_checkpoint, use_fp16=use_fp16,
        num_heads=num_heads, num_head_channels=num_head_channels, num_heads_upsample=num_heads_upsample,
        use_scale_shift_norm=use_scale_shift_norm, resblock_updown=resblock_updown,
        use_new_attention_order=use_new_attention_order)


def create_gaussian_diffusion(*, steps=1000, learn_sigma=False, sigma_small=False, noise_schedule="linear",
                              use_kl=False, predict_xstart=False, rescale_timesteps=False,
                              rescale_learned_sigmas=False, timestep_respacing=""):
    """script_util.py:578-616."""
    betas = gd.get_named_beta_schedule(noise_schedule, steps)
    if use_kl:
        loss_type = gd.LossType.RESCALED_KL
    elif rescale_learned_sigmas:
        loss_type = gd.LossType.RESCALED_MSE
    else:
        loss_type = gd.LossType.MSE
    if learn_sigma:
        var_type = gd.ModelVarType.LEARNED_RANGE
    else:
        var_type = gd.ModelVarType.FIXED_SMALL if sigma_small else gd.ModelVarType.FIXED_LARGE
    return SpacedDiffusion(
        use_timesteps=space_timesteps(steps, timestep_respacing if timestep_respacing else [steps]),
        betas=betas,
        model_mean_type=gd.ModelMeanType.START_X if predict_xstart else gd.ModelMeanType.EPSILON,
        model_var_type=var_type, loss_type=loss_type, rescale_timesteps=rescale_timesteps)


def add_dict_to_argparser(parser, default_dict):
    """script_util.py:619-626: one --flag per key, type inferred from the default."""
    for k, v in default_dict.items():
        if v is None:
            v_type = str
        elif isinstance(v, bool):
            v_type = str2bool
        else:
            v_type = type(v)
        parser.add_argument(f"--{k}", default=v, type=v_type)


def args_to_dict(args, keys):
    return {k: getattr(args, k) for k in keys}


def str2bool(v):
    """script_util.py:633-644."""
    if isinstance(v, bool):
        return v
    s = v.lower()
    if s in ("yes", "true", "t", "y", "1"):
        return True
    if s in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("boolean value expected")
