"""
Pins oracle/ (the CPU restatement) against outputs of the reference itself
(tests/golden/*, produced by tests/golden/make_golden.py in the build
container).  CPU only.
"""

import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_err
from guided_diffusion import synth
from oracle import sampler_ref, schedule_ref, unet_ref

PUBLISHED = dict(large_size=96, num_channels=128, num_res_blocks=2, num_head_channels=64,
                 attention_resolutions="1000", learn_sigma=True, resblock_updown=True,
                 use_scale_shift_norm=True)
TINY = dict(PUBLISHED, num_channels=32, num_res_blocks=1)

TABLES = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "alphas_cumprod_next",
          "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod",
          "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
          "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"]


@pytest.mark.parametrize("tag,resp", [("full", ""), ("250", "250"), ("50", "50"),
                                      ("ddim50", "ddim50"), ("10", "10"), ("sect", "10,15,20")])
def test_schedule_tables_exact(golden, tag, resp):
    g = golden("schedules.npz")
    tmap, tb = schedule_ref.spaced_schedule(1000, "linear", resp)
    assert list(g[tag + "/timestep_map"]) == tmap
    for n in TABLES:
        assert np.array_equal(g[tag + "/" + n], tb[n]), n   # exact fp64


def test_cosine_betas(golden):
    # d.betas of a SpacedDiffusion is recomputed from the cumulative product
    # (respace.py:79-83), also when every step is kept.
    _, tb = schedule_ref.spaced_schedule(100, "cosine", "")
    assert np.array_equal(golden("schedules.npz")["cos100/betas"], tb["betas"])


def test_timestep_embedding(golden):
    g = golden("timestep_embedding.npz")
    t = torch.from_numpy(g["t"])
    for dim, key in [(128, "e128"), (32, "e32"), (33, "e33")]:
        assert np.array_equal(unet_ref.timestep_embedding(t, dim).numpy(), g[key])


def _sd(cfg, seed=0):
    return {k: torch.from_numpy(v)
            for k, v in synth.synth_state_dict(unet_ref.param_shapes(cfg), seed).items()}


def test_state_dict_keys_match_reference():
    with open(os.path.join(GOLDEN, "state_keys.json")) as f:
        ref = json.load(f)
    cases = {
        "tiny": unet_ref.sr_config(**TINY),
        "published": unet_ref.sr_config(**PUBLISHED),
        "tiny_attn": unet_ref.sr_config(**dict(TINY, large_size=32, attention_resolutions="8,4",
                                               num_head_channels=32)),
        "tiny_ls64": unet_ref.sr_config(**dict(TINY, large_size=64)),
    }
    for tag, cfg in cases.items():
        mine = [(k, list(s)) for k, s in unet_ref.param_shapes(cfg)]
        assert mine == [(x[0], x[1]) for x in ref[tag]], tag
    n = sum(int(np.prod(s)) for _, s in unet_ref.param_shapes(cases["published"]))
    assert n == ref["published_param_count"] == 206964610


def test_resblocks(golden):
    g = golden("resblocks.npz")
    for tag, cin, cout, ud in [("plain", 32, 32, None), ("widen", 32, 64, None),
                               ("narrow", 64, 32, None), ("down", 32, 32, "down"),
                               ("up", 32, 32, "up")]:
        shapes = [("in_layers.0.weight", (cin,)), ("in_layers.0.bias", (cin,)),
                  ("in_layers.2.weight", (cout, cin, 3, 3, 3)), ("in_layers.2.bias", (cout,)),
                  ("emb_layers.1.weight", (2 * cout, 128)), ("emb_layers.1.bias", (2 * cout,)),
                  ("out_layers.0.weight", (cout,)), ("out_layers.0.bias", (cout,)),
                  ("out_layers.3.weight", (cout, cout, 3, 3, 3)), ("out_layers.3.bias", (cout,))]
        if cin != cout:
            shapes += [("skip_connection.weight", (cout, cin, 1, 1, 1)),
                       ("skip_connection.bias", (cout,))]
        sd = {"rb." + k: torch.from_numpy(synth.synth_param("rb_%s.%s" % (tag, k), s))
              for k, s in shapes}
        rng = np.random.default_rng(7)
        x = torch.from_numpy(rng.standard_normal((2, cin, 4, 8, 8), dtype=np.float32))
        emb = torch.from_numpy(rng.standard_normal((2, 128), dtype=np.float32))
        y = unet_ref.resblock(sd, "rb", x, emb, ud, True)
        assert rel_err(y.numpy(), g[tag + "/y"]) < 1e-6, tag


def _run(cfg, shape, t_vals, seed=0):
    sd = _sd(cfg, seed)
    x = torch.from_numpy(synth.synth_noise(shape, 1, seed=3)[0])
    lr = torch.from_numpy(synth.synth_low_res(shape, seed=1234))
    t = torch.tensor(t_vals[:shape[0]], dtype=torch.long)
    with torch.no_grad():
        return unet_ref.unet_forward(sd, cfg, x, t, lr).numpy()


@pytest.mark.parametrize("tag,over,shape,t", [
    ("tiny_8x16x16", {}, (2, 1, 8, 16, 16), [37, 999]),
    ("tiny_32", {}, (1, 1, 32, 32, 32), [500]),
    ("tiny_odd", {}, (1, 1, 5, 48, 16), [3]),
    ("tiny_attn", dict(large_size=32, attention_resolutions="8,4", num_head_channels=32),
     (1, 1, 8, 32, 32), [10]),
    ("tiny_midattn", dict(large_size=32, attention_resolutions="1000", num_head_channels=32,
                          mid_attention=True), (1, 1, 4, 32, 32), [77]),
    ("tiny_convresample", dict(resblock_updown=False), (1, 1, 4, 16, 16), [5]),
    ("tiny_additive", dict(use_scale_shift_norm=False), (1, 1, 4, 16, 16), [5]),
    ("tiny_nosigma", dict(learn_sigma=False), (1, 1, 4, 16, 16), [5]),
    ("tiny_ls64", dict(large_size=64), (1, 1, 4, 16, 16), [5]),
])
def test_unet_forward_tiny(golden, tag, over, shape, t):
    cfg = unet_ref.sr_config(**dict(TINY, **over))
    y = _run(cfg, shape, t)
    ref = golden("unet_forward.npz")[tag]
    assert y.shape == ref.shape
    # same ATen ops in the same order as the reference -> agreement to rounding
    assert rel_err(y, ref) < 1e-6, tag


@pytest.mark.parametrize("tag,over,shape,t", [
    ("tiny_attn", dict(large_size=32, attention_resolutions="8,4", num_head_channels=32),
     (1, 1, 8, 32, 32), [10]),
    ("tiny_midattn", dict(large_size=32, attention_resolutions="1000", num_head_channels=32,
                          mid_attention=True), (1, 1, 4, 32, 32), [77]),
])
def test_query_blocked_attention_oracle_vs_reference(golden, monkeypatch, tag, over, shape, t):
    """The oracle walks the attention's QUERY axis in blocks (it has to at BASELINE config 5's T = 32 768, where
    the reference's materialised T x T matrix is 4.3 GB per head).  With a block smaller than T and not dividing
    it (T = 128 and 512 here) it still reproduces the reference's own outputs: the blocked form is the pinned
    one, and is what the full-size GPU tests of the attention kernel are compared with."""
    monkeypatch.setattr(unet_ref, "ATTN_QUERY_BLOCK", 48)
    cfg = unet_ref.sr_config(**dict(TINY, **over))
    y = _run(cfg, shape, t)
    assert rel_err(y, golden("unet_forward.npz")[tag]) < 1e-6, tag
    # and the fp64 evaluation used as the bar of those tests agrees with the fp32 one to fp32 rounding
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(2, 32, 200, generator=g) for _ in range(3))
    a32 = unet_ref.qkv_attention(q, k, v, block=64)
    a64 = unet_ref.qkv_attention(q, k, v, block=200, dtype=torch.float64)
    assert rel_err(a32.numpy(), a64.numpy()) < 2e-6


def test_unet_forward_published_arch(golden):
    cfg = unet_ref.sr_config(**PUBLISHED)
    y = _run(cfg, (1, 1, 8, 32, 32), [251])
    assert rel_err(y, golden("unet_forward.npz")["published_8x32x32"]) < 1e-6


@pytest.mark.parametrize("tag,over,shape,resp,kind,eta,kw", [
    ("ddpm10_32", {}, (1, 1, 32, 32, 32), "10", "ddpm", 0.0, {}),
    ("ddim10_8x16x16", {}, (2, 1, 8, 16, 16), "ddim10", "ddim", 0.0, {}),
    ("ddim10_eta_8x16x16", {}, (1, 1, 8, 16, 16), "ddim10", "ddim", 0.5, {}),
    ("ddpm10_nosigma", dict(learn_sigma=False), (1, 1, 4, 16, 16), "10", "ddpm", 0.0, {}),
    ("ddpm10_noclip", {}, (1, 1, 4, 16, 16), "10", "ddpm", 0.0, dict(clip_denoised=False)),
    ("ddpm10_xstart", {}, (1, 1, 4, 16, 16), "10", "ddpm", 0.0, dict(predict_xstart=True)),
])
def test_sampler_loops(golden, tag, over, shape, resp, kind, eta, kw):
    cfg = unet_ref.sr_config(**dict(TINY, **over))
    sd = _sd(cfg)
    learn_sigma = dict(TINY, **over)["learn_sigma"]
    tmap, tb = schedule_ref.spaced_schedule(1000, "linear", resp)
    draws = [torch.from_numpy(a) for a in synth.synth_noise(shape, len(tmap) + 1, seed=10)]
    lr = torch.from_numpy(synth.synth_low_res(shape, seed=1234))

    def model_fn(x, t, low_res):
        return unet_ref.unet_forward(sd, cfg, x, t, low_res)

    trace = []
    with torch.no_grad():
        if kind == "ddpm":
            out = sampler_ref.p_sample_loop(model_fn, tmap, tb, draws[0], draws[1:], lr,
                                            learn_sigma=learn_sigma, trace=trace, **kw)
        else:
            out = sampler_ref.ddim_sample_loop(model_fn, tmap, tb, draws[0], draws[1:], lr,
                                               eta=eta, learn_sigma=learn_sigma, trace=trace, **kw)
    g = golden("sampler.npz")
    assert rel_err(out.numpy(), g[tag + "/sample"]) < 1e-5, tag
    assert np.allclose(np.array(trace), g[tag + "/trace"], rtol=1e-4, atol=1e-5)


def test_published_250_step_chain_head(golden):
    """The long-horizon golden (published architecture, timestep_respacing "250", 1x1x8x32x32,
    tests/golden/sampler_published250.npz) is what the GPU path is held to at 250 steps; here the
    oracle walks the first steps of the same chain (the whole chain is ~2.5 CPU-minutes) and must
    reproduce the reference's per-step sample mean / pred_xstart mean / sample std."""
    g = golden("sampler_published250.npz")
    cfg = unet_ref.sr_config(**PUBLISHED)
    sd = _sd(cfg)
    shape = (1, 1, 8, 32, 32)
    tmap, tb = schedule_ref.spaced_schedule(1000, "linear", "250")
    steps = 5
    draws = [torch.from_numpy(a) for a in synth.synth_noise(shape, 251, seed=10)]
    lr = torch.from_numpy(synth.synth_low_res(shape, seed=1234))
    img, rows = draws[0], []
    with torch.no_grad():
        for k in range(steps):
            i = 249 - k
            out = unet_ref.unet_forward(sd, cfg, img, torch.full((1,), tmap[i], dtype=torch.long), lr)
            mean, log_var, x0 = sampler_ref.mean_variance(tb, out, img, i)
            img = mean + torch.exp(0.5 * log_var) * draws[k + 1]
            rows.append((float(img.mean()), float(x0.mean()), float(img.std())))
    assert np.allclose(np.array(rows), g["trace"][:steps], rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------- the 2-D network (create_model)
MODEL2D = dict(image_size=64, num_channels=32, num_res_blocks=1, channel_mult="1,2,2", num_head_channels=32,
               attention_resolutions="16", learn_sigma=True, use_scale_shift_norm=True)
MODEL2D_VARIANTS = {
    "film": {},
    "updown_additive": dict(resblock_updown=True, use_scale_shift_norm=False, learn_sigma=False),
}


@pytest.mark.parametrize("tag", sorted(MODEL2D_VARIANTS))
def test_model2d_oracle_vs_reference(golden, tag):
    """create_model_and_diffusion's 2-D RGB network (script_util.py:74-184) through the oracle:
    state_dict layout, one forward, 6-step DDPM and DDIM loops on (2, 3, 32, 48) images -- against
    the reference's outputs (tests/golden/model2d.npz, make_golden.py gen_model2d)."""
    fl = dict(MODEL2D, **MODEL2D_VARIANTS[tag])
    cfg = unet_ref.model2d_config(**fl)
    with open(os.path.join(GOLDEN, "model2d_keys.json")) as f:
        ref_keys = json.load(f)[tag]
    assert [[k, list(s)] for k, s in unet_ref.param_shapes(cfg)] == ref_keys
    sd = _sd(cfg, seed=2)
    g = golden("model2d.npz")
    shape = (2, 3, 32, 48)
    x = torch.from_numpy(synth.synth_noise(shape, 1, seed=3)[0])
    with torch.no_grad():
        y = unet_ref.unet_forward(sd, cfg, x, torch.tensor([617, 3])).numpy()
    assert rel_err(y, g[tag + "/forward"]) < 1e-6
    tmap, tb = schedule_ref.spaced_schedule(1000, "linear", "6")
    draws = [torch.from_numpy(a) for a in synth.synth_noise(shape, 7, seed=10)]

    def model_fn(xx, t, _):
        return unet_ref.unet_forward(sd, cfg, xx, t)

    with torch.no_grad():
        a = sampler_ref.p_sample_loop(model_fn, tmap, tb, draws[0], draws[1:], None, learn_sigma=fl["learn_sigma"])
        b = sampler_ref.ddim_sample_loop(model_fn, tmap, tb, draws[0], draws[1:], None,
                                         learn_sigma=fl["learn_sigma"])
    assert rel_err(a.numpy(), g[tag + "/ddpm"]) < 1e-5
    assert rel_err(b.numpy(), g[tag + "/ddim"]) < 1e-5


# ------------------------------------------- class conditioning and the other attention order
API_EXTRAS_2D = {
    "new_order": dict(use_new_attention_order=True),
    "new_order_class_cond": dict(use_new_attention_order=True, class_cond=True, num_head_channels=-1, num_heads=2),
}


def test_class_conditional_sr_oracle_vs_reference(golden):
    """label_emb (unet.py:476-478, :703-705) on the 3-D SR network: layout, forward with y, and a 3-step
    p_sample_loop with y in model_kwargs -- against the reference (tests/golden/api_extras.npz)."""
    tiny = dict(large_size=96, num_channels=32, num_res_blocks=1, num_head_channels=64, attention_resolutions="1000",
                learn_sigma=True, resblock_updown=True, use_scale_shift_norm=True, class_cond=True)
    cfg = unet_ref.sr_config(**tiny)
    with open(os.path.join(GOLDEN, "api_extras_keys.json")) as f:
        ref_keys = json.load(f)["sr_class_cond"]
    assert [[k, list(s)] for k, s in unet_ref.param_shapes(cfg)] == ref_keys
    sd = _sd(cfg, seed=4)
    g = golden("api_extras.npz")
    shape = (2, 1, 4, 16, 16)
    x = torch.from_numpy(synth.synth_noise(shape, 1, seed=3)[0])
    lr = torch.from_numpy(synth.synth_low_res(shape, seed=1234))
    y = torch.from_numpy(g["sr_class_cond/y"])
    with torch.no_grad():
        out = unet_ref.unet_forward(sd, cfg, x, torch.tensor([37, 999]), lr, y=y).numpy()
    assert rel_err(out, g["sr_class_cond/forward"]) < 1e-6
    tmap, tb = schedule_ref.spaced_schedule(1000, "linear", "3")
    draws = [torch.from_numpy(a) for a in synth.synth_noise(shape, 4, seed=10)]
    with torch.no_grad():
        a = sampler_ref.p_sample_loop(lambda xx, t, c: unet_ref.unet_forward(sd, cfg, xx, t, c, y=y), tmap, tb,
                                      draws[0], draws[1:], lr)
    assert rel_err(a.numpy(), g["sr_class_cond/ddpm3"]) < 1e-5


@pytest.mark.parametrize("tag", sorted(API_EXTRAS_2D))
def test_new_attention_order_oracle_vs_reference(golden, tag):
    """QKVAttention (unet.py:361-389) in the 2-D network, alone and with class conditioning."""
    fl = dict(MODEL2D, **API_EXTRAS_2D[tag])
    cfg = unet_ref.model2d_config(**fl)
    assert cfg["new_attention_order"]
    with open(os.path.join(GOLDEN, "api_extras_keys.json")) as f:
        ref_keys = json.load(f)[tag]
    assert [[k, list(s)] for k, s in unet_ref.param_shapes(cfg)] == ref_keys
    sd = _sd(cfg, seed=2)
    g = golden("api_extras.npz")
    shape = (2, 3, 32, 48)
    x = torch.from_numpy(synth.synth_noise(shape, 1, seed=3)[0])
    y = torch.tensor([5, 0]) if fl.get("class_cond") else None
    with torch.no_grad():
        out = unet_ref.unet_forward(sd, cfg, x, torch.tensor([617, 3]), y=y).numpy()
    assert rel_err(out, g[tag + "/forward"]) < 1e-6
    tmap, tb = schedule_ref.spaced_schedule(1000, "linear", "6")
    draws = [torch.from_numpy(a) for a in synth.synth_noise(shape, 7, seed=10)]
    with torch.no_grad():
        a = sampler_ref.p_sample_loop(lambda xx, t, _: unet_ref.unet_forward(sd, cfg, xx, t, y=y), tmap, tb,
                                      draws[0], draws[1:], None)
    assert rel_err(a.numpy(), g[tag + "/ddpm"]) < 1e-5


def test_oracle_first_step_of_config2_vs_reference(golden):
    """tests/golden/sampler250_64.npz is the reference's own run of BASELINE config 2 (published architecture,
    1x64^3, 250 steps).  The oracle reproduces its first reverse step (one full-size forward + p_sample update on
    the injected noise): the fixture, the oracle and the GPU tests that use either all speak of the same run."""
    g = golden("sampler250_64.npz")
    cfg = unet_ref.sr_config(**PUBLISHED)
    sd = _sd(cfg)
    tmap, tb = schedule_ref.spaced_schedule(1000, "linear", "250")
    shape = (1, 1, 64, 64, 64)
    draws = [torch.from_numpy(a) for a in synth.synth_noise(shape, 2, seed=10)]
    lr = torch.from_numpy(synth.synth_low_res(shape, seed=1234))
    i = len(tmap) - 1
    with torch.no_grad():
        out = unet_ref.unet_forward(sd, cfg, draws[0], torch.full((1,), tmap[i], dtype=torch.long), lr)
        mean, log_var, x0 = sampler_ref.mean_variance(tb, out, draws[0], i, True, False, True)
        img = mean + torch.exp(0.5 * log_var) * draws[1]
    assert rel_err(img.numpy(), g["after1"]) < 1e-5
    assert abs(float(img.mean()) - g["trace"][0][0]) < 1e-6 and abs(float(x0.mean()) - g["trace"][0][1]) < 1e-6
