"""
CPU tier: host-side logic of the product package (flag system, schedules,
state_dict key scheme, topology) against the reference's golden outputs, and
the C-ABI library's exports.  No compute call is made here (no GPU).
"""

import argparse
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT
from guided_diffusion import _hip
from guided_diffusion import gaussian_diffusion as gd
from guided_diffusion import script_util as su
from guided_diffusion.respace import SpacedDiffusion, _WrappedModel, space_timesteps

PUBLISHED = dict(large_size=96, small_size=96, num_channels=128, num_res_blocks=2, num_head_channels=64,
                 attention_resolutions="1000", learn_sigma=True, resblock_updown=True,
                 use_scale_shift_norm=True)
TINY = dict(PUBLISHED, num_channels=32, num_res_blocks=1)
TABLES = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "alphas_cumprod_next", "sqrt_alphas_cumprod",
          "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
          "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
          "posterior_mean_coef1", "posterior_mean_coef2"]


def test_library_exports_every_declared_symbol():
    """include/ddpm3d.h <-> libddpm3d.so <-> the ctypes table agree."""
    hdr = open(os.path.join(ROOT, "include", "ddpm3d.h")).read()
    declared = set(re.findall(r"\b(ddpm3d_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_hip.EXPORTS), declared ^ set(_hip.EXPORTS)
    assert os.path.exists(_hip.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.ddpm3d_abi_version.restype = ctypes.c_int
    assert lib.ddpm3d_abi_version() == _hip.ABI_VERSION
    # pure host-side helpers are callable without a GPU
    lib.ddpm3d_packed_weight_bytes.restype = ctypes.c_size_t
    assert lib.ddpm3d_packed_weight_bytes(128, 128, 3, 0) == 27 * 128 * 128 * 4
    assert lib.ddpm3d_packed_weight_bytes(2, 128, 3, 0) == 27 * 128 * 32 * 4      # Cout padded to 32
    assert lib.ddpm3d_packed_weight_bytes(128, 2, 3, 0) == 27 * 16 * 128 * 4      # Cin padded to 16
    # split-f16 image: hi+lo f16 = the same bytes, plus one fp32 output scale per padded cout
    assert lib.ddpm3d_packed_weight_bytes(128, 128, 3, 1) == 27 * 128 * 128 * 4 + 128 * 4
    assert lib.ddpm3d_packed_weight_bytes(8, 8, 2, 0) == 0 and lib.ddpm3d_packed_weight_bytes(8, 8, 3, 7) == 0
    lib.ddpm3d_conv_workspace_bytes.restype = ctypes.c_size_t
    # 64^3 level: 2048 tiles of 128 voxels (2x8x8) fill the chip, no split, one row per tile
    assert lib.ddpm3d_conv_stats_rows(1, 64, 64, 64, 128, 128, 3, 3) == 32 * 8 * 8
    # 64x32x32 level: 128-voxel tiles (2x8x8)
    assert lib.ddpm3d_conv_stats_rows(1, 64, 32, 32, 128, 128, 3, 3) == 32 * 4 * 4
    assert lib.ddpm3d_conv_workspace_bytes(1, 64, 64, 64, 128, 128, 3, 3) == 0
    # 64x4x4 level: 8 voxel tiles -> split over Cin; the reduce kernel's rows shrink from 16 to 4
    # voxels on this level so that it still launches >= 1024 workgroups (256 rows x 2 quad blocks)
    assert lib.ddpm3d_conv_stats_rows(1, 64, 4, 4, 512, 512, 3, 3) == 256
    # eight samples of the same level: 8 x 64 rows x 2 quad blocks of 16 voxels already fill it
    assert lib.ddpm3d_conv_stats_rows(8, 64, 4, 4, 512, 512, 3, 3) == 64
    ws = lib.ddpm3d_conv_workspace_bytes(1, 64, 4, 4, 512, 512, 3, 3)
    assert ws > 0 and ws % (1024 * 512 * 4) == 0


def test_split_rule_and_prepass_argument_checks_without_a_gpu():
    """Host-side rules a caller sizes buffers by (no kernel is launched): the split over Cin stays within 16
    ways for 3x3x3 convs (r03: the reduce kernel's S slab reads are not in the cost model) and within the
    chunk count for 1x1 convs, the workspace is S whole output tensors; ddpm3d_pool_act refuses channel counts
    that are not a multiple of 4 and an activation without its affine before it touches the device."""
    lib = ctypes.CDLL(_hip.LIB_PATH)
    lib.ddpm3d_conv_workspace_bytes.restype = ctypes.c_size_t
    for (D, H, W, ci, co, k) in [(64, 4, 4, 1024, 384, 3), (64, 4, 4, 512, 512, 3), (64, 8, 8, 768, 384, 3),
                                 (64, 4, 4, 1024, 512, 1), (64, 8, 8, 768, 384, 1), (64, 16, 16, 512, 256, 1)]:
        out_bytes = D * H * W * co * 4
        for prec in ((3, 6, 0) if k == 3 else (1, 5, 0)):      # the rule is per arithmetic mode since ABI 12
            ws = lib.ddpm3d_conv_workspace_bytes(1, D, H, W, ci, co, k, prec)
            assert ws % out_bytes == 0
            S = ws // out_bytes
            assert S <= (16 if k == 3 else ci // 16), (D, H, W, ci, co, k, prec, S)
    assert lib.ddpm3d_conv_workspace_bytes(1, 64, 4, 4, 1024, 384, 3, 3) == 16 * 64 * 4 * 4 * 384 * 4
    assert lib.ddpm3d_conv_workspace_bytes(1, 64, 4, 4, 1024, 384, 3, 7) == 0      # unknown precision
    lib.ddpm3d_pool_act.restype = ctypes.c_int
    vp = ctypes.c_void_p
    lib.ddpm3d_pool_act.argtypes = [vp, vp, vp] + [ctypes.c_int] * 7 + [vp, ctypes.c_int, vp]
    buf = (ctypes.c_float * 64)()
    a = ctypes.addressof(buf)
    a16 = (a + 15) & ~15
    assert lib.ddpm3d_pool_act(a16, None, None, 0, 0, 1, 1, 1, 1, 6, a16, 0, None) == _hip.E_INVAL     # C % 4
    assert lib.ddpm3d_pool_act(a16, None, None, 1, 0, 1, 1, 1, 1, 4, a16, 0, None) == _hip.E_INVAL     # act without affine
    assert lib.ddpm3d_pool_act(a16, a16, None, 0, 0, 1, 1, 1, 1, 4, a16, 0, None) == _hip.E_INVAL      # A without B
    assert lib.ddpm3d_pool_act(a16, None, None, 0, 0, 1, 1, 1, 1, 4, a16, 0x40, None) == _hip.E_INVAL  # unknown io bits
    assert lib.ddpm3d_pool_act(None, None, None, 0, 0, 1, 1, 1, 1, 4, a16, 0, None) == _hip.E_INVAL    # null source


def test_conv_desc_struct_layout_matches_header():
    """Field order/offsets of the ctypes mirror: 8 ints, 2 ptrs, 2 ints, 2 ptrs, 2 ints, ..."""
    f = {n: getattr(_hip.ConvDesc, n).offset for n, _ in _hip.ConvDesc._fields_}
    assert f["src0"] == 32 and f["C0"] == 48 and f["aff_a"] == 56 and f["act"] == 72
    assert f["w_packed"] == 80 and f["bias_stride_n"] == 96 and f["res"] == 104 and f["out"] == 112
    assert f["out_layout"] == 120 and f["stats"] == 128 and f["workspace"] == 136
    assert f["workspace_bytes"] == 144 and f["kernel_hint"] == 152 and f["in_bound_count"] == 156
    assert f["in_bound"] == 160 and f["in_bound_stride"] == 168 and f["io_dtype"] == 172
    assert ctypes.sizeof(_hip.ConvDesc) == 176


def test_sr_defaults_and_flag_parsing():
    d = su.sr_model_and_diffusion_defaults()
    assert list(d) == ["num_channels", "num_res_blocks", "num_heads", "num_heads_upsample",
                       "num_head_channels", "attention_resolutions", "dropout", "class_cond",
                       "use_checkpoint", "use_scale_shift_norm", "resblock_updown", "use_fp16",
                       "learn_sigma", "diffusion_steps", "noise_schedule", "timestep_respacing", "use_kl",
                       "predict_xstart", "rescale_timesteps", "rescale_learned_sigmas", "large_size",
                       "small_size"]
    assert d["large_size"] == 256 and d["small_size"] == 64 and d["attention_resolutions"] == "16,8"
    assert "channel_mult" not in d and "image_size" not in d and "use_new_attention_order" not in d
    p = argparse.ArgumentParser()
    su.add_dict_to_argparser(p, dict(d, save_dir="", clip_denoised=True, none_flag=None))
    a = p.parse_args("--use_fp16 True --large_size 96 --learn_sigma yes --dropout 0.1 --none_flag x".split())
    assert a.use_fp16 is True and a.large_size == 96 and a.learn_sigma is True and a.dropout == 0.1
    assert a.none_flag == "x" and a.clip_denoised is True
    assert su.args_to_dict(a, ["large_size", "use_fp16"]) == {"large_size": 96, "use_fp16": True}
    with pytest.raises(argparse.ArgumentTypeError):
        su.str2bool("maybe")


def test_state_dict_keys_and_init_semantics():
    with open(os.path.join(GOLDEN, "state_keys.json")) as f:
        ref = json.load(f)
    cases = {"tiny": TINY, "published": PUBLISHED,
             "tiny_attn": dict(TINY, large_size=32, attention_resolutions="8,4", num_head_channels=32),
             "tiny_ls64": dict(TINY, large_size=64)}
    for tag, over in cases.items():
        fl = su.sr_model_and_diffusion_defaults()
        fl.update(over)
        model, _ = su.sr_create_model_and_diffusion(**fl)
        mine = [(k, list(v.shape)) for k, v in model.state_dict().items()]
        assert mine == [(a, b) for a, b in ref[tag]], tag
        if tag == "published":
            assert sum(p.numel() for p in model.parameters()) == ref["published_param_count"]
        if tag == "tiny":
            sd = model.state_dict()
            # zero_module'd layers (nn.py:68-74; unet.py:210-212, :996)
            assert float(sd["out.2.weight"].abs().max()) == 0 and float(sd["out.2.bias"].abs().max()) == 0
            assert float(sd["input_blocks.1.0.out_layers.3.weight"].abs().max()) == 0
            assert float(sd["input_blocks.1.0.in_layers.2.weight"].abs().max()) > 0


@pytest.mark.parametrize("tag,resp", [("full", ""), ("250", "250"), ("50", "50"), ("ddim50", "ddim50"),
                                      ("10", "10"), ("sect", "10,15,20")])
def test_diffusion_tables_exact(golden, tag, resp):
    g = golden("schedules.npz")
    d = su.create_gaussian_diffusion(steps=1000, learn_sigma=True, timestep_respacing=resp)
    assert isinstance(d, SpacedDiffusion) and d.timestep_map == list(g[tag + "/timestep_map"])
    assert d.num_timesteps == len(d.timestep_map) and d.original_num_steps == 1000
    for n in TABLES:
        assert np.array_equal(getattr(d, n), g[tag + "/" + n]), n
    tab = d.coef_table()
    assert tab.dtype == np.float32 and tab.shape == (d.num_timesteps, 8)
    assert np.array_equal(tab[:, 0], g[tag + "/sqrt_recip_alphas_cumprod"].astype(np.float32))
    assert np.array_equal(tab[:, 5], np.log(g[tag + "/betas"]).astype(np.float32))


def test_space_timesteps_and_enums():
    assert space_timesteps(1000, "ddim50") == set(range(0, 1000, 20))
    assert sorted(space_timesteps(1000, "10"))[:3] == [0, 111, 222]
    assert sorted(space_timesteps(300, [10, 15, 20]))[-1] == 299
    with pytest.raises(ValueError):
        space_timesteps(1000, "ddim999")
    with pytest.raises(ValueError):
        space_timesteps(10, "20")
    d = su.create_gaussian_diffusion(learn_sigma=True)
    assert d.model_mean_type == gd.ModelMeanType.EPSILON and d.model_var_type == gd.ModelVarType.LEARNED_RANGE
    assert d.loss_type == gd.LossType.MSE and d.rescale_timesteps is False
    d = su.create_gaussian_diffusion(learn_sigma=False, predict_xstart=True, use_kl=True)
    assert d.model_var_type == gd.ModelVarType.FIXED_LARGE and d.model_mean_type == gd.ModelMeanType.START_X
    assert d.loss_type == gd.LossType.RESCALED_KL and d.loss_type.is_vb()
    # FIXED_LARGE log-variance row (gaussian_diffusion.py:281-284)
    tab = d.coef_table()
    want = np.log(np.append(d.posterior_variance[1], d.betas[1:])).astype(np.float32)
    assert np.array_equal(tab[:, 4], want)


def test_timestep_mapping_and_wrapped_model():
    d = su.create_gaussian_diffusion(timestep_respacing="ddim50")
    t = torch.tensor([0, 1, 49])
    assert d._model_timesteps(t).tolist() == [0, 20, 980]
    d2 = su.create_gaussian_diffusion(timestep_respacing="ddim50", rescale_timesteps=True, steps=500)
    assert torch.allclose(d2._model_timesteps(t), torch.tensor([0.0, 20.0, 980.0]))  # 10-stride * 2.0
    seen = {}
    w = _WrappedModel(lambda x, ts, **kw: seen.update(ts=ts, kw=kw) or x, d.timestep_map, False, 1000)
    w(torch.zeros(3), t, low_res=1)
    assert seen["ts"].tolist() == [0, 20, 980] and seen["kw"] == {"low_res": 1}
    assert d._wrap_model(w) is w


def test_no_cpu_path():
    """The product refuses CPU tensors / CPU models instead of silently computing elsewhere."""
    fl = su.sr_model_and_diffusion_defaults()
    fl.update(TINY, timestep_respacing="2")
    model, diff = su.sr_create_model_and_diffusion(**fl)
    x = torch.zeros(1, 1, 4, 8, 8)
    with pytest.raises(RuntimeError, match="GPU"):
        model(x, torch.tensor([0]), low_res=x)
    with pytest.raises(RuntimeError, match="HIP"):
        diff.p_sample_loop(model, (1, 1, 4, 8, 8), model_kwargs={"low_res": x})
    src = open(os.path.join(ROOT, "3d-denoising-diffusion-model_amd", "guided_diffusion", "engine.py")).read()
    assert "oracle" not in src


def test_unsupported_features_fail_loudly():
    from guided_diffusion.unet import UNetModel_noatt
    with pytest.raises(NotImplementedError):
        UNetModel_noatt(32, 2, 32, 2, 1, (), dims=1)
    d = su.create_gaussian_diffusion()
    with pytest.raises(NotImplementedError):
        next(d.p_sample_loop_progressive(None, (1, 1, 2, 2, 2), cond_fn=lambda *a: 0))


def test_model2d_state_dict_layout_matches_reference():
    """create_model_and_diffusion (script_util.py:74-184): the 2-D RGB UNetModel's state_dict keys
    and shapes, in registration order, equal the reference's (tests/golden/model2d_keys.json)."""
    import json
    from conftest import GOLDEN
    with open(os.path.join(GOLDEN, "model2d_keys.json")) as f:
        ref = json.load(f)
    fl = su.model_and_diffusion_defaults()
    fl.update(image_size=64, num_channels=32, num_res_blocks=1, channel_mult="1,2,2", num_head_channels=32,
              attention_resolutions="16", learn_sigma=True, use_scale_shift_norm=True, timestep_respacing="6")
    variants = {"film": {}, "updown_additive": dict(resblock_updown=True, use_scale_shift_norm=False,
                                                    learn_sigma=False)}
    for tag, over in variants.items():
        model, diff = su.create_model_and_diffusion(**dict(fl, **over))
        assert [[k, list(v.shape)] for k, v in model.state_dict().items()] == ref[tag], tag
        assert diff.num_timesteps == 6 and model.dims == 2
        with pytest.raises(RuntimeError):          # no CPU path
            model(torch.zeros(1, 3, 32, 32), torch.zeros(1))


def test_class_cond_and_new_attention_order_layouts_match_reference():
    """class_cond adds label_emb.weight right behind time_embed (unet.py:476-478); use_new_attention_order
    changes no key: state_dict keys and shapes of three such models equal the reference's
    (tests/golden/api_extras_keys.json)."""
    import json
    from conftest import GOLDEN
    with open(os.path.join(GOLDEN, "api_extras_keys.json")) as f:
        ref = json.load(f)
    fl = su.sr_model_and_diffusion_defaults()
    fl.update(large_size=96, small_size=96, num_channels=32, num_res_blocks=1, num_head_channels=64,
              attention_resolutions="1000", learn_sigma=True, resblock_updown=True, use_scale_shift_norm=True,
              class_cond=True)
    model, _ = su.sr_create_model_and_diffusion(**fl)
    assert [[k, list(v.shape)] for k, v in model.state_dict().items()] == ref["sr_class_cond"]
    f2 = su.model_and_diffusion_defaults()
    f2.update(image_size=64, num_channels=32, num_res_blocks=1, channel_mult="1,2,2", num_head_channels=32,
              attention_resolutions="16", learn_sigma=True, use_scale_shift_norm=True, timestep_respacing="6")
    for tag, over in (("new_order", dict(use_new_attention_order=True)),
                      ("new_order_class_cond", dict(use_new_attention_order=True, class_cond=True,
                                                    num_head_channels=-1, num_heads=2))):
        model, _ = su.create_model_and_diffusion(**dict(f2, **over))
        assert model.topology.new_attention_order
        assert [[k, list(v.shape)] for k, v in model.state_dict().items()] == ref[tag], tag


def test_integration_doc_binds_the_current_abi():
    """INTEGRATION.md's condensed binding starts with `assert lib.ddpm3d_abi_version() == N`: N is the header's
    and the Python binding's ABI version (a copy-paste of the documented stub must not fail on its first line)."""
    import re
    from conftest import ROOT
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"ddpm3d_abi_version\(\) == (\d+)", doc)
    assert m, "INTEGRATION.md no longer shows the ABI assertion"
    hdr = open(os.path.join(ROOT, "include", "ddpm3d.h")).read()
    h = re.search(r"#define DDPM3D_ABI_VERSION (\d+)", hdr)
    assert int(m.group(1)) == int(h.group(1)) == _hip.ABI_VERSION
    # and the README's layout table names the same version
    readme = open(os.path.join(ROOT, "README.md")).read()
    r = re.search(r"ABI version (\d+)", readme)
    assert r and int(r.group(1)) == _hip.ABI_VERSION


def test_unet_desc_struct_layouts_match_the_header():
    """The ctypes mirrors of ddpm3d_conv_weights / ddpm3d_layer / ddpm3d_unet_desc have the field order, sizes and
    offsets a C compiler gives the header's structs (checked against a tiny C program compiled here with gcc)."""
    import subprocess
    import tempfile
    from conftest import ROOT
    src = r"""
#include <stdio.h>
#include <stddef.h>
#include "ddpm3d.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu\n", sizeof(ddpm3d_conv_weights), offsetof(ddpm3d_conv_weights, bias), offsetof(ddpm3d_conv_weights, Cout),
           offsetof(ddpm3d_conv_weights, precision_wz), (size_t)0);
    printf("%zu %zu %zu %zu %zu\n", sizeof(ddpm3d_layer), offsetof(ddpm3d_layer, norm1_gamma), offsetof(ddpm3d_layer, conv1),
           offsetof(ddpm3d_layer, conv2), offsetof(ddpm3d_layer, skip));
    printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(ddpm3d_unet_desc), offsetof(ddpm3d_unet_desc, layers),
           offsetof(ddpm3d_unet_desc, input_block_layers), offsetof(ddpm3d_unet_desc, output_block_layers),
           offsetof(ddpm3d_unet_desc, first), offsetof(ddpm3d_unet_desc, out), offsetof(ddpm3d_unet_desc, arithmetic));
    return 0;
}
"""
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "layout.c")
        with open(c, "w") as f:
            f.write(src)
        exe = os.path.join(td, "layout")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split("\n")
    W, L, U = _hip.ConvWeights, _hip.Layer, _hip.UnetDesc
    assert [int(v) for v in out[0].split()][:4] == [ctypes.sizeof(W), W.bias.offset, W.Cout.offset, W.precision_wz.offset]
    assert [int(v) for v in out[1].split()] == [ctypes.sizeof(L), L.norm1_gamma.offset, L.conv1.offset, L.conv2.offset,
                                                L.skip.offset]
    assert [int(v) for v in out[2].split()] == [ctypes.sizeof(U), U.layers.offset, U.input_block_layers.offset,
                                                U.output_block_layers.offset, U.first.offset, U.out.offset,
                                                U.arithmetic.offset]


def test_no_wide_store_is_followed_by_a_write_of_its_data_registers():
    """gfx950 reads the first data register of an 8- / 16-byte VMEM store late, and hipcc leaves the classic
    store-data hazard unprotected when the store's soffset is an SGPR: a VALU write placed right behind such a store
    overtook it in r04's lean epilogue (profiles/r04_store_data_hazard_plain.txt; the GPU-side regression test is
    test_gpu_ops.py::test_conv3d_wide_epilogue_stores_are_repeatable).  Host-side guard: no kernel of the built library
    may contain a wide buffer / global / flat store whose data registers a VALU instruction rewrites within the next
    two instructions (tools/check_store_hazard.py disassembles every code object of libddpm3d.so)."""
    import subprocess
    import sys
    from conftest import ROOT
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("no llvm-objdump here")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_store_hazard.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " 0 wide stores" in r.stdout
