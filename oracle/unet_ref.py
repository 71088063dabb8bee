"""
ORACLE (test infrastructure): the reference's 3-D UNet forward as a pure
function of (state_dict, config, x, timesteps, low_res), torch CPU fp32.

Follows unet.py:751-997 (constructor bookkeeping -> topology()),
unet.py:1015-1044 (forward), unet.py:236-256 (ResBlock), unet.py:102-105 /
:129-136 (H,W-only up/down-sampling for dims=3), unet.py:296-305 + :337-354
(AttentionBlock with the legacy head layout), nn.py:93-121 (GroupNorm32,
timestep_embedding), unet.py:1687-1694 (SuperRes concat of low_res) and
script_util.py:334-450 (flag -> architecture mapping).  cfg["dims"] = 2 gives
the 2-D network of create_model (script_util.py:130-184: UNetModel, dims=2,
RGB in, mid-block attention): same topology, Conv2d / 2x2 pooling.
"""

import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# flags -> config  (script_util.py:334-450, live return at :432-450)
# --------------------------------------------------------------------------
def sr_config(large_size=256, num_channels=128, num_res_blocks=2, learn_sigma=False,
              attention_resolutions="16,8", num_heads=4, num_head_channels=-1,
              num_heads_upsample=-1, use_scale_shift_norm=True, resblock_updown=False,
              mid_attention=False, class_cond=False, **_ignored):
    if large_size in (512, 256):
        mult = (1, 1, 2, 2, 4, 4)
    elif large_size == 64:
        mult = (1, 2, 3, 4)
    else:
        mult = (1, 1, 2, 3, 4)
    return dict(
        in_channels=2,
        model_channels=num_channels,
        out_channels=2 if learn_sigma else 1,
        num_res_blocks=num_res_blocks,
        attention_ds=tuple(large_size // int(r) for r in attention_resolutions.split(",")),
        channel_mult=mult,
        num_heads=num_heads,
        num_head_channels=num_head_channels,
        num_heads_upsample=num_heads if num_heads_upsample == -1 else num_heads_upsample,
        use_scale_shift_norm=use_scale_shift_norm,
        resblock_updown=resblock_updown,
        mid_attention=mid_attention,  # False = UNetModel_noatt, True = UNetModel
        num_classes=1000 if class_cond else None,   # script_util.py:8, :442
    )


def model2d_config(image_size=64, num_channels=128, num_res_blocks=2, channel_mult="", learn_sigma=False,
                   attention_resolutions="16", num_heads=1, num_head_channels=-1, num_heads_upsample=-1,
                   use_scale_shift_norm=False, resblock_updown=False, class_cond=False,
                   use_new_attention_order=False, **_ignored):
    """script_util.py:130-184 (create_model): the 2-D RGB UNetModel."""
    if channel_mult == "":
        mult = {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4),
                64: (1, 2, 3, 4)}[image_size]
    else:
        mult = tuple(int(m) for m in channel_mult.split(","))
    return dict(
        dims=2, in_channels=3, model_channels=num_channels, out_channels=6 if learn_sigma else 3,
        num_res_blocks=num_res_blocks,
        attention_ds=tuple(image_size // int(r) for r in attention_resolutions.split(",")),
        channel_mult=mult, num_heads=num_heads, num_head_channels=num_head_channels,
        num_heads_upsample=num_heads if num_heads_upsample == -1 else num_heads_upsample,
        use_scale_shift_norm=use_scale_shift_norm, resblock_updown=resblock_updown, mid_attention=True,
        num_classes=1000 if class_cond else None, new_attention_order=use_new_attention_order)


# --------------------------------------------------------------------------
# topology: which layers exist, under which state_dict prefixes
# --------------------------------------------------------------------------
def topology(cfg):
    """Lists of blocks; a block is a list of (prefix, kind, params...)."""
    mc = cfg["model_channels"]
    mult = cfg["channel_mult"]
    nrb = cfg["num_res_blocks"]
    att = cfg["attention_ds"]
    updown = cfg["resblock_updown"]

    def heads(ch, n):
        return n if cfg["num_head_channels"] == -1 else ch // cfg["num_head_channels"]

    ch = int(mult[0] * mc)
    inp = [[("input_blocks.0.0", "conv", cfg["in_channels"], ch)]]
    chans = [ch]
    ds = 1
    for level, m in enumerate(mult):
        for _ in range(nrb):
            k = len(inp)
            co = int(m * mc)
            blk = [("input_blocks.%d.0" % k, "res", ch, co, None)]
            ch = co
            if ds in att:
                blk.append(("input_blocks.%d.1" % k, "attn", ch, heads(ch, cfg["num_heads"])))
            inp.append(blk)
            chans.append(ch)
        if level != len(mult) - 1:
            k = len(inp)
            if updown:
                inp.append([("input_blocks.%d.0" % k, "res", ch, ch, "down")])
            else:
                inp.append([("input_blocks.%d.0" % k, "downconv", ch)])
            chans.append(ch)
            ds *= 2

    mid = [("middle_block.0", "res", ch, ch, None)]
    if cfg["mid_attention"]:
        mid.append(("middle_block.1", "attn", ch, heads(ch, cfg["num_heads"])))
        mid.append(("middle_block.2", "res", ch, ch, None))
    else:
        mid.append(("middle_block.1", "res", ch, ch, None))

    out = []
    outch = ch
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nrb + 1):
            # unet.py:947-951: pop the skip's width, then pop AGAIN for the
            # block's output width, and push that back (:990).
            inch = chans.pop()
            outch = chans.pop() if chans else inch
            k = len(out)
            blk = [("output_blocks.%d.0" % k, "res", inch * 2, outch, None)]
            j = 1
            if ds in att:
                blk.append(("output_blocks.%d.%d" % (k, j), "attn", outch,
                            heads(outch, cfg["num_heads_upsample"])))
                j += 1
            if level and i == nrb:
                if updown:
                    blk.append(("output_blocks.%d.%d" % (k, j), "res", outch, outch, "up"))
                else:
                    blk.append(("output_blocks.%d.%d" % (k, j), "upconv", outch))
                ds //= 2
            out.append(blk)
            chans.append(outch)
    return dict(input=inp, middle=mid, output=out, final_ch=outch,
                input_ch=int(mult[0] * mc))


def param_shapes(cfg):
    """(key, shape) for every state_dict entry, in module-registration order."""
    mc = cfg["model_channels"]
    ted = 4 * mc
    res = [("time_embed.0.weight", (ted, mc)), ("time_embed.0.bias", (ted,)),
           ("time_embed.2.weight", (ted, ted)), ("time_embed.2.bias", (ted,))]
    if cfg.get("num_classes") is not None:
        res.append(("label_emb.weight", (cfg["num_classes"], ted)))     # unet.py:476-478
    topo = topology(cfg)
    k3 = (3,) * cfg.get("dims", 3)
    k1 = (1,) * cfg.get("dims", 3)

    def layer(entry):
        p, kind = entry[0], entry[1]
        if kind == "conv":
            _, _, ci, co = entry
            return [(p + ".weight", (co, ci) + k3), (p + ".bias", (co,))]
        if kind == "res":
            _, _, ci, co, _ud = entry
            e = 2 * co if cfg["use_scale_shift_norm"] else co
            r = [(p + ".in_layers.0.weight", (ci,)), (p + ".in_layers.0.bias", (ci,)),
                 (p + ".in_layers.2.weight", (co, ci) + k3), (p + ".in_layers.2.bias", (co,)),
                 (p + ".emb_layers.1.weight", (e, ted)), (p + ".emb_layers.1.bias", (e,)),
                 (p + ".out_layers.0.weight", (co,)), (p + ".out_layers.0.bias", (co,)),
                 (p + ".out_layers.3.weight", (co, co) + k3), (p + ".out_layers.3.bias", (co,))]
            if ci != co:
                r += [(p + ".skip_connection.weight", (co, ci) + k1),
                      (p + ".skip_connection.bias", (co,))]
            return r
        if kind == "attn":
            _, _, c, _h = entry
            return [(p + ".norm.weight", (c,)), (p + ".norm.bias", (c,)),
                    (p + ".qkv.weight", (3 * c, c, 1)), (p + ".qkv.bias", (3 * c,)),
                    (p + ".proj_out.weight", (c, c, 1)), (p + ".proj_out.bias", (c,))]
        if kind == "downconv":
            c = entry[2]
            return [(p + ".op.weight", (c, c) + k3), (p + ".op.bias", (c,))]
        if kind == "upconv":
            c = entry[2]
            return [(p + ".conv.weight", (c, c) + k3), (p + ".conv.bias", (c,))]
        raise ValueError(kind)

    for blk in topo["input"]:
        for e in blk:
            res += layer(e)
    for e in topo["middle"]:
        res += layer(e)
    for blk in topo["output"]:
        for e in blk:
            res += layer(e)
    res += [("out.0.weight", (topo["final_ch"],)), ("out.0.bias", (topo["final_ch"],)),
            ("out.2.weight", (cfg["out_channels"], topo["input_ch"]) + k3),
            ("out.2.bias", (cfg["out_channels"],))]
    return res


# --------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------
def timestep_embedding(t, dim, max_period=10000):
    # nn.py:103-121
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def gn32(sd, p, x):
    # nn.py:17-19, :93-100: 32 groups, eps 1e-5, computed in fp32
    return F.group_norm(x.float(), 32, sd[p + ".weight"], sd[p + ".bias"], 1e-5).type(x.dtype)


def conv(x, w, b, **kw):
    # nn.py:28-38 conv_nd
    return (F.conv3d if x.dim() == 5 else F.conv2d)(x, w, b, **kw)


def pool_hw(x):
    # unet.py:129-136: AvgPool3d kernel=stride=(1,2,2); dims=2: 2x2
    if x.dim() == 4:
        return F.avg_pool2d(x, 2, 2)
    return F.avg_pool3d(x, (1, 2, 2), (1, 2, 2))


def up_hw(x):
    # unet.py:102-105: nearest to (D, 2H, 2W); dims=2: scale_factor 2
    if x.dim() == 4:
        return F.interpolate(x, scale_factor=2, mode="nearest")
    return F.interpolate(x, (x.shape[2], x.shape[3] * 2, x.shape[4] * 2), mode="nearest")


def resblock(sd, p, x, emb, updown, film):
    # unet.py:236-256
    h = F.silu(gn32(sd, p + ".in_layers.0", x))
    if updown == "down":
        h, x = pool_hw(h), pool_hw(x)
    elif updown == "up":
        h, x = up_hw(h), up_hw(x)
    h = conv(h, sd[p + ".in_layers.2.weight"], sd[p + ".in_layers.2.bias"], padding=1)
    e = F.linear(F.silu(emb), sd[p + ".emb_layers.1.weight"], sd[p + ".emb_layers.1.bias"])
    e = e[(..., ) + (None,) * (x.dim() - 2)]
    if film:
        scale, shift = torch.chunk(e, 2, dim=1)
        h = gn32(sd, p + ".out_layers.0", h) * (1 + scale) + shift
        h = F.silu(h)
    else:
        h = F.silu(gn32(sd, p + ".out_layers.0", h + e))
    h = conv(h, sd[p + ".out_layers.3.weight"], sd[p + ".out_layers.3.bias"], padding=1)
    if (p + ".skip_connection.weight") in sd:
        x = conv(x, sd[p + ".skip_connection.weight"], sd[p + ".skip_connection.bias"])
    return x + h


# Query rows per block of the attention product.  The reference materialises the whole (T x T) weight
# matrix (unet.py:349-353); at BASELINE config 5's T = 32 768 that is 4.3 GB per head, so the oracle walks the
# QUERY axis in blocks.  A softmax row depends on its own query only, so every row is computed from exactly the
# same numbers as in the materialised form (tests/test_oracle_golden.py pins the blocked form to the
# reference's own `tiny_attn` outputs with a block smaller than T).
ATTN_QUERY_BLOCK = 2048


def qkv_attention(q, k, v, block=None, dtype=None):
    """softmax((q s)^T (k s)) v per (batch x head) row of q, k, v [B, ch, T]; s = ch^-1/4 (unet.py:346-353).
    block: query rows per pass (None = ATTN_QUERY_BLOCK).  dtype float64 = an fp64 evaluation for tests."""
    ch, length = q.shape[1], q.shape[2]
    s = 1 / math.sqrt(math.sqrt(ch))
    block = block or ATTN_QUERY_BLOCK
    if dtype is not None:
        q, k, v = q.to(dtype), k.to(dtype), v.to(dtype)
    out = torch.empty_like(q)
    ks = k * s
    for t0 in range(0, length, block):
        w = torch.einsum("bct,bcs->bts", q[:, :, t0:t0 + block] * s, ks)
        w = torch.softmax(w.float() if dtype is None else w, dim=-1).type(w.dtype)
        out[:, :, t0:t0 + block] = torch.einsum("bts,bcs->bct", w, v)
    return out


def attention(sd, p, x, n_heads, new_order=False):
    # unet.py:296-305; QKVAttentionLegacy :337-354, or (new_order) QKVAttention :361-389
    b, c = x.shape[:2]
    spatial = x.shape[2:]
    xf = x.reshape(b, c, -1)
    qkv = F.conv1d(gn32(sd, p + ".norm", xf), sd[p + ".qkv.weight"], sd[p + ".qkv.bias"])
    length = qkv.shape[-1]
    ch = c // n_heads
    if new_order:
        q, k, v = (t.reshape(b * n_heads, ch, length) for t in qkv.chunk(3, dim=1))
    else:
        q, k, v = qkv.reshape(b * n_heads, ch * 3, length).split(ch, dim=1)
    a = qkv_attention(q, k, v).reshape(b, -1, length)
    h = F.conv1d(a, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    return (xf + h).reshape(b, c, *spatial)


def run_layer(sd, cfg, entry, h, emb):
    p, kind = entry[0], entry[1]
    if kind == "conv":
        return conv(h, sd[p + ".weight"], sd[p + ".bias"], padding=1)
    if kind == "res":
        return resblock(sd, p, h, emb, entry[4], cfg["use_scale_shift_norm"])
    if kind == "attn":
        return attention(sd, p, h, entry[3], cfg.get("new_attention_order", False))
    if kind == "downconv":
        return conv(h, sd[p + ".op.weight"], sd[p + ".op.bias"], stride=(1, 2, 2) if h.dim() == 5 else 2,
                    padding=1)
    if kind == "upconv":
        return conv(up_hw(h), sd[p + ".conv.weight"], sd[p + ".conv.bias"], padding=1)
    raise ValueError(kind)


def unet_forward(sd, cfg, x, timesteps, low_res=None, taps=None, y=None):
    """x: (N,1,D,H,W); low_res same shape (SuperRes concat, unet.py:1690-1693);
    returns (N,out_channels,D,H,W).  ``taps``: optional dict that receives
    named intermediate activations for layer-level parity tests."""
    if low_res is not None:
        x = torch.cat([x, low_res.clone()], dim=1)
    topo = topology(cfg)
    mc = cfg["model_channels"]
    emb = timestep_embedding(timesteps, mc)
    emb = F.linear(emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    emb = F.linear(F.silu(emb), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    assert (y is not None) == (cfg.get("num_classes") is not None)      # unet.py:696-698
    if y is not None:
        emb = emb + sd["label_emb.weight"][y]                            # unet.py:703-705
    h = x.float()
    hs = []
    for k, blk in enumerate(topo["input"]):
        for e in blk:
            h = run_layer(sd, cfg, e, h, emb)
        hs.append(h)
        if taps is not None:
            taps["input_blocks.%d" % k] = h
    for e in topo["middle"]:
        h = run_layer(sd, cfg, e, h, emb)
    if taps is not None:
        taps["middle_block"] = h
    for k, blk in enumerate(topo["output"]):
        h = torch.cat([h, hs.pop()], dim=1)
        for e in blk:
            h = run_layer(sd, cfg, e, h, emb)
        if taps is not None:
            taps["output_blocks.%d" % k] = h
    h = F.silu(gn32(sd, "out.0", h))
    return conv(h, sd["out.2.weight"], sd["out.2.bias"], padding=1)
