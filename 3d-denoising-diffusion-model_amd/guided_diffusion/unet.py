"""
The reference's UNet family (unet.py:396-1044, :1655-1694) as parameter
containers whose forward pass is a compiled HIP launch plan (engine.py).

What is kept from the reference so that this is a drop-in:
  * class names and constructor signatures (UNetModel, UNetModel_noatt,
    SuperResModel, SuperResModel_noatt),
  * the state_dict key scheme (time_embed.{0,2}, input_blocks.K.J.in_layers.{0,2},
    emb_layers.1, out_layers.{0,3}, skip_connection, middle_block.J,
    output_blocks.K.J, out.{0,2}) so reference checkpoints load unchanged,
  * initialisation semantics (torch defaults + zeroed second conv / proj_out /
    final conv, nn.py:68-74),
  * forward(x, timesteps, low_res=...) -> (N, out_channels, D, H, W),
    convert_to_fp16 / convert_to_fp32, .dtype.

What is different: no module here computes anything with ATen.  The layer
list is flattened into `Topology`, and `forward` hands x to UNetEngine.
"""

import os
import warnings
from collections import namedtuple

import torch
import torch.nn as nn

from . import _hip as H
from .engine import UNetEngine

Layer = namedtuple("Layer", "prefix kind cin cout updown heads")


class Topology:
    """Flat description of the network: which layers exist, their state_dict
    prefixes and channel widths.  Mirrors the bookkeeping of the reference
    constructor (unet.py:805-991), including the decoder's pop-pop-push
    width rule (unet.py:947-951, :990)."""

    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 channel_mult, num_heads, num_head_channels, num_heads_upsample, resblock_updown,
                 mid_attention, new_attention_order=False):
        # QKVAttention instead of QKVAttentionLegacy in every AttentionBlock (unet.py:287-292)
        self.new_attention_order = bool(new_attention_order)

        def nheads(ch, n):
            return n if num_head_channels == -1 else ch // num_head_channels

        ch = int(channel_mult[0] * model_channels)
        self.input = [[Layer("input_blocks.0.0", "conv", in_channels, ch, None, 0)]]
        widths = [ch]
        ds = 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                k = len(self.input)
                co = int(mult * model_channels)
                blk = [Layer("input_blocks.%d.0" % k, "res", ch, co, None, 0)]
                ch = co
                if ds in attention_resolutions:
                    blk.append(Layer("input_blocks.%d.1" % k, "attn", ch, ch, None, nheads(ch, num_heads)))
                self.input.append(blk)
                widths.append(ch)
            if level != len(channel_mult) - 1:
                k = len(self.input)
                kind = ("res", "down") if resblock_updown else ("downconv", None)
                self.input.append([Layer("input_blocks.%d.0" % k, kind[0], ch, ch, kind[1], 0)])
                widths.append(ch)
                ds *= 2
        self.middle = [Layer("middle_block.0", "res", ch, ch, None, 0)]
        if mid_attention:
            self.middle.append(Layer("middle_block.1", "attn", ch, ch, None, nheads(ch, num_heads)))
            self.middle.append(Layer("middle_block.2", "res", ch, ch, None, 0))
        else:
            self.middle.append(Layer("middle_block.1", "res", ch, ch, None, 0))
        self.output = []
        outch = ch
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                inch = widths.pop()
                outch = widths.pop() if widths else inch
                k = len(self.output)
                blk = [Layer("output_blocks.%d.0" % k, "res", 2 * inch, outch, None, 0)]
                if ds in attention_resolutions:
                    blk.append(Layer("output_blocks.%d.%d" % (k, len(blk)), "attn", outch, outch, None,
                                     nheads(outch, num_heads_upsample)))
                if level and i == num_res_blocks:
                    kind = ("res", "up") if resblock_updown else ("upconv", None)
                    blk.append(Layer("output_blocks.%d.%d" % (k, len(blk)), kind[0], outch, outch, kind[1], 0))
                    ds //= 2
                self.output.append(blk)
                widths.append(outch)
        self.final_ch = outch
        self.input_ch = int(channel_mult[0] * model_channels)
        self.out_channels = out_channels

    def all_layers(self):
        for blk in self.input:
            yield from blk
        yield from self.middle
        for blk in self.output:
            yield from blk


# ---------------------------------------------------------------- containers
class GroupNorm32(nn.GroupNorm):
    """32-group norm parameters (nn.py:17-19); applied inside the conv prologue."""


def _zero(m):
    for p in m.parameters():
        p.detach().zero_()
    return m


def _conv(dims, cin, cout, k, **kw):
    """nn.py:22-32 conv_nd: parameter container of a 2-D or 3-D convolution"""
    return (nn.Conv2d if dims == 2 else nn.Conv3d)(cin, cout, k, **kw)


class ResBlock(nn.Module):
    """Parameters of one residual block (unet.py:143-256)."""

    def __init__(self, channels, emb_channels, out_channels, use_scale_shift_norm, dims=3):
        super().__init__()
        self.in_layers = nn.Sequential(GroupNorm32(32, channels), nn.SiLU(),
                                       _conv(dims, channels, out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(
            nn.SiLU(), nn.Linear(emb_channels, 2 * out_channels if use_scale_shift_norm else out_channels))
        self.out_layers = nn.Sequential(GroupNorm32(32, out_channels), nn.SiLU(), nn.Dropout(p=0.0),
                                        _zero(_conv(dims, out_channels, out_channels, 3, padding=1)))
        self.skip_connection = (nn.Identity() if out_channels == channels
                                else _conv(dims, channels, out_channels, 1))


class AttentionBlock(nn.Module):
    """Parameters of one attention block (unet.py:259-305)."""

    def __init__(self, channels):
        super().__init__()
        self.norm = GroupNorm32(32, channels)
        self.qkv = nn.Conv1d(channels, 3 * channels, 1)
        self.proj_out = _zero(nn.Conv1d(channels, channels, 1))


class Downsample(nn.Module):
    def __init__(self, channels, dims=3):
        super().__init__()
        self.op = _conv(dims, channels, channels, 3, stride=(2 if dims == 2 else (1, 2, 2)), padding=1)


class Upsample(nn.Module):
    def __init__(self, channels, dims=3):
        super().__init__()
        self.conv = _conv(dims, channels, channels, 3, padding=1)


class _Block(nn.Sequential):
    """A TimestepEmbedSequential-shaped holder (indices are state_dict key parts)."""


def _container(layer, ted, film, dims=3):
    if layer.kind == "conv":
        return _conv(dims, layer.cin, layer.cout, 3, padding=1)
    if layer.kind == "res":
        return ResBlock(layer.cin, ted, layer.cout, film, dims)
    if layer.kind == "attn":
        return AttentionBlock(layer.cin)
    if layer.kind == "downconv":
        return Downsample(layer.cin, dims)
    if layer.kind == "upconv":
        return Upsample(layer.cin, dims)
    raise ValueError(layer.kind)


class UNetModel_noatt(nn.Module):
    """unet.py:720-1044 (the class scripts/test.py actually builds, via
    SuperResModel_noatt): no attention in the middle block."""

    MID_ATTENTION = False

    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks,
                 attention_resolutions, dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2,
                 num_classes=None, use_checkpoint=False, use_fp16=False, num_heads=1, num_head_channels=-1,
                 num_heads_upsample=-1, use_scale_shift_norm=False, resblock_updown=False,
                 use_new_attention_order=False):
        super().__init__()
        if dims not in (2, 3):
            raise NotImplementedError("the HIP engine implements the 2-D and 3-D models (dims 2 or 3)")
        # dropout: nn.Dropout is the identity in eval mode (unet.py:209), and this package only
        # samples -- any p is accepted and ignored
        if not conv_resample and not resblock_updown:
            raise NotImplementedError("conv_resample=False")
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        self.image_size = image_size
        self.in_channels = in_channels
        self.model_channels = model_channels
        self.out_channels = out_channels
        self.num_res_blocks = num_res_blocks
        self.attention_resolutions = attention_resolutions
        self.dropout = dropout
        self.channel_mult = channel_mult
        self.conv_resample = conv_resample
        self.num_classes = num_classes
        self.use_checkpoint = use_checkpoint
        self.dtype = torch.float16 if use_fp16 else torch.float32
        self.num_heads = num_heads
        self.num_head_channels = num_head_channels
        self.num_heads_upsample = num_heads_upsample
        self.use_scale_shift_norm = use_scale_shift_norm
        self.resblock_updown = resblock_updown
        self.use_fp16 = use_fp16
        self.dims = dims

        self.topology = Topology(in_channels, model_channels, out_channels, num_res_blocks,
                                 tuple(attention_resolutions), tuple(channel_mult), num_heads,
                                 num_head_channels, num_heads_upsample, resblock_updown, self.MID_ATTENTION,
                                 use_new_attention_order)
        ted = 4 * model_channels
        t = self.topology
        self.time_embed = nn.Sequential(nn.Linear(model_channels, ted), nn.SiLU(), nn.Linear(ted, ted))
        if num_classes is not None:
            self.label_emb = nn.Embedding(num_classes, ted)     # unet.py:476-478
        self.input_blocks = nn.ModuleList(
            [_Block(*[_container(l, ted, use_scale_shift_norm, dims) for l in blk]) for blk in t.input])
        self.middle_block = _Block(*[_container(l, ted, use_scale_shift_norm, dims) for l in t.middle])
        self.output_blocks = nn.ModuleList(
            [_Block(*[_container(l, ted, use_scale_shift_norm, dims) for l in blk]) for blk in t.output])
        self.out = nn.Sequential(GroupNorm32(32, t.final_ch), nn.SiLU(),
                                 _zero(_conv(dims, t.input_ch, out_channels, 3, padding=1)))
        self._engine = None
        self._engine_key = None
        # arithmetic of the 3x3x3 convolutions' products: "f16x3" (default: each fp32
        # product = three f16 MFMAs on hi/lo-split operands; error vs fp64 within 2x of
        # the exact path, tests/test_gpu_ops.py) or "f32" (exact fp32 MFMA, 2.4x slower)
        self.conv_precision = os.environ.get("DDPM3D_PRECISION", "f16x3")
        # replay one captured hipGraph per forward instead of ~230 launches issued from Python (engine.py:
        # _Plan._replay); same kernels and arguments, bit-identical results.  DDPM3D_STEP_GRAPH=0/1 overrides.
        self.step_graph = os.environ.get("DDPM3D_STEP_GRAPH", "0") == "1"
        # run on the C-level launch plan (ddpm3d_unet_plan_create / ddpm3d_unet_forward) instead of engine.py's
        # Python one: the same calls in the same order, bit-identical.  DDPM3D_NATIVE_PLAN=0/1 overrides.
        self.native_plan = os.environ.get("DDPM3D_NATIVE_PLAN", "0") == "1"

    # ---- precision switches (unet.py:999-1013) -------------------------------
    def convert_to_fp16(self):
        """The reference halves the conv weights of the torso and feeds fp16 activations
        (fp32 GroupNorm, fp32 time-embed / emb_layers / final conv; unet.py:999-1005, :1035).
        Here: every conv's operands are rounded to f16 on their way into the matrix cores
        (one f16 MFMA per product) and the residual stream -- every tensor the torso's convs
        write and read -- is STORED in f16 (ddpm3d_conv_desc.io_dtype, DDPM3D_IO_HALF_IS_F16);
        accumulation, GroupNorm statistics, the timestep path, the network's input and its
        output stay fp32.  Parameters keep their fp32 storage (`state_dict` is unchanged)."""
        self.dtype = torch.float16
        self.conv_precision = "f16"

    def convert_to_bf16(self):
        """BASELINE config 4's arithmetic (no counterpart in the reference, which has fp16 only):
        bf16 operands on the matrix cores and the residual stream stored in bf16, with the fp16
        mode's placement otherwise -- GroupNorm statistics, timestep path, first conv's inputs and
        the final conv's output in fp32 (unet.py:999-1005, :1035, :1043)."""
        self.dtype = torch.bfloat16
        self.conv_precision = "bf16"

    def convert_to_fp32(self):
        self.dtype = torch.float32
        self.conv_precision = os.environ.get("DDPM3D_PRECISION", "f16x3")

    # ---- engine management ----------------------------------------------------
    def _params_key(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def engine(self):
        p0 = next(self.parameters())
        if not p0.is_cuda:
            raise RuntimeError("model parameters are on %s: move the model to the GPU (model.to('cuda')); "
                               "this package has no CPU path" % p0.device)
        key = (self.conv_precision,) + self._params_key()
        if self._engine is None or self._engine_key != key:
            params = {k: v.detach().float().contiguous() for k, v in self.state_dict().items()}
            with torch.cuda.device(p0.device):
                self._engine = UNetEngine(self.topology, params, self.model_channels,
                                          self.use_scale_shift_norm, p0.device, self.conv_precision,
                                          in_channels=self.in_channels, planar=self._planar(),
                                          winograd=self.dims == 3, step_graph=self.step_graph)
            self._engine_key = key
        self._engine.step_graph = self.step_graph
        self._engine.native_plan = self.native_plan
        return self._engine

    PLANAR_INPUT = False   # SuperRes models: the first conv reads x and low_res as two planes

    def _planar(self):
        """The two-pointer first conv serves the case every caller in the reference has: one image
        channel + one low_res channel.  A SuperRes model with more channels (the reference's RGB
        super-resolution networks) concatenates at the API edge like unet.py:1693 does."""
        return self.PLANAR_INPUT and self.in_channels == 2

    def forward(self, x, timesteps, y=None, low_res=None):
        """unet.py:687-716 / :1015-1044.  x: (N, in_channels, [D,] H, W).  The SuperRes subclasses
        pass low_res, which supplies the second input channel (unet.py:1687-1694)."""
        assert (y is not None) == (self.num_classes is not None), \
            "must specify y if and only if the model is class-conditional"
        if self.PLANAR_INPUT and low_res is None:
            raise RuntimeError("the super-resolution model is conditional: pass low_res=... (unet.py:1687)")
        if not self.PLANAR_INPUT and low_res is not None:
            raise RuntimeError("low_res is an argument of the SuperRes models only")
        H.require_device(x, "x")
        eng = self.engine()
        flat = x.dim() == 4                       # dims=2: (N, C, H, W) runs as depth-1 volumes
        if flat != (self.dims == 2):
            raise RuntimeError("a dims=%d model takes %d-D tensors" % (self.dims, self.dims + 2))
        with torch.cuda.device(x.device):
            yy = None
            if y is not None:
                assert tuple(y.shape) == (x.shape[0],)
                yy = y.to(device=x.device, dtype=torch.int64).contiguous()
                # nn.Embedding's own check, once per label tensor (a sampling loop passes the same `y` at every
                # step: one host round trip per loop, not per step; the kernel itself never reads out of range)
                seen = (y.data_ptr(), y._version, tuple(y.shape), y.device)
                if getattr(self, "_labels_checked", None) != seen:
                    if int(yy.min()) < 0 or int(yy.max()) >= self.num_classes:
                        raise IndexError("class label out of range [0, %d)" % self.num_classes)
                    self._labels_checked = seen
            rows = eng.film_rows(timesteps.to(device=x.device, dtype=torch.float32).contiguous(), yy)
            xv = x.unsqueeze(2) if flat else x
            lr = None
            if low_res is not None:
                lr = low_res.to(x.device).contiguous()
                lr = lr.unsqueeze(2) if flat else lr
                if not self._planar():
                    # more than one image channel: th.cat([x, low_res], dim=1) (unet.py:1693) at the edge
                    if lr.shape[0] != xv.shape[0] or lr.shape[2:] != xv.shape[2:]:
                        raise RuntimeError("low_res %s does not match x %s" % (tuple(lr.shape), tuple(xv.shape)))
                    xv = torch.cat([xv, lr.to(xv.dtype)], dim=1).contiguous()
                    lr = None
                elif xv.shape[1] != 1 or lr.shape[1] != 1:
                    raise RuntimeError("x and low_res must have one channel each (got %d and %d)"
                                       % (xv.shape[1], lr.shape[1]))
            out = eng.forward(xv, lr, rows, eng.film_total).clone()
            return out.squeeze(2) if flat else out


class UNetModel(UNetModel_noatt):
    """unet.py:396-716: the same network with an AttentionBlock between the two
    middle ResBlocks."""
    MID_ATTENTION = True


class SuperResModel_noatt(UNetModel_noatt):
    """unet.py:1676-1694: conditions on `low_res` by channel concatenation (the
    concat itself is virtual: the first conv reads the two volumes directly)."""
    PLANAR_INPUT = True

    def __init__(self, image_size, in_channels, *args, **kwargs):
        super().__init__(image_size, int(in_channels * 2), *args, **kwargs)

    def forward(self, x, timesteps, low_res=None, **kwargs):
        return super().forward(x, timesteps, low_res=low_res, **kwargs)   # y=... passes through (unet.py:1694)


class SuperResModel(UNetModel):
    """unet.py:1655-1673."""
    PLANAR_INPUT = True

    def __init__(self, image_size, in_channels, *args, **kwargs):
        super().__init__(image_size, int(in_channels * 2), *args, **kwargs)

    def forward(self, x, timesteps, low_res=None, **kwargs):
        return super().forward(x, timesteps, low_res=low_res, **kwargs)
