"""
Launch-plan compiler for the 3-D UNet forward.

A model (its layer list from unet.py and its parameters) plus an input shape
(N, D, H, W) is compiled ONCE into a flat list of C-ABI calls with every
buffer, descriptor and statistics row count fixed; a forward pass replays the
list.  What the reference does as ~600 ATen calls per step
(unet.py:1015-1044, :236-256) becomes, per ResBlock,

    gn_finalize -> conv3d(GN+SiLU [+pool/up] prologue, stats epilogue)
    gn_finalize(FiLM) -> [conv3d k=1 skip] -> conv3d(GN*(1+s)+t, SiLU prologue;
                                                      + residual, stats epilogue)

Activations live channels-last ([N][D][H][W][C], fp32).  The timestep path
(timestep_embedding -> time_embed -> every ResBlock's emb_layers) does not
depend on x, so all emb_layers are fused into one Linear whose output row
("film row") is either computed per call or, in the samplers, precomputed
for every step of the schedule.
"""

import collections
import ctypes as C
import math

import torch

from . import _hip as H


class Act:
    """An NDHWC activation produced by a conv epilogue (fp32, or bf16 in the bf16 mode), with its
    GN partial sums."""
    __slots__ = ("buf", "C", "D", "H", "W", "stats", "rows")

    def __init__(self, buf, Cn, D, Hh, W, stats, rows):
        self.buf, self.C, self.D, self.H, self.W, self.stats, self.rows = buf, Cn, D, Hh, W, stats, rows

    @property
    def half(self):
        """stored in 16 bits (bf16 in the bf16 mode, IEEE f16 in the f16 mode)"""
        return self.buf.dtype in (torch.bfloat16, torch.float16)

    @property
    def voxels(self):
        return self.D * self.H * self.W


class PackedConv:
    __slots__ = ("w", "b", "Cout", "Cin", "k", "precision", "wz")

    def __init__(self, weight, bias, precision, stream):
        lib = H.load()
        self.wz = None   # optional Winograd-along-depth packing of the same layer
        Cout, Cin, k = weight.shape[0], weight.shape[1], weight.shape[2]
        n = lib.ddpm3d_packed_weight_bytes(Cout, Cin, k, precision)
        if n == 0:
            raise RuntimeError("unsupported conv weight shape %s" % (tuple(weight.shape),))
        w32 = weight.detach().float().contiguous()
        self.w = torch.empty(n, dtype=torch.uint8, device=weight.device)
        H.check(lib.ddpm3d_pack_conv_weight(H.ptr(w32), Cout, Cin, k, precision, H.ptr(self.w), stream))
        self.b = bias.detach().float().contiguous()
        self.Cout, self.Cin, self.k, self.precision = Cout, Cin, k, precision


class UNetEngine:
    """Executes one model on one device.  `layers` is unet.py's topology."""

    def __init__(self, topo, params, model_channels, film, device, precision="f32", in_channels=2, planar=True,
                 winograd=True, step_graph=False):
        """precision: "f32" = exact fp32 MFMA everywhere; "f16x3" = every conv evaluates
        each fp32 product as three f16 MFMAs (fp32-equivalent accuracy, see
        include/ddpm3d.h); "f16" = one f16 MFMA per product (the reference's --use_fp16);
        "bf16" = one bf16 MFMA per product AND the residual stream stored in bf16."""
        if precision not in H.PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(H.PRECISIONS))
        self.precision = precision
        # replay a captured hipGraph of the forward instead of issuing its launches one by one (_Plan._replay)
        self.step_graph = step_graph
        # run forwards on the C-level plan (ddpm3d_unet_plan: the same launch list compiled by libddpm3d itself)
        # instead of this module's Python one; the Python plan stays the one that is instrumented (bench.py)
        self.native_plan = False
        self.native_plans = collections.OrderedDict()
        self._native_desc = None
        self.lib = H.load()
        self.topo = topo
        self.device = device
        self.mc = model_channels
        self.film = film  # use_scale_shift_norm
        self.p = params    # name -> device fp32 tensor
        self.plans = collections.OrderedDict()   # (N, D, H, W) -> _Plan, least recently used first
        self.split_above = {}                    # (D, H, W) -> largest batch one launch can address
        st = H.stream()
        self.conv = {}
        self.winograd = winograd
        # planar: the SuperRes first conv reads x and low_res as two single-channel volumes; otherwise
        # the input is an ordinary (N, in_channels, ...) tensor, padded to 16 channels at the edge
        if planar and in_channels != 2:
            raise ValueError("the planar first conv reads ONE image and ONE low_res channel (got in_channels=%d)"
                             % in_channels)
        self.planar = planar
        self.in_channels = in_channels
        self.cin_pad = (in_channels + 15) // 16 * 16
        first = topo.input[0][0].prefix
        # QKVAttention ("new order", unet.py:361-389) splits the qkv conv's 3C outputs as q | k | v, each
        # heads x ch wide; the attention kernel reads the legacy layout (per head: q | k | v).  Same
        # arithmetic either way, so the conv's output channels are permuted once, here, at pack time.
        qkv_perm = {}
        if getattr(topo, "new_attention_order", False):
            for e in topo.all_layers():
                if e.kind == "attn":
                    Cn, ch = e.cin, e.cin // e.heads
                    leg = torch.arange(3 * Cn)
                    h, part, i = leg // (3 * ch), (leg // ch) % 3, leg % ch
                    qkv_perm[e.prefix + ".qkv"] = (part * Cn + h * ch + i).to(device)
        params = dict(params)
        for base, perm in qkv_perm.items():
            params[base + ".weight"] = params[base + ".weight"][perm].contiguous()
            params[base + ".bias"] = params[base + ".bias"][perm].contiguous()
        self.p = params
        for name, t in params.items():
            if name.endswith(".weight") and t.dim() >= 3:
                base = name[:-len(".weight")]
                if t.dim() == 5:
                    w = t
                elif t.dim() == 4 and t.shape[2] == 3:
                    # a 2-D 3x3 conv (dims=2 models) = the 3x3x3 conv whose only non-zero depth tap is
                    # the centre one, on depth-1 volumes
                    w = torch.zeros(t.shape[0], t.shape[1], 3, 3, 3, dtype=t.dtype, device=t.device)
                    w[:, :, 1] = t
                else:                               # Conv1d / 1x1 Conv2d: pointwise
                    w = t.reshape(t.shape[0], t.shape[1], 1, 1, 1)
                if base == first and not planar and w.shape[1] != self.cin_pad:
                    wp = torch.zeros(w.shape[0], self.cin_pad, *w.shape[2:], dtype=w.dtype, device=w.device)
                    wp[:, :w.shape[1]] = w
                    w = wp
                prec = H.PRECISIONS[precision]
                self.conv[base] = PackedConv(w, params[base + ".bias"], prec, st)
                # f16x3 / f16: the big 3x3x3 layers also get the Winograd-along-depth form (1.5x
                # fewer MFMAs); conv_step picks it per call where the shape / input mode allow
                if (prec in H.WINOGRAD_OF and self.winograd and w.shape[2] == 3 and w.shape[0] % 128 == 0
                        and w.shape[1] % 16 == 0):
                    self.conv[base].wz = PackedConv(w, params[base + ".bias"], H.WINOGRAD_OF[prec], st)
        # fuse every ResBlock's emb_layers Linear into one [total, ted] matrix
        ws, bs, self.film_off = [], [], {}
        off = 0
        for entry in topo.all_layers():
            if entry.kind != "res":
                continue
            w = params[entry.prefix + ".emb_layers.1.weight"]
            b = params[entry.prefix + ".emb_layers.1.bias"]
            if not film:
                # additive embedding: h = conv1(.) + emb_out, so the Linear's bias also
                # carries conv1's bias and its output row is conv1's per-sample bias
                b = b + params[entry.prefix + ".in_layers.2.bias"]
            self.film_off[entry.prefix] = off
            ws.append(w.float())
            bs.append(b.float())
            off += w.shape[0]
        self.film_total = off
        self.ted = 4 * model_channels
        # frequency table of timestep_embedding, evaluated on the host exactly as the
        # reference does (nn.py:113-115: th.exp(...) on CPU, then .to(device))
        half = model_channels // 2
        self.freqs = torch.exp(-math.log(10000) * torch.arange(start=0, end=half, dtype=torch.float32)
                               / half).to(device)
        self.emb_w = torch.cat(ws, 0).contiguous()
        self.emb_b = torch.cat(bs, 0).contiguous()
        torch.cuda.current_stream().synchronize()

    # ------------------------------------------------------------ timestep path
    def film_rows(self, t_float, y=None):
        """[R] timesteps (float, original-process index) [+ [R] int64 class labels] -> [R, film_total]."""
        lib, st, p = self.lib, H.stream(), self.p
        R = t_float.numel()
        dev = self.device
        temb = torch.empty(R, self.mc, dtype=torch.float32, device=dev)
        e1 = torch.empty(R, self.ted, dtype=torch.float32, device=dev)
        e2 = torch.empty(R, self.ted, dtype=torch.float32, device=dev)
        rows = torch.empty(R, self.film_total, dtype=torch.float32, device=dev)
        H.check(lib.ddpm3d_timestep_embedding(H.ptr(t_float), R, self.mc, H.ptr(self.freqs), H.ptr(temb), st))
        H.check(lib.ddpm3d_linear(H.ptr(temb), R, self.mc, H.ptr(p["time_embed.0.weight"]),
                                  H.ptr(p["time_embed.0.bias"]), self.ted, 0, H.ptr(e1), self.ted, st))
        H.check(lib.ddpm3d_linear(H.ptr(e1), R, self.ted, H.ptr(p["time_embed.2.weight"]),
                                  H.ptr(p["time_embed.2.bias"]), self.ted, 1, H.ptr(e2), self.ted, st))
        if y is not None:
            # emb = time_embed(.) + label_emb(y), unet.py:703-705
            table = p["label_emb.weight"]
            H.check(lib.ddpm3d_add_embedding(H.ptr(e2), H.ptr(table), H.ptr(y), R, self.ted, table.shape[0], st))
        H.check(lib.ddpm3d_linear(H.ptr(e2), R, self.ted, H.ptr(self.emb_w), H.ptr(self.emb_b),
                                  self.film_total, 1, H.ptr(rows), self.film_total, st))
        return rows

    # ------------------------------------------------------------ plan building
    MAX_PLANS = 4   # a plan owns every activation buffer of its shape; keep the last few shapes only

    def plan(self, N, D, Hh, W):
        key = (N, D, Hh, W)
        pl = self.plans.get(key)
        if pl is None:
            while len(self.plans) >= self.MAX_PLANS:
                # drop the least recently used shape (its launches are stream-ordered before anything
                # the caching allocator hands the memory to next)
                self.plans.popitem(last=False)
            pl = _Plan(self, N, D, Hh, W)
            self.plans[key] = pl
        else:
            self.plans.move_to_end(key)
        return pl

    # ------------------------------------------------------------ the C-level plan
    def _conv_weights(self, pc):
        w = H.ConvWeights()
        if pc is not None:
            w.w_packed, w.bias = H.ptr(pc.w), H.ptr(pc.b)
            w.Cout, w.Cin, w.ksize, w.precision = pc.Cout, pc.Cin, pc.k, pc.precision
            if pc.wz is not None:
                w.w_packed_wz, w.precision_wz = H.ptr(pc.wz.w), pc.wz.precision
        return w

    def native_desc(self):
        """struct ddpm3d_unet_desc of this model (built once; borrows the engine's packed weights and parameters)"""
        if self._native_desc is not None:
            return self._native_desc[0]
        topo, p = self.topo, self.p
        layers, keep = [], []

        def one(e):
            L = H.Layer()
            L.updown, L.heads = H.UPDOWN[e.updown], e.heads
            if e.kind == "res":
                L.kind = H.LAYER_RES
                L.film_off = self.film_off[e.prefix]
                L.norm1_gamma, L.norm1_beta = H.ptr(p[e.prefix + ".in_layers.0.weight"]), H.ptr(p[e.prefix + ".in_layers.0.bias"])
                L.norm2_gamma, L.norm2_beta = H.ptr(p[e.prefix + ".out_layers.0.weight"]), H.ptr(p[e.prefix + ".out_layers.0.bias"])
                L.conv1 = self._conv_weights(self.conv[e.prefix + ".in_layers.2"])
                L.conv2 = self._conv_weights(self.conv[e.prefix + ".out_layers.3"])
                L.skip = self._conv_weights(self.conv.get(e.prefix + ".skip_connection"))
            elif e.kind == "attn":
                L.kind = H.LAYER_ATTN
                L.norm1_gamma, L.norm1_beta = H.ptr(p[e.prefix + ".norm.weight"]), H.ptr(p[e.prefix + ".norm.bias"])
                L.conv1 = self._conv_weights(self.conv[e.prefix + ".qkv"])
                L.conv2 = self._conv_weights(self.conv[e.prefix + ".proj_out"])
            elif e.kind == "downconv":
                L.kind = H.LAYER_DOWNCONV
                L.conv1 = self._conv_weights(self.conv[e.prefix + ".op"])
            elif e.kind == "upconv":
                L.kind = H.LAYER_UPCONV
                L.conv1 = self._conv_weights(self.conv[e.prefix + ".conv"])
            else:
                raise ValueError(e.kind)
            return L

        in_sizes, out_sizes = [], []
        for blk in topo.input[1:]:
            in_sizes.append(len(blk))
            layers += [one(e) for e in blk]
        layers += [one(e) for e in topo.middle]
        for blk in topo.output:
            out_sizes.append(len(blk))
            layers += [one(e) for e in blk]
        arr = (H.Layer * len(layers))(*layers)
        ins = (C.c_int32 * max(1, len(in_sizes)))(*in_sizes)
        outs = (C.c_int32 * max(1, len(out_sizes)))(*out_sizes)
        d = H.UnetDesc()
        d.n_layers, d.layers = len(layers), arr
        d.n_input_blocks, d.input_block_layers = len(in_sizes), ins
        d.n_middle_layers = len(topo.middle)
        d.n_output_blocks, d.output_block_layers = len(out_sizes), outs
        d.first = self._conv_weights(self.conv[topo.input[0][0].prefix])
        d.out_gamma, d.out_beta = H.ptr(p["out.0.weight"]), H.ptr(p["out.0.bias"])
        d.out = self._conv_weights(self.conv["out.2"])
        d.film, d.planar = int(self.film), int(self.planar)
        d.in_channels, d.cin_pad = self.in_channels, self.cin_pad
        d.arithmetic = H.PRECISIONS[self.precision]
        self._native_desc = (d, arr, ins, outs)      # ctypes arrays stay alive with the engine
        return d

    def native(self, N, D, Hh, W):
        key = (N, D, Hh, W)
        pl = self.native_plans.get(key)
        if pl is None:
            while len(self.native_plans) >= self.MAX_PLANS:
                self.native_plans.popitem(last=False)
            pl = _NativePlan(self, N, D, Hh, W)
            self.native_plans[key] = pl
        else:
            self.native_plans.move_to_end(key)
        return pl

    def forward(self, x, low_res, film_rows, film_stride, out=None):
        """x, low_res: (N,1,D,H,W) device fp32.  film_rows: device tensor whose row n
        (stride film_stride floats; 0 = one row shared by the batch) holds the
        fused emb_layers output for sample n.  Returns (N, Cout, D, H, W)."""
        N, _, D, Hh, W = x.shape
        # The kernels address a tensor with 32-bit byte offsets and the C ABI refuses (before
        # launching anything for that conv) a batch whose tensors pass 4 GiB: such a batch runs as
        # two halves, recursively (>= 32 volumes of 64^3 on the published network).
        if N > self.split_above.get((D, Hh, W), 1 << 30):
            return self._forward_halves(x, low_res, film_rows, film_stride, out)
        try:
            # (a batch that cannot be addressed is refused while its plan is being built, at the first
            # conv whose tensors pass 4 GiB -- conv_step applies the C ABI's own rule -- or, failing
            # that, by the library with DDPM3D_E2BIG before anything is enqueued for that conv)
            if self.native_plan:
                return self.native(N, D, Hh, W).run(x, low_res, film_rows, film_stride, out)
            pl = self.plan(N, D, Hh, W)
            return pl.run(x, low_res, film_rows, film_stride, out)
        except H.Ddpm3dError as e:
            if N == 1 or e.code != H.E_2BIG:
                raise
        self.plans.pop((N, D, Hh, W), None)
        self.split_above[(D, Hh, W)] = min(self.split_above.get((D, Hh, W), 1 << 30), N - 1)
        return self._forward_halves(x, low_res, film_rows, film_stride, out)

    def _forward_halves(self, x, low_res, film_rows, film_stride, out):
        N = x.shape[0]
        res = out
        h = (N + 1) // 2
        for a, b in ((0, h), (h, N)):
            rows = film_rows if film_stride == 0 else film_rows.reshape(-1)[a * film_stride:]
            part = self.forward(x[a:b].contiguous(), None if low_res is None else low_res[a:b].contiguous(),
                                rows, film_stride, None if res is None else res[a:b])
            if res is None:
                res = torch.empty((N,) + tuple(part.shape[1:]), dtype=part.dtype, device=part.device)
                res[a:b] = part
        return res


class _PlanBase:
    """What a plan offers the engine: run() = enqueue one forward (launch by launch, or as one captured graph)."""
    timing = None

    def run(self, x, low_res, film_rows, film_stride, out=None):
        H.require_device(x, "x")
        if self.eng.planar:
            H.require_device(low_res, "low_res")
        if self.eng.step_graph and self.timing is None:
            return self._replay(x, low_res, film_rows, film_stride, out)
        target = out if out is not None else self.out_buf
        self._enqueue(x.data_ptr(), low_res.data_ptr() if self.eng.planar else 0, film_rows.data_ptr(), film_stride,
                      target.data_ptr())
        return target

    # ---- step graph ----------------------------------------------
    def _replay(self, x, low_res, film_rows, film_stride, out):
        """One hipGraph launch instead of ~230 kernel launches issued from the host (what the reference does as
        ~600 ATen launches per step, gaussian_diffusion.py:522-535).  The C ABI only enqueues, so a forward is
        captured once per (plan, film-row mode) on torch's capture stream; what changes from step to step -- x,
        the conditioning volume, the step's film row -- is copied into buffers the graph was captured on (two or
        three small device-to-device copies in front of the launch), and the result is the plan's own output
        buffer.  Same kernels, same arguments, same order as the eager replay: bit-identical results."""
        eng, N = self.eng, self.N
        shared = film_stride == 0
        g = self.graphs.get(shared)
        if g is None:
            ft = eng.film_total
            st = {"x": torch.empty_like(x),
                  "lr": torch.empty_like(low_res) if eng.planar else None,
                  "film": torch.empty((1 if shared else N) * ft, dtype=torch.float32, device=x.device)}
            args = (st["x"].data_ptr(), st["lr"].data_ptr() if eng.planar else 0, st["film"].data_ptr(),
                    0 if shared else ft, self.out_buf.data_ptr())
            self._stage_inputs(st, x, low_res, film_rows, film_stride)
            self._enqueue(*args)                    # eager once: lazy per-kernel attributes, first-touch of every buffer
            torch.cuda.current_stream().synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._enqueue(*args)
            g = self.graphs[shared] = (graph, st)
        graph, st = g
        self._stage_inputs(st, x, low_res, film_rows, film_stride)
        graph.replay()
        if out is not None:
            out.copy_(self.out_buf)
            return out
        return self.out_buf

    def _stage_inputs(self, st, x, low_res, film_rows, film_stride):
        ft = self.eng.film_total
        st["x"].copy_(x)
        if st["lr"] is not None:
            # (every call: a caller's temporary may come back at the same address with other contents, so there is
            # no safe way to tell "the same conditioning volume" -- the copy is 1 MiB per 64^3 volume, ~3 us)
            st["lr"].copy_(low_res)
        flat = film_rows.reshape(-1)
        if film_stride == 0:
            st["film"].copy_(flat[:ft])
        else:
            st["film"].view(self.N, ft).copy_(torch.as_strided(flat, (self.N, ft), (film_stride, 1)))


class _NativePlan(_PlanBase):
    """The C-level plan (include/ddpm3d.h: ddpm3d_unet_plan): libddpm3d compiles the launch list itself, into one
    arena this object owns.  Same calls in the same order as _Plan below: bit-identical results."""

    def __init__(self, eng, N, D, Hh, W):
        self.eng, self.N = eng, N
        self.graphs = {}
        lib, desc = eng.lib, eng.native_desc()
        need = lib.ddpm3d_unet_plan_bytes(C.byref(desc), N, D, Hh, W)
        if need == 0:
            msg = lib.ddpm3d_unet_last_error().decode()
            # (the sizing query has no status code: a tensor beyond 32-bit offsets is what makes the engine split a batch)
            raise H.Ddpm3dError(H.E_2BIG if "4 GiB" in msg else H.E_INVAL, msg)
        self.arena = torch.empty(need, dtype=torch.uint8, device=eng.device)
        self.act_bytes = need
        handle = C.c_void_p()
        rc = lib.ddpm3d_unet_plan_create(C.byref(desc), N, D, Hh, W, H.ptr(self.arena), need, C.byref(handle))
        if rc != 0:
            raise H.Ddpm3dError(rc, lib.ddpm3d_unet_last_error().decode())
        self.handle = handle
        self.out_buf = torch.empty((N, eng.conv["out.2"].Cout, D, Hh, W), dtype=torch.float32, device=eng.device)

    def _enqueue(self, xptr, lrptr, fptr, film_stride, outptr):
        rc = self.eng.lib.ddpm3d_unet_forward(self.handle, xptr, lrptr, fptr, film_stride, outptr, H.stream())
        if rc != 0:
            raise H.Ddpm3dError(rc, self.eng.lib.ddpm3d_unet_last_error().decode())

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.eng.lib.ddpm3d_unet_plan_destroy(self.handle)
        except Exception:
            pass


class _Plan(_PlanBase):
    def __init__(self, eng, N, D, Hh, W):
        self.eng = eng
        self.N = N
        self.steps = []       # (fn, args) replayed in order
        self.keep = []        # keeps ctypes structs / tensors alive
        self.film_patches = []  # gn_finalize arg lists whose film pointer is per call
        self.bias_patches = []  # (conv desc, film offset): additive-embedding conv1 bias rows
        self.conv_meta = {}     # step index -> (kernel tag, algorithmic FLOPs)
        self.timing = None      # set to a list to collect (tag, flops, start_evt, end_evt) per conv
        self.graphs = {}        # film-row mode -> (captured forward, its static input buffers): _replay()
        dev = eng.device
        topo = eng.topo
        lib = eng.lib

        # Activation buffers come from a pool keyed by size and go back to it when their last
        # reader has been ENQUEUED (every step runs in order on one stream, so a later step may
        # overwrite them): a plan holds the skip stack plus a handful of working tensors instead of
        # one private buffer per layer (published net @ 64^3: 1.3 GB instead of 3 GB).
        self.pool = {}
        self.act_bytes = 0

        # bf16 / f16 modes: the residual stream (every tensor a conv writes and convs read) is stored in
        # 16 bits -- bf16, or IEEE f16 as the reference's --use_fp16 torso does (unet.py:1035,
        # fp16_util.py:15-22); tensors read by the fp32-only kernels (attention, subsample, gn_stats) stay fp32
        self.half_dtype = {"bf16": torch.bfloat16, "f16": torch.float16}.get(eng.precision)
        # only the split-f16 arithmetic scales its operands from in_bound (api.hip: prec_scaled); the exact
        # and the bf16 plans enqueue neither the input-range pass nor bound-only finalizes
        self.scaled = eng.precision in ("f16x3", "f16")

        def new_act(Cn, d, h, w, fp32=False):
            # the statistics buffer is attached by the conv step that produces the tensor
            # (its row count depends on how that conv is tiled / split)
            numel = N * d * h * w * Cn
            dt = self.half_dtype if (self.half_dtype is not None and not fp32) else torch.float32
            free = self.pool.get((numel, dt))
            if free:
                buf = free.pop()
            else:
                buf = torch.empty(numel, dtype=dt, device=dev)
                self.act_bytes += buf.element_size() * numel
                self.keep.append(buf)  # descriptors hold raw pointers: the plan owns every buffer
            return Act(buf, Cn, d, h, w, None, 0)

        def release(act):
            self.pool.setdefault((act.buf.numel(), act.buf.dtype), []).append(act.buf)

        self.ws_descs = []      # conv descriptors that need the shared split-K workspace
        self.ws_bytes = 0

        self.new_act = new_act
        self.release = release
        first = topo.input[0][0]
        cin_conv = eng.conv[first.prefix]
        self.first_desc = None
        h = new_act(cin_conv.Cout, D, Hh, W)
        if eng.planar:
            # range of the two input volumes (they carry no statistics): [N][2] = max |x|, max |low_res|
            self.in_absmax = torch.empty(N * 2, dtype=torch.float32, device=dev)
            self.absmax_args = [0, 0, N, D * Hh * W, H.ptr(self.in_absmax), 0]
            if self.scaled:
                self.steps.append((lib.ddpm3d_absmax, self.absmax_args))
            self.first_desc = self.conv_step(cin_conv, srcs=None, out=h, planar=True,
                                             bound=(self.in_absmax, 0, 2, 1))
        else:
            # ordinary multi-channel input: (N, C, voxels) -> NDHWC padded to 16 channels, its range
            ci = eng.in_channels
            xin = new_act(eng.cin_pad, D, Hh, W, fp32=True)
            self.in_absmax = torch.empty(N, dtype=torch.float32, device=dev)
            self.absmax_args = [0, 0, N, ci * D * Hh * W, H.ptr(self.in_absmax), 0]
            self.pad_args = [0, N, ci, D * Hh * W, eng.cin_pad, H.ptr(xin.buf), 0]
            if self.scaled:
                self.steps.append((lib.ddpm3d_absmax, self.absmax_args))
            self.steps.append((lib.ddpm3d_ncdhw_to_ndhwc_pad, self.pad_args))
            self.first_desc = self.conv_step(cin_conv, srcs=[xin], out=h, bound=(self.in_absmax, 0, 1, 1))
        hs = [h]
        for blk in topo.input[1:]:
            for i, e in enumerate(blk):
                prev, h = h, self.layer(e, [h])
                if i > 0:
                    release(prev)          # a block's intermediate; its input stays on the skip stack
            hs.append(h)
        for e in topo.middle:
            prev, h = h, self.layer(e, [h])
            if prev is not hs[-1]:
                release(prev)
        for blk in topo.output:
            skip = hs.pop()
            srcs = [h, skip]
            for e in blk:
                h = self.layer(e, srcs)
                for a in srcs:
                    release(a)
                srcs = [h]
        # out: GN -> SiLU -> conv, stored NCDHW
        A, B, bnd = self.finalize([h], "out.0", None)
        oc = eng.conv["out.2"]
        self.out_C = oc.Cout
        self.out_shape = (N, oc.Cout, D, Hh, W)
        self.out_buf = torch.empty(self.out_shape, dtype=torch.float32, device=dev)
        self.last_desc = self.conv_step(oc, srcs=[h], out=None, aff=(A, B), act=H.ACT_SILU, bound=(bnd, 0, 32, 2),
                                        out_tensor=self.out_buf, out_layout=H.OUT_NCDHW)
        release(h)
        # one split-K scratch buffer shared by every conv of the plan (they run in stream order)
        if self.ws_bytes:
            self.workspace = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)
            for dsc in self.ws_descs:
                dsc.workspace, dsc.workspace_bytes = H.ptr(self.workspace), self.ws_bytes

    # ---- helpers -------------------------------------------------------------
    def finalize(self, srcs, gn_prefix, film_prefix, need_bound=None):
        """gn_finalize over the virtual concat of `srcs`; returns (A, B, bound) tensors.  bound =
        [N][32][2]: upper bounds of |act(A*x + B)| (entry 0) and of |x| (entry 1) per group, the
        in_bound of the convs that read the tensor normalised / raw.  gn_prefix None: bounds only
        (no launch at all when nothing reads them: exact / bf16 plans, unless need_bound)."""
        eng, N = self.eng, self.N
        if gn_prefix is None and not (self.scaled if need_bound is None else need_bound):
            return None, None, None
        Cn = sum(s.C for s in srcs)
        A = torch.empty(N * Cn, dtype=torch.float32, device=eng.device) if gn_prefix else None
        B = torch.empty(N * Cn, dtype=torch.float32, device=eng.device) if gn_prefix else None
        bound = torch.empty(N * 32 * 2, dtype=torch.float32, device=eng.device)
        s0 = srcs[0]
        s1 = srcs[1] if len(srcs) > 1 else None
        if s1 is not None and s1.voxels != s0.voxels:
            raise RuntimeError("concat of tensors with different spatial size")
        gamma = eng.p[gn_prefix + ".weight"] if gn_prefix else None
        beta = eng.p[gn_prefix + ".bias"] if gn_prefix else None
        args = [H.ptr(s0.stats), s0.C, s0.rows,
                H.ptr(s1.stats) if s1 else 0, s1.C if s1 else 0, s1.rows if s1 else 0,
                N, 32, float(s0.voxels), 1e-5, H.ptr(gamma), H.ptr(beta),
                0, 0, 0, H.ptr(A), H.ptr(B), H.ptr(bound), 0]
        if film_prefix is not None:
            args[14] = eng.film_off[film_prefix]
            self.film_patches.append(args)
        self.steps.append((eng.lib.ddpm3d_gn_finalize, args))
        self.keep += [A, B, bound]
        return A, B, bound

    def conv_step(self, pc, srcs, out, aff=None, act=H.ACT_NONE, in_mode=H.IN_SAME, res=None,
                  res_mode=H.RES_NONE, planar=False, out_tensor=None, out_layout=H.OUT_NDHWC,
                  bias_per_n=False, want_stats=True, bound=None):
        """bound = (tensor, first entry, entries per sample, stride): ddpm3d_conv_desc.in_bound"""
        N = self.N
        d = H.ConvDesc()
        lib = self.eng.lib
        cin_total = 2 if planar else sum(s.C for s in srcs)
        if pc.wz is not None and not planar and in_mode in (H.IN_SAME, H.IN_UP):
            pc_use = pc.wz       # Winograd-D form: same layer, same arithmetic, 2/3 of the MFMAs
        else:
            pc_use = pc
        if out is not None:
            d.N, d.D, d.H, d.W = N, out.D, out.H, out.W
            d.out = H.ptr(out.buf)
        else:
            s = srcs[0]
            d.N, d.D, d.H, d.W = N, s.D, s.H, s.W
            d.out = H.ptr(out_tensor)
            d.stats = 0
            d.stats_rows = 0
        d.Cout, d.ksize = pc.Cout, pc.k
        d.out_layout = out_layout
        if planar:
            d.in_mode, d.Cin, d.C0, d.C1 = H.IN_PLANAR2, 2, 1, 1
        else:
            d.in_mode = in_mode
            d.src0, d.C0 = H.ptr(srcs[0].buf), srcs[0].C
            if len(srcs) > 1:
                d.src1, d.C1 = H.ptr(srcs[1].buf), srcs[1].C
            d.Cin = d.C0 + d.C1
            if d.Cin != pc.Cin:
                raise RuntimeError("conv %dx%d fed %d channels" % (pc.Cout, pc.Cin, d.Cin))
        # the C ABI's addressing rule (api.hip: 32-bit byte offsets into each source tensor and into one
        # output sample), applied here so an oversized batch is refused BEFORE its buffers are allocated
        for s in ([] if planar else srcs):
            if N * s.voxels * s.C * s.buf.element_size() >= 0xFFFFFFF0:
                raise H.Ddpm3dError(H.E_2BIG, "a source tensor of %d x %d voxels x %d channels exceeds 4 GiB; "
                                              "split the batch" % (N, s.voxels, s.C))
        if aff is not None:
            d.aff_a, d.aff_b = H.ptr(aff[0]), H.ptr(aff[1])
        d.act = act
        io = 0
        if not planar:
            io |= H.IO_SRC0_BF16 if srcs[0].half else 0
            io |= H.IO_SRC1_BF16 if (len(srcs) > 1 and srcs[1].half) else 0
        io |= H.IO_OUT_BF16 if (out is not None and out.half) else 0
        io |= H.IO_RES_BF16 if (res is not None and res.half) else 0
        if io and self.half_dtype == torch.float16:
            io |= H.IO_HALF_IS_F16
        d.io_dtype = io
        if self.scaled:
            if bound is None or bound[0] is None:
                raise RuntimeError("conv_step without an input bound")
            d.in_bound = bound[0].data_ptr() + 4 * bound[1]
            d.in_bound_count, d.in_bound_stride = bound[2], bound[3]
        d.precision = pc_use.precision
        d.w_packed, d.bias = H.ptr(pc_use.w), H.ptr(pc.b)
        d.bias_stride_n = 0  # per-sample bias rows are patched in run()
        d.res_mode = res_mode
        if res is not None and res.C != pc.Cout:
            # the epilogue reads the residual at stride Cout (ddpm3d.h): a narrower tensor would be
            # read out of bounds
            raise RuntimeError("residual has %d channels, the conv writes %d" % (res.C, pc.Cout))
        d.res = H.ptr(res.buf) if res is not None else 0
        # how the library will run this descriptor: statistics rows, split-K scratch
        rows, need, _ = H.conv_plan(d)
        if out is not None and want_stats:
            out.rows = rows
            out.stats = torch.empty(N * rows * pc.Cout * 2, dtype=torch.float64,
                                    device=self.eng.device)     # fp64 (sum, sum of squares) rows
            self.keep.append(out.stats)    # the descriptor holds a raw pointer
            d.stats, d.stats_rows = H.ptr(out.stats), rows
        if need:
            self.ws_descs.append(d)
            self.ws_bytes = max(self.ws_bytes, need)
        self.keep.append(d)
        # bookkeeping for measurement: algorithmic FLOPs (2 per MAC) and the kernel family the C side
        # routes this descriptor to (its own answer: no copy of the routing rules here)
        flops = 2.0 * N * d.D * d.H * d.W * pc.Cout * d.Cin * pc.k ** 3
        name = C.create_string_buffer(64)
        H.check(lib.ddpm3d_conv_kernel_family(C.byref(d), name, 64))
        tag = name.value.decode()
        self.conv_meta[len(self.steps)] = (tag, flops)
        self.steps.append((self.eng.lib.ddpm3d_conv3d, [C.byref(d), 0]))
        return d

    def layer(self, e, srcs):
        eng = self.eng
        if e.kind == "res":
            return self.resblock(e, srcs)
        if e.kind == "upconv":
            x = srcs[0]
            pc = eng.conv[e.prefix + ".conv"]
            y = self.new_act(pc.Cout, x.D, x.H * 2, x.W * 2)
            _, _, bnd = self.finalize([x], None, None)
            self.conv_step(pc, [x], y, in_mode=H.IN_UP, bound=(bnd, 1, 32, 2))
            return y
        if e.kind == "attn":
            return self.attention(e, srcs[0])
        if e.kind == "downconv":
            # Downsample(use_conv=True), unet.py:129-133: Conv3d(stride=(1,2,2), padding=1), the
            # factory's default resampling (resblock_updown=False).  One strided conv launch (the halo
            # tile is laid out in the source grid); its epilogue emits the statistics of the result.
            x = srcs[0]
            if (x.H | x.W) & 1:
                raise RuntimeError("Downsample needs even H, W (got %dx%d)" % (x.H, x.W))
            pc = eng.conv[e.prefix + ".op"]
            y = self.new_act(pc.Cout, x.D, x.H // 2, x.W // 2)
            _, _, bnd = self.finalize([x], None, None)
            self.conv_step(pc, [x], y, in_mode=H.IN_STRIDE2, bound=(bnd, 1, 32, 2))
            return y
        raise ValueError(e.kind)

    def attention(self, e, x):
        """AttentionBlock (unet.py:296-305): x + proj_out(attn(qkv(norm(x)))) as
        gn_finalize -> 1x1 conv (GroupNorm prologue, no activation) -> streaming attention
        -> 1x1 conv (+ residual x, + statistics for the next GroupNorm)."""
        eng, N = self.eng, self.N
        p = e.prefix
        heads = e.heads
        Cn = x.C
        if Cn % heads:
            raise RuntimeError("attention: %d channels not divisible by %d heads" % (Cn, heads))
        ch = Cn // heads
        if ch not in (32, 64, 128):
            raise NotImplementedError("attention with %d channels per head (32, 64 or 128 are built)" % ch)
        A, B, bnd = self.finalize([x], p + ".norm", None)
        qkv = self.new_act(3 * Cn, x.D, x.H, x.W, fp32=True)           # read by the attention kernel
        self.conv_step(eng.conv[p + ".qkv"], [x], qkv, aff=(A, B), act=H.ACT_NONE, bound=(bnd, 0, 32, 2))
        # range of q, k, v (and of the attention output, a convex combination of v) from qkv's
        # own partial sums
        # (the bf16 plan keeps this one: its attention products run in the f16x3 arithmetic)
        _, _, qb = self.finalize([qkv], None, None, need_bound=eng.precision != "f32")
        a = self.new_act(Cn, x.D, x.H, x.W, fp32=True)
        # two T x T x ch products per head (the reference's count_flops_attn, unet.py:308-325)
        self.conv_meta[len(self.steps)] = ("attention_ch%d" % ch, 4.0 * N * heads * float(x.voxels) ** 2 * ch)
        # the two products in the model's arithmetic: exact fp32 MFMA in the "f32" mode, fp32-grade
        # f16x3 otherwise (in the "f16" mode too: the reference's fp16 torso keeps the softmax in
        # fp32 (unet.py:351), and f16-rounded scores would cost more accuracy than the convs do)
        aprec = H.PREC_F32 if eng.precision == "f32" else H.PREC_F16X3   # (bf16 mode too: fp32-grade scores)
        self.steps.append((eng.lib.ddpm3d_attention_p,
                           [H.ptr(qkv.buf), N, x.voxels, heads, ch, aprec, (H.ptr(qb) + 4) if qb is not None else 0, 32, 2, H.ptr(a.buf), 0]))
        self.release(qkv)
        y = self.new_act(Cn, x.D, x.H, x.W)
        self.conv_step(eng.conv[p + ".proj_out"], [a], y, res=x, res_mode=H.RES_SAME, bound=(qb, 1, 32, 2))
        self.release(a)
        return y

    def resblock(self, e, srcs):
        eng = self.eng
        p = e.prefix
        x0 = srcs[0]
        if e.updown == "down":
            if (x0.H | x0.W) & 1:
                raise RuntimeError("Downsample needs even H, W (got %dx%d)" % (x0.H, x0.W))
            d, h, w, im, rm = x0.D, x0.H // 2, x0.W // 2, H.IN_POOL, H.RES_POOL
        elif e.updown == "up":
            d, h, w, im, rm = x0.D, x0.H * 2, x0.W * 2, H.IN_UP, H.RES_UP
        else:
            d, h, w, im, rm = x0.D, x0.H, x0.W, H.IN_SAME, H.RES_SAME
        c1 = eng.conv[p + ".in_layers.2"]
        c2 = eng.conv[p + ".out_layers.3"]
        A1, B1, bnd1 = self.finalize(srcs, p + ".in_layers.0", None)
        h1 = self.new_act(c1.Cout, d, h, w)
        aff1, act1, srcs1 = (A1, B1), H.ACT_SILU, srcs
        pooled = None
        if e.updown == "down" and c1.wz is not None and len(srcs) == 1:
            # h_upd(in_rest(x)) as a pass of its own (ddpm3d_pool_act, fp32 result): conv1 then reads a plain
            # tensor and runs its Winograd-D form instead of the direct kernel with the pool in its staging
            # (128 -> 128 @ 64x32x32 from a 64^3 input: 0.244 -> 0.04 + 0.11 ms).  Same values: the pass
            # evaluates what that staging evaluates; their bound (entry 0 of the finalize) bounds the means.
            pooled = self.new_act(x0.C, d, h, w, fp32=True)
            io = (H.IO_SRC0_BF16 if x0.half else 0) | (H.IO_HALF_IS_F16 if x0.buf.dtype == torch.float16 else 0)
            self.conv_meta[len(self.steps)] = ("pool_act", 0.0)      # timed with the convs (no FLOPs of its own)
            self.steps.append((eng.lib.ddpm3d_pool_act,
                               [H.ptr(x0.buf), H.ptr(A1), H.ptr(B1), H.ACT_SILU, 1, self.N, d, h, w, x0.C,
                                H.ptr(pooled.buf), io, 0]))
            aff1, act1, srcs1, im = None, H.ACT_NONE, [pooled], H.IN_SAME
        if eng.film:
            self.conv_step(c1, srcs1, h1, aff=aff1, act=act1, in_mode=im, bound=(bnd1, 0, 32, 2))
            A2, B2, bnd2 = self.finalize([h1], p + ".out_layers.0", p)
        else:
            # additive embedding: conv1's bias is the per-sample film row slice
            dsc = self.conv_step(c1, srcs1, h1, aff=aff1, act=act1, in_mode=im, bias_per_n=True,
                                 bound=(bnd1, 0, 32, 2))
            self.bias_patches.append((dsc, eng.film_off[p]))
            A2, B2, bnd2 = self.finalize([h1], p + ".out_layers.0", None)
        if pooled is not None:
            self.release(pooled)
        y = self.new_act(c2.Cout, d, h, w)
        skip = eng.conv.get(p + ".skip_connection")
        if skip is not None:
            if e.updown is not None:
                raise RuntimeError("up/down ResBlock with a channel change is not in the reference")
            # y = skip(x) on the RAW block input (its range: entry 1 of the same finalize), then
            # accumulated into
            self.conv_step(skip, srcs, y, want_stats=False, bound=(bnd1, 1, 32, 2))
            self.conv_step(c2, [h1], y, aff=(A2, B2), act=H.ACT_SILU, res=y, res_mode=H.RES_SAME,
                           bound=(bnd2, 0, 32, 2))
        else:
            if len(srcs) > 1:
                # Identity skip over the decoder's virtual concat [h, skip] (2*inch == outch, e.g. a
                # decreasing channel_mult on a directly built model): the residual would be the
                # concatenation itself, which no epilogue mode reads.
                raise NotImplementedError("ResBlock with an Identity skip over a concatenated input "
                                          "(%d + %d -> %d channels)" % (srcs[0].C, srcs[1].C, c2.Cout))
            self.conv_step(c2, [h1], y, aff=(A2, B2), act=H.ACT_SILU, res=x0, res_mode=rm, bound=(bnd2, 0, 32, 2))
        self.release(h1)
        return y

    # ---- execution -----------------------------------------------------------
    def _enqueue(self, xptr, lrptr, fptr, film_stride, outptr):
        """Patch the per-call pointers into the descriptors and enqueue every step on the current stream."""
        st = H.stream()
        if self.eng.planar:
            self.first_desc.src0 = xptr
            self.first_desc.src1 = lrptr
            self.absmax_args[0], self.absmax_args[1] = xptr, lrptr
        else:
            self.absmax_args[0] = self.pad_args[0] = xptr
        for a in self.film_patches:
            a[12], a[13] = fptr, film_stride
        for dsc, off in self.bias_patches:
            dsc.bias = fptr + 4 * off
            dsc.bias_stride_n = film_stride
        self.last_desc.out = outptr
        if self.timing is None:
            for fn, args in self.steps:
                args[-1] = st
                rc = fn(*args)
                if rc != 0:
                    H.check(rc)
        else:
            # measurement replay: HIP events (on the stream the kernels are enqueued on)
            # around every conv launch
            for i, (fn, args) in enumerate(self.steps):
                args[-1] = st
                meta = self.conv_meta.get(i)
                if meta is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                rc = fn(*args)
                if rc != 0:
                    H.check(rc)
                if meta is not None:
                    e1.record()
                    self.timing.append((meta[0], meta[1], e0, e1))
