"""UNetModel: MI355X-native drop-in for lvdm.modules.networks.openaimodel3d.UNetModel.

Same constructor keywords (openaimodel3d.py:314-345), same `forward(x, timesteps, context,
features_adapter, fs, **kwargs)` (:552) and an identical state_dict key set / shapes (SURVEY §2.4,
including the reference's `temopral_conv` spelling), so checkpoints and the YAML `target:` seam
(inference_512_v1.0.yaml:24-25) carry over.  What differs is everything underneath:

* activations never exist as NCHW: they are channels-last token matrices [frames*H*W, C], which makes
  every Linear a plain GEMM, every 3x3 / temporal conv an implicit GEMM, and removes all the
  `(b f) c h w <-> b c f h w` shuffles around the temporal modules (openaimodel3d.py:36-48);
* the residual stream (the tensor every ResBlock / transformer block adds into) is kept in f32 in
  HBM; branch operands (normalised activations, q/k/v, GEGLU products) are 16-bit.  16-bit rounding
  then happens once per branch operand instead of once per residual add (~180 adds per forward), which
  is what brings a whole forward to ~1e-3 of the f32 reference; the price is 2 extra bytes per element
  on the norm reads of the stream (HBM-bound kernels, ~1 % of a step);
* every op goes through an op table (`self.ops`): `HipOps` (gfx950 kernels over the C-ABI) in
  production.  nn.Module parameters are only the reference-format storage; `prepare()` builds the
  kernel-side packed copies (fused qkv, tap-major conv weights, interleaved GEGLU rows).
* `fp` (a FrameParallel object, optional) supplies the three cross-frame exchanges when the frame
  axis is sharded over ranks: (T,H,W) GroupNorm statistics, temporal-conv halos, temporal K/V.
"""
import math

import torch
import torch.nn as nn

from . import packing


def _unsupported(flag, name):
    if flag:
        raise NotImplementedError(f"{name} is not used by the shipped Open-Pandora configs and is not built")


_FREQS = {}


def timestep_embedding(timesteps, dim, max_period=10000):
    """Sinusoidal embedding with the reference's bf16-quantised frequency table
    (utils_diffusion.py:8-28: `torch.arange(..., dtype=torch.bfloat16)` feeds the exp)."""
    key = (dim, max_period, str(timesteps.device))
    if key not in _FREQS:  # built on the host once per device (no H2D copy inside a graph capture)
        half = dim // 2
        freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.bfloat16) / half)
        _FREQS[key] = freqs.float().to(timesteps.device)
    args = timesteps[:, None].float() * _FREQS[key][None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _timestep_row(ops, t, dim):
    """Row 0 of the embedding through the op table's one-launch kernel (pm_timestep_embedding) where it has one."""
    fn = getattr(ops, "timestep_embedding", None)
    t = t.to(ops.device)
    if fn is None:
        return timestep_embedding(t, dim)[0].contiguous()
    key = (dim, 10000, str(t.device))
    if key not in _FREQS:
        timestep_embedding(t, dim)
    if t.dtype not in (torch.int64, torch.float32):
        t = t.float()
    return fn(t[:1].contiguous(), _FREQS[key])[0]


# ------------------------------------------------------------------------------------------------
# parameter containers: same attribute names / child indices as the reference modules
# ------------------------------------------------------------------------------------------------
class CrossAttention(nn.Module):
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, image_cross_attention=False,
                 image_cross_attention_scale_learnable=False):
        super().__init__()
        assert dim_head == 64, "the attention kernels are specialised for head dim 64"
        inner = heads * dim_head
        self.heads = heads
        self.self_attn = context_dim is None
        context_dim = context_dim or query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(0.0))
        self.image_cross_attention = image_cross_attention
        if image_cross_attention:
            self.to_k_ip = nn.Linear(context_dim, inner, bias=False)
            self.to_v_ip = nn.Linear(context_dim, inner, bias=False)
            # attention.py:77-78: the 256x256 checkpoint's per-block scalar; out + scale * out_ip * (tanh(alpha) + 1) (:139-140)
            if image_cross_attention_scale_learnable:
                self.register_parameter("alpha", nn.Parameter(torch.tensor(0.)))

    def image_scale(self):
        """Weight of the image branch (attention.py:138-142): image_cross_attention_scale (1.0 in every shipped yaml), times
        tanh(alpha) + 1 when the scale is learnable."""
        alpha = getattr(self, "alpha", None)
        return 1.0 if alpha is None else 1.0 * (float(torch.tanh(alpha.detach().float())) + 1.0)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.Sequential(GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim))


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, n_heads, d_head, context_dim=None, image_cross_attention=False,
                 image_cross_attention_scale_learnable=False):
        super().__init__()
        self.attn1 = CrossAttention(dim, None, n_heads, d_head)
        self.ff = FeedForward(dim)
        self.attn2 = CrossAttention(dim, context_dim, n_heads, d_head, image_cross_attention,
                                    image_cross_attention_scale_learnable)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)


class _Transformer(nn.Module):
    def __init__(self, in_channels, n_heads, d_head, context_dim, use_linear, image_cross_attention, conv1d,
                 image_cross_attention_scale_learnable=False):
        super().__init__()
        inner = n_heads * d_head
        self.in_channels, self.inner, self.heads = in_channels, inner, n_heads
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6, affine=True)
        if use_linear:
            self.proj_in = nn.Linear(in_channels, inner)
        elif conv1d:
            self.proj_in = nn.Conv1d(in_channels, inner, kernel_size=1)
        else:
            self.proj_in = nn.Conv2d(in_channels, inner, kernel_size=1)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, n_heads, d_head, context_dim, image_cross_attention,
                                   image_cross_attention_scale_learnable)])
        if use_linear:
            self.proj_out = nn.Linear(inner, in_channels)
        elif conv1d:
            self.proj_out = nn.Conv1d(inner, in_channels, kernel_size=1)
        else:
            self.proj_out = nn.Conv2d(inner, in_channels, kernel_size=1)
        for p in self.proj_out.parameters():  # zero_module (attention.py:288-290,360-362)
            nn.init.zeros_(p)


class SpatialTransformer(_Transformer):
    def __init__(self, in_channels, n_heads, d_head, context_dim, use_linear, image_cross_attention,
                 image_cross_attention_scale_learnable=False):
        super().__init__(in_channels, n_heads, d_head, context_dim, use_linear, image_cross_attention, False,
                         image_cross_attention_scale_learnable)


class TemporalTransformer(_Transformer):
    def __init__(self, in_channels, n_heads, d_head, use_linear):
        # only_self_att=True => context_dim None => attn2 is self-attention too (attention.py:347-348)
        super().__init__(in_channels, n_heads, d_head, None, use_linear, False, True)


class TemporalConvBlock(nn.Module):
    def __init__(self, channels):
        super().__init__()
        k, pad = (3, 1, 1), (1, 0, 0)
        self.conv1 = nn.Sequential(nn.GroupNorm(32, channels), nn.SiLU(), nn.Conv3d(channels, channels, k, padding=pad))
        for name in ("conv2", "conv3", "conv4"):
            setattr(self, name, nn.Sequential(nn.GroupNorm(32, channels), nn.SiLU(), nn.Dropout(0.1),
                                              nn.Conv3d(channels, channels, k, padding=pad)))
        nn.init.zeros_(self.conv4[-1].weight)
        nn.init.zeros_(self.conv4[-1].bias)


class ResBlock(nn.Module):
    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_temporal_conv=False):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.in_layers = nn.Sequential(nn.GroupNorm(32, channels), nn.SiLU(),
                                       nn.Conv2d(channels, self.out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, self.out_channels))
        self.out_layers = nn.Sequential(nn.GroupNorm(32, self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        nn.Conv2d(self.out_channels, self.out_channels, 3, padding=1))
        for p in self.out_layers[-1].parameters():
            nn.init.zeros_(p)
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        else:
            self.skip_connection = nn.Conv2d(channels, self.out_channels, 1)
        self.use_temporal_conv = use_temporal_conv
        if use_temporal_conv:
            self.temopral_conv = TemporalConvBlock(self.out_channels)  # (sic) reference spelling


class Downsample(nn.Module):
    def __init__(self, channels, out_channels=None):
        super().__init__()
        self.op = nn.Conv2d(channels, out_channels or channels, 3, stride=2, padding=1)


class Upsample(nn.Module):
    def __init__(self, channels, out_channels=None):
        super().__init__()
        self.conv = nn.Conv2d(channels, out_channels or channels, 3, padding=1)


class TimestepEmbedSequential(nn.Sequential):
    pass


# ------------------------------------------------------------------------------------------------
class _Ctx:
    """Per-forward execution state."""
    __slots__ = ("ops", "fp", "F", "H", "W", "emb_bias", "ctx_text", "ctx_img", "w", "stats", "kv_text", "kv_img",
                 "img_shared", "B", "T", "level")  # B clips of T frames each batched along the rows: F = B * T (B = 1: the reference's call)


class UNetModel(packing.PackedWeights, nn.Module):
    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0.0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, context_dim=None,
                 use_scale_shift_norm=False, resblock_updown=False, num_heads=-1, num_head_channels=-1,
                 transformer_depth=1, use_linear=False, use_checkpoint=False, temporal_conv=False,
                 tempspatial_aware=False, temporal_attention=True, use_relative_position=True,
                 use_causal_attention=False, temporal_length=None, use_fp16=False, addition_attention=False,
                 temporal_selfatt_only=True, image_cross_attention=False,
                 image_cross_attention_scale_learnable=False, default_fs=4, fs_condition=False):
        super().__init__()
        _unsupported(dims != 2, "dims != 2")
        _unsupported(use_scale_shift_norm, "use_scale_shift_norm")
        _unsupported(resblock_updown, "resblock_updown")
        _unsupported(not conv_resample, "conv_resample=False")
        _unsupported(transformer_depth != 1, "transformer_depth != 1")
        _unsupported(tempspatial_aware, "tempspatial_aware")
        _unsupported(use_relative_position, "use_relative_position")
        _unsupported(use_causal_attention, "use_causal_attention")
        _unsupported(not temporal_selfatt_only, "temporal cross-attention")
        _unsupported(num_head_channels == -1, "num_heads without num_head_channels")
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.num_res_blocks, self.attention_resolutions = num_res_blocks, attention_resolutions
        self.dropout, self.channel_mult, self.conv_resample = dropout, channel_mult, conv_resample
        self.temporal_attention = temporal_attention
        self.use_checkpoint = use_checkpoint
        self.dtype = torch.bfloat16  # as the reference hard-codes (openaimodel3d.py:364)
        self.addition_attention = addition_attention
        self.temporal_length = temporal_length
        self.image_cross_attention = image_cross_attention
        self.default_fs, self.fs_condition = default_fs, fs_condition
        ted = model_channels * 4

        self.time_embed = nn.Sequential(nn.Linear(model_channels, ted), nn.SiLU(), nn.Linear(ted, ted))
        if fs_condition:
            self.fps_embedding = nn.Sequential(nn.Linear(model_channels, ted), nn.SiLU(), nn.Linear(ted, ted))
            nn.init.zeros_(self.fps_embedding[-1].weight)
            nn.init.zeros_(self.fps_embedding[-1].bias)
        self.input_blocks = nn.ModuleList(
            [TimestepEmbedSequential(nn.Conv2d(in_channels, model_channels, 3, padding=1))])
        if addition_attention:
            self.init_attn = TimestepEmbedSequential(
                TemporalTransformer(model_channels, 8, num_head_channels, use_linear=False))

        def attn_layers(ch):
            heads = ch // num_head_channels
            layers = [SpatialTransformer(ch, heads, num_head_channels, context_dim, use_linear, image_cross_attention,
                                         image_cross_attention_scale_learnable)]
            if temporal_attention:
                layers.append(TemporalTransformer(ch, heads, num_head_channels, use_linear))
            return layers

        input_block_chans = [model_channels]
        ch, ds = model_channels, 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [ResBlock(ch, ted, dropout, mult * model_channels, temporal_conv)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    layers += attn_layers(ch)
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                input_block_chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, ch)))
                input_block_chans.append(ch)
                ds *= 2
        mid = [ResBlock(ch, ted, dropout, None, temporal_conv)] + attn_layers(ch)[:2 if temporal_attention else 1]
        mid.append(ResBlock(ch, ted, dropout, None, temporal_conv))
        self.middle_block = TimestepEmbedSequential(*mid)
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = input_block_chans.pop()
                layers = [ResBlock(ch + ich, ted, dropout, mult * model_channels, temporal_conv)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    layers += attn_layers(ch)
                if level and i == num_res_blocks:
                    layers.append(Upsample(ch, ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(nn.GroupNorm(32, ch), nn.SiLU(), nn.Conv2d(model_channels, out_channels, 3, padding=1))
        for p in self.out[-1].parameters():
            nn.init.zeros_(p)

        self.ops = None
        self.fp = None
        self._init_packed()

    # ---- kernel-side weights (invalidated by packing.PackedWeights on every kind of parameter change) ----
    def bind(self, ops, fp=None):
        """Select the op table (HipOps in production) and optional frame-parallel context."""
        self.ops, self.fp = ops, fp
        if fp is not None and getattr(ops, "stats_i64", False):
            # frame shards exchange f32 partial sums between ranks (frame_parallel / csrc/peer.hip): the op table goes back to
            # the column-sum + finalize form, whose totals ARE f32 (converting the integer limbs would cost more launches)
            ops.stats_i64 = ops.stats_nsum = False
        self.invalidate_packed()
        return self

    def prepare(self):
        """Build the packed, device-resident weight set for the bound op table."""
        ops = self.ops
        dev, dt = ops.device, ops.dtype
        W = {}
        wt = lambda t: t.detach().to(device=dev, dtype=dt).contiguous()
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()

        inner_of = lambda attn: attn.to_q.weight.shape[0]

        def lin(mod):
            w = mod.weight
            if w.dim() > 2:  # 1x1 Conv1d / Conv2d
                w = w.reshape(w.shape[0], w.shape[1])
            return wt(w), (None if mod.bias is None else f32(mod.bias))

        emb_w, emb_b, off = [], [], 0
        for name, mod in self.named_modules():
            if isinstance(mod, ResBlock):
                e = {}
                e["gn1"] = (f32(mod.in_layers[0].weight), f32(mod.in_layers[0].bias))
                e["conv1"] = wt(packing.pack_conv3x3(mod.in_layers[2].weight))
                e["gn2"] = (f32(mod.out_layers[0].weight), f32(mod.out_layers[0].bias))
                e["conv2"] = (wt(packing.pack_conv3x3(mod.out_layers[3].weight)), f32(mod.out_layers[3].bias))
                e["skip"] = None if isinstance(mod.skip_connection, nn.Identity) else lin(mod.skip_connection)
                # emb_layers of every ResBlock run as ONE batched GEMV per forward; the conv1 bias is
                # folded into the same vector: h = conv1(.) + (b_conv1 + W_emb silu(emb) + b_emb)
                emb_w.append(mod.emb_layers[1].weight)
                emb_b.append(mod.emb_layers[1].bias.float() + mod.in_layers[2].bias.float())
                e["emb_slice"] = (off, off + mod.out_channels)
                off += mod.out_channels
                if mod.use_temporal_conv:
                    tc = mod.temopral_conv
                    e["tconv"] = [((f32(s[0].weight), f32(s[0].bias)), wt(packing.pack_conv_t3(s[-1].weight)),
                                   f32(s[-1].bias)) for s in (tc.conv1, tc.conv2, tc.conv3, tc.conv4)]
                W[name] = e
            elif isinstance(mod, _Transformer):
                blk = mod.transformer_blocks[0]
                e = {"norm": (f32(mod.norm.weight), f32(mod.norm.bias)), "proj_in": lin(mod.proj_in),
                     "proj_out": lin(mod.proj_out)}
                for i, ln in enumerate((blk.norm1, blk.norm2, blk.norm3), 1):
                    e[f"ln{i}"] = (f32(ln.weight), f32(ln.bias))
                a1 = blk.attn1
                # spatial self-attention: the softmax scale (and the base-2 conversion) ride on the q third of the
                # fused projection as a per-column scale of its epilogue (f32, before the one rounding: exact)
                pre = getattr(ops, "q_prescale", None) if isinstance(mod, SpatialTransformer) else None
                if pre is not None:
                    e["a1_qscale"] = torch.cat([torch.full((inner_of(a1),), float(pre)), torch.ones(2 * inner_of(a1))]
                                               ).to(device=dev, dtype=torch.float32)
                e["a1_qkv"] = wt(torch.cat([a1.to_q.weight, a1.to_k.weight, a1.to_v.weight], 0))
                e["a1_out"] = lin(a1.to_out[0])
                a2 = blk.attn2
                if a2.self_attn:
                    e["a2_qkv"] = wt(torch.cat([a2.to_q.weight, a2.to_k.weight, a2.to_v.weight], 0))
                else:
                    e["a2_q"] = wt(a2.to_q.weight)
                    e["a2_kv"] = wt(torch.cat([a2.to_k.weight, a2.to_v.weight], 0))
                    if a2.image_cross_attention:
                        e["a2_kv_ip"] = wt(torch.cat([a2.to_k_ip.weight, a2.to_v_ip.weight], 0))
                        e["a2_w2"] = a2.image_scale()  # host scalar (w2 of pm_attention), re-read with every re-pack
                e["a2_out"] = lin(a2.to_out[0])
                gw, gb = packing.pack_geglu(blk.ff.net[0].proj.weight.detach(), blk.ff.net[0].proj.bias.detach())
                e["ff1"] = (wt(gw), f32(gb))
                e["ff2"] = lin(blk.ff.net[2])
                W[name] = e
            elif isinstance(mod, Downsample):
                W[name] = (wt(packing.pack_conv3x3(mod.op.weight)), f32(mod.op.bias))
            elif isinstance(mod, Upsample):
                W[name] = (wt(packing.pack_conv3x3(mod.conv.weight)), f32(mod.conv.bias))
        W["emb_all"] = (wt(torch.cat(emb_w, 0)), f32(torch.cat(emb_b, 0)))
        W["stem"] = (wt(packing.pack_conv3x3(self.input_blocks[0][0].weight)), f32(self.input_blocks[0][0].bias))
        W["out_gn"] = (f32(self.out[0].weight), f32(self.out[0].bias))
        W["out_conv"] = (wt(packing.pack_conv3x3(self.out[2].weight)), f32(self.out[2].bias))
        W["time_embed"] = [lin(self.time_embed[0]), lin(self.time_embed[2])]
        if self.fs_condition:
            W["fps_embedding"] = [lin(self.fps_embedding[0]), lin(self.fps_embedding[2])]
        self._names = {m: n for n, m in self.named_modules()}
        # The text / image context is the same for every cross-attention block of a forward: ALL their k|v
        # projections run as one GEMM per context kind at the top of forward (16 + 16 launches of a 77- / 256-row
        # GEMM become 2); a block then reads its [rows, 2*inner] column slice of the result in place.
        for key, cat_key in (("a2_kv", "kv_text_all"), ("a2_kv_ip", "kv_img_all")):
            off, parts = 0, []
            for name, e in W.items():
                if isinstance(e, dict) and key in e:
                    e[key + "_slice"] = (off, off + e[key].shape[0])
                    off += e[key].shape[0]
                    parts.append(e.pop(key))
            W[cat_key] = torch.cat(parts, 0).contiguous() if parts else None
        self._packed = W
        for t in ([0], [self.default_fs]):  # warm the frequency-table cache for this device
            timestep_embedding(torch.tensor(t, device=dev), self.model_channels)
        return self

    # ---- graph ---------------------------------------------------------------------------------
    @staticmethod
    def _site(c, kind):
        """Name the norm / conversion site the next op-table call belongs to: (kind, pyramid level).  An op table that carries
        selected norm outputs at twice the mantissa (HipOps(parity="selective" | {...})) decides per
        site; others ignore the attribute.  Kinds: gn3 / gnt = GroupNorm + SiLU in front of a 3x3 / temporal conv, gnp = the
        transformers' GroupNorm in front of proj_in, lns1 / lns2 / lns3 (lnt*) = LayerNorm in front of the spatial (temporal)
        block's attn1 projection / attn2 projection / GEGLU, split = the f32 stream into a Downsample / Upsample / stem conv."""
        c.ops.site = (kind, c.level)

    def _gn(self, c, x, gb, eps, silu, per_frame, totals=None):
        """`totals`: {sum, sumsq} already produced by the epilogue of the op that wrote x."""
        ops = c.ops
        self._site(c, ("gn3" if per_frame else "gnt") if silu else "gnp")
        if totals is None:  # statistics a previous module's last op left behind ON this very tensor object
            tot = getattr(x, "_pm_gn_totals", None)
            if tot is not None:
                need = c.F if per_frame else c.B
                if tot.shape[0] == need:
                    totals = tot
                elif not per_frame and tot.shape[0] == c.F:
                    # per-frame sums add up to the (T,H,W) sums of their clip: by the apply kernel itself where the op table
                    # sums entries (HipOps.stats_nsum: no reduce launch; exact for the integer totals), else here
                    if getattr(ops, "stats_nsum", False) and c.fp is None:
                        totals = tot  # (int64 limbs or f32 sums alike: entries b T .. b T + T - 1 belong to clip b)
                    else:
                        totals = tot.sum(0, keepdim=True) if c.B == 1 else tot.view(c.B, c.T, *tot.shape[1:]).sum(1)
        if per_frame:
            return ops.groupnorm(x, gb[0], gb[1], eps, c.F, silu, totals=totals)
        red = c.fp.reduce_stats if c.fp is not None else None
        return ops.groupnorm(x, gb[0], gb[1], eps, c.B, silu, stats_reduce=red, totals=totals)

    def _stream_stats(self, c):
        """Statistics request for an op that writes the residual stream: per frame when a frame is a whole
        number of 64-row blocks (the (T,H,W) sums follow by addition), else over the clip; "lazy" = only if the
        epilogue can emit them (no separate statistics pass is added on their account)."""
        return (c.F if (c.H * c.W) % 64 == 0 else c.B, 32, "lazy")

    @staticmethod
    def _keep_stats(c, out_tot):
        # the statistics ride on the tensor OBJECT the op returned (a Python attribute): a view, a copy or another
        # tensor that later occupies the same address never carries them (they used to be keyed by data_ptr())
        out, tot = out_tot
        if tot is not None:
            out._pm_gn_totals = tot
        return out

    def _res_block(self, c, mod, x, out=None):
        ops, e = c.ops, c.w[self._names[mod]]
        P = c.H * c.W
        h = self._gn(c, x, e["gn1"], 1e-5, True, True)
        lo, hi = e["emb_slice"]
        # conv outputs that only feed a GroupNorm stay f32 too (one rounding less per branch)
        # every conv whose output feeds a GroupNorm also emits that norm's statistics from its epilogue
        h, tot = ops.conv3x3(h, e["conv1"], c.emb_bias[lo:hi], c.F, c.H, c.W, stream=True, stats=(c.F, 32))
        h = self._gn(c, h, e["gn2"], 1e-5, True, True, totals=tot)
        # the 1x1 skip conv puts the WHOLE stream through a 16-bit operand: carried as hi + lo (two passes), its
        # rounding was 10 % of the end-to-end error (tests/test_error_budget_gpu.py)
        skip = x if e["skip"] is None else ops.gemm(x, e["skip"][0], e["skip"][1], stream=True, split_a=True)
        if not mod.use_temporal_conv:
            return self._keep_stats(c, ops.conv3x3(h, e["conv2"][0], e["conv2"][1], c.F, c.H, c.W, residual=skip,
                                                   stream=True, stats=self._stream_stats(c), out=out))
        h, tot = ops.conv3x3(h, e["conv2"][0], e["conv2"][1], c.F, c.H, c.W, residual=skip, stream=True,
                             stats=(c.B, 32))
        ident = h
        for i, (gb, wp, b) in enumerate(e["tconv"]):
            lo_h = hi_h = None
            if c.fp is None:
                t = self._gn(c, h, gb, 1e-5, True, False, totals=tot)
            else:
                # frame shards: ONE grouped exchange carries this stage's GroupNorm partial sums and the raw boundary
                # frames; the halo frames are normalised here, with the same global totals as this rank's own frames
                part = tot if tot is not None else ops.groupnorm_stats(h, 1, 32)
                part = getattr(ops, "totals_f32", lambda t_: t_)(part)  # (the exchange carries f32 {sum, sumsq})
                tot_g, count, lo_raw, hi_raw = c.fp.exchange_stats_halo(h, P, part, h.shape[0] * (h.shape[1] // 32))
                glob = lambda _part, _cnt: (tot_g, count)
                t = ops.groupnorm(h, gb[0], gb[1], 1e-5, 1, True, stats_reduce=glob, totals=part)
                if lo_raw is not None:
                    lo_h = ops.groupnorm(lo_raw, gb[0], gb[1], 1e-5, 1, True, stats_reduce=glob, totals=part)
                if hi_raw is not None:
                    hi_h = ops.groupnorm(hi_raw, gb[0], gb[1], 1e-5, 1, True, stats_reduce=glob, totals=part)
            if i < 3:
                h, tot = self._conv_t3(c, t, wp, b, P, halo_lo=lo_h, halo_hi=hi_h, stats=(1, 32))
            else:
                st = self._stream_stats(c)
                h = self._keep_stats(c, self._conv_t3(c, t, wp, b, P, residual=ident, halo_lo=lo_h, halo_hi=hi_h,
                                                      stats=(st[0] // c.B,) + st[1:], out=out))
        return h

    @staticmethod
    def _conv_t3(c, t, wp, b, P, residual=None, halo_lo=None, halo_hi=None, stats=None, out=None):
        """The 3-tap conv over frames of every clip (zero padding at BOTH ends of each clip, openaimodel3d.py:258-269 on
        `(b c t h w)`): the taps must not reach into the neighbouring clip - one clip-aware launch (pm_conv_temporal_k3_clips)
        or, on op tables without it, one launch per clip on its row block.
        `stats` counts instances PER CLIP; totals of the clips are stacked.  -> (out f32 [F*P, Cout], totals or None)."""
        ops = c.ops
        if c.B == 1:
            return ops.conv_t3(t, wp, b, c.F, P, residual=residual, halo_lo=halo_lo, halo_hi=halo_hi, stream=True,
                               stats=stats, out=out)
        if getattr(ops, "conv_t3_clips", False):  # ONE launch: the kernel keeps the taps inside each clip
            st = None if stats is None else (stats[0] * c.B,) + tuple(stats[1:])
            return ops.conv_t3(t, wp, b, c.F, P, residual=residual, stream=True, stats=st, out=out, clip_frames=c.T)
        rows = c.T * P
        if out is None:
            out = torch.empty(c.F * P, wp.shape[0], dtype=torch.float32, device=t.device)
        tots = []
        for j in range(c.B):
            sl = slice(j * rows, (j + 1) * rows)
            _, tot = ops.conv_t3(t[sl], wp, b, c.T, P, residual=None if residual is None else residual[sl], stream=True,
                                 stats=stats, out=out[sl])
            tots.append(tot)
        return out, (None if any(x is None for x in tots) else torch.cat(tots, 0))

    def _block(self, c, e, h, mod, temporal, F, P, gather=False):
        """BasicTransformerBlock on tokens h [F*P, inner] (attention.py:242-246)."""
        ops = c.ops
        inner, heads = mod.inner, mod.heads
        v3 = lambda t, n: t.view(F, P, n)
        for which in (1, 2):
            ln = e[f"ln{which}"]  # (LayerNorm and the projection behind it are one op: ops.ln_gemm)
            key = f"a{which}_qkv"
            if key in e:  # self-attention, fused q|k|v projection
                qs = e.get("a1_qscale") if which == 1 else None
                self._site(c, f"ln{'t' if temporal else 's'}{which}")
                qkv = v3(ops.ln_gemm(h, *ln, e[key], col_scale=qs), 3 * inner)
                q, k, v = qkv[..., :inner], qkv[..., inner:2 * inner], qkv[..., 2 * inner:]
                if temporal:
                    if gather:  # frame-sharded without the pixel re-shard: all-gather K|V over frames
                        k, v = c.fp.gather_kv(qkv, inner, P)
                    if c.B == 1 or c.fp is not None:
                        a = ops.attention_temporal(q, k, v, heads)
                    else:  # batched clips: frames attend within their own clip
                        a = ops.empty(F, P, inner)
                        for j in range(c.B):
                            sl = slice(j * c.T, (j + 1) * c.T)
                            ops.attention_temporal(q[sl], k[sl], v[sl], heads, out=a[sl])
                elif getattr(ops, "fp8_attention", False) and P >= ops.fp8_min_tokens:
                    a = ops.attention_fp8(q, k, v, heads, prescaled=qs is not None)
                elif qs is not None:
                    a = ops.attention(q, k, v, heads, prescaled=True)
                else:
                    a = ops.attention(q, k, v, heads)
            else:  # spatial cross-attention: text keys shared by all frames + per-frame image keys
                self._site(c, "lns2")
                q = v3(ops.ln_gemm(h, *ln, e["a2_q"]), inner)
                lo, hi = e["a2_kv_slice"]
                kv_t = c.kv_text[:, lo:hi].unflatten(0, (c.B, -1))  # [B, 77, 2*inner] view of the batched projection
                kv_i = None
                if "a2_kv_ip_slice" in e and c.kv_img is not None:
                    lo, hi = e["a2_kv_ip_slice"]
                    # [F, 16, 2*inner] view of the per-frame tokens, or [B, n, 2*inner] shared by every frame of a clip
                    kv_i = (c.kv_img[:, lo:hi].unflatten(0, (c.B, -1)) if c.img_shared
                            else c.kv_img[:, lo:hi].unflatten(0, (F, -1)))
                a = ops.empty(F, P, inner) if c.B > 1 else None
                for j in range(c.B):  # one launch per clip: the text keys (and shared image keys) are the clip's own
                    sl = slice(j * c.T, (j + 1) * c.T) if c.B > 1 else slice(None)
                    k2 = v2 = None
                    if kv_i is not None:
                        ki = kv_i[j:j + 1] if c.img_shared else kv_i[sl]
                        k2, v2 = ki[..., :inner], ki[..., inner:]
                    got = ops.attention(q[sl], kv_t[j:j + 1, :, :inner], kv_t[j:j + 1, :, inner:], heads, k2, v2, e.get("a2_w2", 1.0),
                                        out=None if a is None else a[sl])
                    a = got if a is None else a
            h = ops.gemm(a.view(F * P, inner), *e[f"a{which}_out"], residual=h, stream=True)
        self._site(c, f"ln{'t' if temporal else 's'}3")
        g = ops.ln_gemm(h, *e["ln3"], e["ff1"][0], e["ff1"][1], act="geglu")
        # the block's last add: its only consumer is proj_out's A operand (16-bit anyway), so the sum
        # (formed in f32 against the f32 stream) is stored as 16 bit
        return ops.gemm(g, e["ff2"][0], e["ff2"][1], residual=h)

    def _transformer(self, c, mod, x, temporal, out=None):
        ops, e = c.ops, c.w[self._names[mod]]
        F, P = c.F, c.H * c.W
        h = self._gn(c, x, e["norm"], 1e-6, False, not temporal)
        sharded = temporal and c.fp is not None and P % c.fp.world == 0 and not c.fp.kv_gather
        gather = temporal and c.fp is not None and not sharded  # (pixel count not divisible: tiny maps)
        if sharded:  # frames <-> pixels: the temporal block needs all T frames of a pixel, nothing else
            h = c.fp.frames_to_pixels(h, P)
            F, P = c.fp.total_frames, P // c.fp.world
        h = ops.gemm(h, *e["proj_in"], stream=True)  # the block's own residual stream (f32)
        h = self._block(c, e, h, mod, temporal, F, P, gather)
        if sharded:
            h = c.fp.pixels_to_frames(h, c.H * c.W)
        return self._keep_stats(c, ops.gemm(h, *e["proj_out"], residual=x, stream=True, stats=self._stream_stats(c),
                                            out=out))

    def _run(self, c, seq, h, out=None):
        """`out`: an f32 [rows, channels] view (row stride >= channels) the LAST op of the sequence writes into -
        one half of a skip-concatenation buffer (forward), so that no torch.cat pass is needed."""
        layers = list(seq)
        for li, layer in enumerate(layers):
            dst = out if li == len(layers) - 1 else None
            if isinstance(layer, ResBlock):
                h = self._res_block(c, layer, h, dst)
            elif isinstance(layer, SpatialTransformer):
                h = self._transformer(c, layer, h, False, dst)
            elif isinstance(layer, TemporalTransformer):
                h = self._transformer(c, layer, h, True, dst)
            elif isinstance(layer, Downsample):
                wp, b = c.w[self._names[layer]]
                self._site(c, "split")
                h = c.ops.conv3x3(h, wp, b, c.F, c.H, c.W, stride=2, stream=True, out=dst)
                c.H, c.W, c.level = (c.H + 1) // 2, (c.W + 1) // 2, c.level + 1
            elif isinstance(layer, Upsample):
                wp, b = c.w[self._names[layer]]
                # (r03 kept the gathered form for frame shards because tests/test_segmented_gpu.py aborted with the written-out
                # interpolation; the abort was the process group's watchdog querying an event of a capturing stream -
                # frame_parallel.FrameParallel._comm - and had nothing to do with this op: one form everywhere again)
                self._site(c, "split")
                h = c.ops.conv3x3(h, wp, b, c.F, c.H, c.W, upsample=True, stream=True, out=dst)
                c.H, c.W, c.level = 2 * c.H, 2 * c.W, c.level - 1
            elif isinstance(layer, nn.Conv2d):  # stem
                wp, b = c.w["stem"]
                self._site(c, "split")
                h = c.ops.conv3x3(h, wp, b, c.F, c.H, c.W, stream=True, out=dst)
            else:
                raise TypeError(type(layer))
        return h

    @staticmethod
    def _out_geometry(seq, ch, H, W):
        """(channels, H, W) of the stream after the layers of `seq` (static: from the module definitions)."""
        for layer in seq:
            if isinstance(layer, ResBlock):
                ch = layer.out_channels
            elif isinstance(layer, nn.Conv2d):
                ch = layer.out_channels
            elif isinstance(layer, Downsample):
                H, W = (H + 1) // 2, (W + 1) // 2
            elif isinstance(layer, Upsample):
                H, W = 2 * H, 2 * W
        return ch, H, W

    def _embed(self, c, timesteps, fs):
        ops, W = c.ops, c.w
        mlp = lambda p, v: ops.gemv(p[1][0], ops.gemv(p[0][0], v, p[0][1], act="silu"), p[1][1])
        dev = ops.device
        emb = mlp(W["time_embed"], _timestep_row(ops, timesteps, self.model_channels))
        if self.fs_condition:
            if fs is None:
                fs = torch.tensor([self.default_fs], dtype=torch.long, device=dev)
            emb = emb + mlp(W["fps_embedding"], _timestep_row(ops, fs, self.model_channels))
        # every ResBlock's  b_conv1 + b_emb + W_emb . silu(emb)  in one launch
        return ops.gemv(W["emb_all"][0], emb, W["emb_all"][1], silu_in=True)

    def forward(self, x, timesteps, context=None, features_adapter=None, fs=None, **kwargs):
        """x (1, C_in, t_local, h, w), timesteps (1,), context (1, 77 + 16*T, 1024), fs (1,) ->
        (1, C_out, t_local, h, w) in f32.  In frame-sharded mode x holds this rank's
        frames and `context` is the full-clip context (image tokens are sliced by `fp`)."""
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # the training seam (model.py:926-942 -> ddpm3d.py:741-797 -> apply_model): the HIP op table is forward-only
            # (packed, detached weights), so the graph is walked with differentiable torch ops over the module's own
            # parameters instead; inference - eval() or no_grad - never takes this branch
            from . import unet_train
            return unet_train.forward(self, x, timesteps, context, fs, features_adapter)
        with torch.no_grad():
            try:
                return self._forward(x, timesteps, context, features_adapter, fs, **kwargs)
            finally:
                # (a forward that raises must not leave its last site on the shared op table - first-stage /
                # Resampler calls on the same HipOps would silently take a U-Net site's [hi | lo] decision)
                if self.ops is not None and hasattr(self.ops, "site"):
                    self.ops.site = None

    def _forward(self, x, timesteps, context=None, features_adapter=None, fs=None, **kwargs):
        if self.ops is None:
            raise RuntimeError("UNetModel.bind(ops) must be called before forward (no implicit CPU fallback)")
        packed = self.packed()
        b, cin, t, hh, ww = x.shape
        # b > 1: clips batched along the rows - the sampler's cond / uncond pair of a CFG step as ONE forward over 2 x 16
        # frames (weights read once, every grid twice as full: VERDICT r03 #4a).  Per-frame ops see 2T frames; the ops that
        # couple frames ((T,H,W) GroupNorm, temporal conv, temporal attention) and the cross-attention keep the clips apart.
        # The reference itself runs batch size 1 (model.py:794).
        _unsupported(b > 1 and self.fp is not None, "batched clips in frame-sharded mode")
        c = _Ctx()
        c.ops, c.fp, c.w = self.ops, self.fp, packed
        if hasattr(self.ops, "begin_forward"):
            self.ops.begin_forward()  # (HipOps: one memset over this stream's GroupNorm-totals arena)
        c.B, c.T, c.F, c.H, c.W = b, t, b * t, hh, ww
        c.level = 0  # (pyramid level of the norm sites, UNetModel._site: +1 behind every Downsample, -1 behind every Upsample)
        ops = c.ops
        T_total = t if c.fp is None else c.fp.total_frames
        ctx = context.to(device=ops.device, dtype=ops.dtype)
        assert ctx.shape[0] == b, "one context per clip"
        c.img_shared = False
        if ctx.shape[1] == 77 + T_total * 16:  # per-frame image conditioning (openaimodel3d.py:559-564)
            c.ctx_text = ctx[:, :77].reshape(b * 77, -1).contiguous()
            img = ctx[:, 77:].reshape(b, T_total, 16, -1)
            if c.fp is not None:
                img = img[:, c.fp.frame_offset:c.fp.frame_offset + t]
            c.ctx_img = img.reshape(b * t * 16, -1).contiguous()
        else:
            # openaimodel3d.py:565-566: the same context for every frame; CrossAttention still splits it at the
            # hard-coded text length 77 (attention.py:89-99): what lies beyond are image tokens shared by all frames
            c.ctx_text = ctx[:, :77].reshape(b * 77, -1).contiguous()
            c.ctx_img = ctx[:, 77:].reshape(b * (ctx.shape[1] - 77), -1).contiguous() if ctx.shape[1] > 77 else None
            c.img_shared = True
        c.kv_text = ops.gemm(c.ctx_text, c.w["kv_text_all"]) if c.w["kv_text_all"] is not None else None
        c.kv_img = (ops.gemm(c.ctx_img, c.w["kv_img_all"])
                    if c.w["kv_img_all"] is not None and c.ctx_img is not None else None)
        # batched clips share ONE timestep (the CFG pair of a DDIM step): checked where that costs nothing - a host tensor;
        # a device tensor would need a synchronising read, which is illegal inside the sampler's graph capture
        # and fs likewise (_embed reads element 0 of both for every clip, so a batch with per-clip values used to
        # get clip 0's embedding silently).  Device tensors are checked whenever a synchronising read is legal, i.e. outside a
        # stream capture (the sampler's own batched call: t = cat([t, t]), validated on its warm-up forward).
        if b > 1:
            for name, v in (("timestep", timesteps), ("fs", fs)):
                # (a device tensor costs a blocking read-back per eager forward: checked on the FIRST forward of a
                # batch size only - the sampler's warm-up forward - host tensors every time)
                if torch.is_tensor(v) and v.numel() > 1 and not (v.is_cuda and torch.cuda.is_current_stream_capturing()):
                    if v.is_cuda:
                        seen = self.__dict__.setdefault("_batched_checked", set())
                        if (name, b) in seen:
                            continue
                        seen.add((name, b))
                    if not bool((v == v.reshape(-1)[0]).all()):
                        raise NotImplementedError(f"batched clips share one {name} (the CFG pair of a DDIM step)")
        c.emb_bias = self._embed(c, timesteps, fs)

        xr = x.permute(1, 0, 2, 3, 4).reshape(cin, b * t, hh * ww)  # [C, (clip, frame), pixel]; a view for b == 1
        if x.dtype == torch.float32:
            h = ops.pack_input(xr.contiguous(), None)
        else:
            h = xr.permute(1, 2, 0).reshape(b * t * hh * ww, cin).to(ops.dtype).contiguous()
        # Skip concatenations without a copy pass: decoder block k reads cat([h, skip_k]) (openaimodel3d.py:598-600).
        # Its input buffer [rows, C_h + C_skip] is allocated when the encoder produces skip_k: the encoder block's last
        # op writes the right half (and the encoder simply continues on that strided view), the op that produces the
        # decoder's h writes the left half.
        nblk = len(self.output_blocks)
        cat_in = [blk[0].channels for blk in self.output_blocks]  # input channels of every decoder block
        bufs = []
        ch = None
        for i, module in enumerate(self.input_blocks):
            seqs = [module] + ([self.init_attn] if i == 0 and self.addition_attention else [])
            gh, gw = c.H, c.W
            for sq in seqs:
                ch, gh, gw = self._out_geometry(sq, ch, gh, gw)
            total = cat_in[nblk - 1 - i]
            buf = torch.empty(c.F * gh * gw, total, dtype=torch.float32, device=h.device)
            for si, sq in enumerate(seqs):
                h = self._run(c, sq, h, buf[:, total - ch:] if si == len(seqs) - 1 else None)
            assert h.data_ptr() == buf[:, total - ch:].data_ptr() and (c.H, c.W) == (gh, gw)
            if features_adapter is not None and (i + 1) % 3 == 0:
                # plug-in adapter features (openaimodel3d.py:589-593): added to the stream behind input blocks 2, 5, 8, 11 -
                # in place in the skip-concatenation buffer, so the skip carries them too, as `hs.append(h)` after the add does
                fa = features_adapter[i // 3]
                assert fa.shape == (c.F, ch, c.H, c.W), (tuple(fa.shape), (c.F, ch, c.H, c.W))
                h.add_(fa.to(device=h.device, dtype=torch.float32).permute(0, 2, 3, 1).reshape(c.F * c.H * c.W, ch))
                if hasattr(h, "_pm_gn_totals"):
                    del h._pm_gn_totals  # (the producer's GroupNorm statistics describe the tensor before the add)
            bufs.append((buf, total - ch, c.H, c.W))
        if features_adapter is not None:
            assert len(features_adapter) == len(self.input_blocks) // 3, "Wrong features_adapter"
        buf, c_h, sh, sw = bufs.pop()
        h = self._run(c, self.middle_block, h, buf[:, :c_h])
        for k, module in enumerate(self.output_blocks):
            assert (sh, sw) == (c.H, c.W) and h.data_ptr() == buf.data_ptr()
            h = buf  # = cat([h, skip], dim=1)
            if k + 1 < nblk:
                buf, c_h, sh, sw = bufs.pop()
                h = self._run(c, module, h, buf[:, :c_h])
            else:
                h = self._run(c, module, h)
        h = self._gn(c, h, c.w["out_gn"], 1e-5, True, True)
        y = ops.conv3x3(h, c.w["out_conv"][0], c.w["out_conv"][1], c.F, c.H, c.W, stream=True)
        y = ops.unpack_output(y, c.F, hh * ww)  # [C, (clip, frame), pixel]
        ops.site = None  # (calls outside the U-Net - first stage, Resampler - belong to no site)
        if b == 1:
            return y.reshape(1, self.out_channels, t, hh, ww)
        return y.reshape(self.out_channels, b, t, hh, ww).permute(1, 0, 2, 3, 4).contiguous()
