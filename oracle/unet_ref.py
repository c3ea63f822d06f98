"""ORACLE (test infrastructure, not product code): CPU restatement of the reference 3-D U-Net
forward in the reference's own NCHW / eager-PyTorch formulation, written functionally over a
state_dict with the reference's key names (so it runs on real or synthesised checkpoints alike).

Pinned by tests/test_oracle_vs_reference.py against the REAL reference modules imported from
/root/reference in the build container, and by the golden fixtures under tests/golden/ that
oracle/make_golden.py captured from them.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this file.

Reference call sites (relative to /root/reference/DynamiCrafter/lvdm):
  unet_forward         modules/networks/openaimodel3d.py:552-607 (graph built at :387-550)
  timestep_embedding   models/utils_diffusion.py:8-28  (bf16 arange -> bf16 freqs, [cos | sin])
  res_block            modules/networks/openaimodel3d.py:213-239
  temporal_conv_block  modules/networks/openaimodel3d.py:258-282
  spatial_transformer  modules/attention.py:294-310
  temporal_transformer modules/attention.py:365-412 (only_self_att: attn1 AND attn2 are self-attention)
  basic_block          modules/attention.py:242-246
  cross_attention      modules/attention.py:81-144
  feed_forward         modules/attention.py:415-442
  up/downsample        modules/networks/openaimodel3d.py:51-109
"""
import math

import torch
import torch.nn.functional as F


def timestep_embedding(timesteps, dim, max_period=10000):
    half = dim // 2
    # NOTE the reference quantises the frequency table to bf16 (utils_diffusion.py:19-21)
    freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.bfloat16) / half)
    args = timesteps[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


class _SD:
    """state_dict view with a key prefix."""

    def __init__(self, sd, prefix=""):
        self.sd, self.prefix = sd, prefix

    def __getitem__(self, k):
        return self.sd[self.prefix + k].float()

    def has(self, k):
        return (self.prefix + k) in self.sd

    def sub(self, p):
        return _SD(self.sd, self.prefix + p)


def _linear(sd, name, x):
    w = sd[name + ".weight"]
    if w.dim() == 3:  # Conv1d kernel 1 (init_attn proj_in / proj_out)
        w = w[:, :, 0]
    b = sd[name + ".bias"] if sd.has(name + ".bias") else None
    return F.linear(x, w, b)


def _gn(sd, name, x, eps):
    return F.group_norm(x, 32, sd[name + ".weight"], sd[name + ".bias"], eps)


def _ln(sd, name, x):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], 1e-5)


def _heads(t, h):
    b, n, _ = t.shape
    return t.reshape(b, n, h, -1).permute(0, 2, 1, 3).reshape(b * h, n, -1)


def _unheads(t, h):
    bh, n, d = t.shape
    return t.reshape(bh // h, h, n, d).permute(0, 2, 1, 3).reshape(bh // h, n, h * d)


def _softmax_attn(q, k, v, scale, chunk=8):
    # (b*h, n, d); chunked over the batch axis to bound the materialised score tensor
    out = torch.empty_like(q)
    for i in range(0, q.shape[0], chunk):
        sim = torch.einsum("bid,bjd->bij", q[i:i + chunk], k[i:i + chunk]) * scale
        out[i:i + chunk] = torch.einsum("bij,bjd->bid", sim.softmax(dim=-1), v[i:i + chunk])
    return out


def cross_attention(sd, x, context=None, text_len=77):
    h = sd["to_q.weight"].shape[0] // 64
    scale = 64 ** -0.5
    q = _heads(_linear(sd, "to_q", x), h)
    ctx = x if context is None else context
    out_ip = None
    if context is not None and sd.has("to_k_ip.weight"):
        ctx, ctx_img = context[:, :text_len], context[:, text_len:]
        k_ip = _heads(_linear(sd, "to_k_ip", ctx_img), h)
        v_ip = _heads(_linear(sd, "to_v_ip", ctx_img), h)
        out_ip = _unheads(_softmax_attn(q, k_ip, v_ip, scale), h)
    elif context is not None:
        ctx = context[:, :text_len]
    k = _heads(_linear(sd, "to_k", ctx), h)
    v = _heads(_linear(sd, "to_v", ctx), h)
    out = _unheads(_softmax_attn(q, k, v, scale), h)
    if out_ip is not None:
        # image_cross_attention_scale = 1.0; learnable in the 256 yaml (attention.py:77-78,138-142): x (tanh(alpha) + 1)
        out = out + (1.0 * out_ip * (torch.tanh(sd["alpha"].float()) + 1) if sd.has("alpha") else 1.0 * out_ip)
    return _linear(sd, "to_out.0", out)


def feed_forward(sd, x):
    xg = _linear(sd, "net.0.proj", x)
    xv, gate = xg.chunk(2, dim=-1)
    return _linear(sd, "net.2", xv * F.gelu(gate))


def basic_block(sd, x, context=None):
    x = cross_attention(sd.sub("attn1."), _ln(sd, "norm1", x)) + x
    x = cross_attention(sd.sub("attn2."), _ln(sd, "norm2", x), context) + x
    x = feed_forward(sd.sub("ff."), _ln(sd, "norm3", x)) + x
    return x


def spatial_transformer(sd, x, context):
    b, c, h, w = x.shape
    y = _gn(sd, "norm", x, 1e-6)
    y = y.permute(0, 2, 3, 1).reshape(b, h * w, c)
    y = _linear(sd, "proj_in", y)
    y = basic_block(sd.sub("transformer_blocks.0."), y, context)
    y = _linear(sd, "proj_out", y)
    return y.reshape(b, h, w, c).permute(0, 3, 1, 2) + x


def temporal_transformer(sd, x):
    """x: (b, c, t, h, w)."""
    b, c, t, h, w = x.shape
    y = _gn(sd, "norm", x, 1e-6)
    y = y.permute(0, 3, 4, 2, 1).reshape(b * h * w, t, c)
    y = _linear(sd, "proj_in", y)
    y = basic_block(sd.sub("transformer_blocks.0."), y, None)
    y = _linear(sd, "proj_out", y)
    return y.reshape(b, h, w, t, c).permute(0, 4, 3, 1, 2) + x


def temporal_conv_block(sd, x):
    """x: (b, c, t, h, w)."""
    y = x
    for i, conv_idx in ((1, 2), (2, 3), (3, 3), (4, 3)):
        y = F.silu(F.group_norm(y, 32, sd[f"conv{i}.0.weight"], sd[f"conv{i}.0.bias"], 1e-5))
        y = F.conv3d(y, sd[f"conv{i}.{conv_idx}.weight"], sd[f"conv{i}.{conv_idx}.bias"], padding=(1, 0, 0))
    return x + y


def res_block(sd, x, emb, batch):
    h = F.conv2d(F.silu(_gn(sd, "in_layers.0", x, 1e-5)), sd["in_layers.2.weight"], sd["in_layers.2.bias"], padding=1)
    emb_out = F.linear(F.silu(emb), sd["emb_layers.1.weight"], sd["emb_layers.1.bias"])
    h = h + emb_out[:, :, None, None]
    h = F.conv2d(F.silu(_gn(sd, "out_layers.0", h, 1e-5)), sd["out_layers.3.weight"], sd["out_layers.3.bias"], padding=1)
    if sd.has("skip_connection.weight"):
        x = F.conv2d(x, sd["skip_connection.weight"], sd["skip_connection.bias"])
    h = x + h
    if sd.has("temopral_conv.conv1.0.weight"):
        bt, c, hh, ww = h.shape
        h5 = h.reshape(batch, bt // batch, c, hh, ww).permute(0, 2, 1, 3, 4)
        h5 = temporal_conv_block(sd.sub("temopral_conv."), h5)
        h = h5.permute(0, 2, 1, 3, 4).reshape(bt, c, hh, ww)
    return h


def _run_sequential(sd, prefix, h, emb, context, batch):
    """Dispatch the children of a TimestepEmbedSequential by the parameters they own."""
    idx = 0
    while True:
        p = f"{prefix}{idx}."
        s = sd.sub(p)
        if s.has("in_layers.0.weight"):
            h = res_block(s, h, emb, batch)
        elif s.has("transformer_blocks.0.attn2.to_k_ip.weight"):
            h = spatial_transformer(s, h, context)
        elif s.has("transformer_blocks.0.attn1.to_q.weight"):
            bt, c, hh, ww = h.shape
            h5 = h.reshape(batch, bt // batch, c, hh, ww).permute(0, 2, 1, 3, 4)
            h5 = temporal_transformer(s, h5)
            h = h5.permute(0, 2, 1, 3, 4).reshape(bt, c, hh, ww)
        elif s.has("op.weight"):  # Downsample
            h = F.conv2d(h, s["op.weight"], s["op.bias"], stride=2, padding=1)
        elif s.has("conv.weight"):  # Upsample: nearest x2 (in f32) then conv
            h = F.interpolate(h.float(), scale_factor=2, mode="nearest")
            h = F.conv2d(h, s["conv.weight"], s["conv.bias"], padding=1)
        elif s.has("weight") and s["weight"].dim() == 4:  # bare stem conv
            h = F.conv2d(h, s["weight"], s["bias"], padding=1)
        else:
            return h
        idx += 1


def _count_blocks(sd, stem):
    n = 0
    while any(k.startswith(f"{stem}.{n}.") for k in sd.sd):
        n += 1
    return n


@torch.no_grad()
def unet_forward(state_dict, x, timesteps, context, fs, model_channels=320, features_adapter=None):
    """x (b, 8, t, h, w), timesteps (b,) long, context (b, 77 + 16 t, 1024), fs (b,) long
    -> (b, 4, t, h, w); everything in f32."""
    sd = _SD(state_dict)
    b, _, t, _, _ = x.shape
    emb = _linear(sd, "time_embed.2", F.silu(_linear(sd, "time_embed.0", timestep_embedding(timesteps, model_channels))))
    l_context = context.shape[1]
    if l_context == 77 + t * 16:
        ctx_text = context[:, :77].repeat_interleave(repeats=t, dim=0)
        ctx_img = context[:, 77:].reshape(b, t, 16, -1).reshape(b * t, 16, -1)
        context = torch.cat([ctx_text, ctx_img], dim=1)
    else:
        context = context.repeat_interleave(repeats=t, dim=0)
    emb = emb.repeat_interleave(repeats=t, dim=0)
    h = x.permute(0, 2, 1, 3, 4).reshape(b * t, x.shape[1], x.shape[3], x.shape[4])
    if sd.has("fps_embedding.0.weight"):
        fs_emb = _linear(sd, "fps_embedding.2", F.silu(_linear(sd, "fps_embedding.0", timestep_embedding(fs, model_channels))))
        emb = emb + fs_emb.repeat_interleave(repeats=t, dim=0)

    hs = []
    n_in = _count_blocks(sd, "input_blocks")
    for i in range(n_in):
        h = _run_sequential(sd, f"input_blocks.{i}.", h, emb, context, b)
        if i == 0 and sd.has("init_attn.0.norm.weight"):
            h = _run_sequential(sd, "init_attn.", h, emb, context, b)
        if (i + 1) % 3 == 0 and features_adapter is not None:  # plug-in adapter features (openaimodel3d.py:589-593)
            h = h + features_adapter[i // 3]
        hs.append(h)
    if features_adapter is not None:
        assert len(features_adapter) == n_in // 3, "Wrong features_adapter"
    h = _run_sequential(sd, "middle_block.", h, emb, context, b)
    for i in range(_count_blocks(sd, "output_blocks")):
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_sequential(sd, f"output_blocks.{i}.", h, emb, context, b)
    y = F.conv2d(F.silu(_gn(sd, "out.0", h, 1e-5)), sd["out.2.weight"], sd["out.2.bias"], padding=1)
    return y.reshape(b, t, y.shape[1], y.shape[2], y.shape[3]).permute(0, 2, 1, 3, 4)
