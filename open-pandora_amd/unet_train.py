"""Differentiable forward of `unet.UNetModel` for the training seam ("when
`torch.is_grad_enabled()` the dispatch table falls back to eager PyTorch ops").

`WorldModel.training_step` (model.py:926-942) reaches the U-Net through `LatentDiffusion.p_losses`
(ddpm3d.py:741-797) -> `apply_model` (:724-739) -> `DiffusionWrapper.forward` (:1077-1081) and then
calls `loss.backward()`.  The HIP op table is forward-only (packed, detached weights), so in that one
situation - module in training mode, autograd on, parameters that require grad - `UNetModel.forward`
dispatches here: the SAME graph, walked over the SAME `nn.Parameter`s (no packed copies, so autograd
reaches the storage `configure_optimizers` hands to AdamW, model.py:951-962), with plain torch ops in
the parameters' dtype on whatever device they live on.  It mirrors the reference modules' arithmetic
(openaimodel3d.py:36-48,213-239,258-282,552-607; attention.py:81-144,242-246,294-310,365-412,415-442)
including dropout in training mode; attention goes through `scaled_dot_product_attention` (the same
softmax(q k^T / sqrt d) v, without materialising N x N scores for the backward pass).

This is a product path for TRAINING only; inference never comes here (`forward` raises if `bind()`
was not called instead of falling back).  Batch sizes > 1 are supported (the trainer batches clips).
"""
import torch
import torch.nn.functional as F

from . import unet as U


def _lin(mod, x):
    w = mod.weight
    if w.dim() > 2:  # kernel-1 Conv1d / Conv2d of the non-linear projection variants
        w = w.reshape(w.shape[0], w.shape[1])
    return F.linear(x, w, mod.bias)


def _attend(q, k, v, heads):
    """(b, n, h*d) x (b, m, h*d) -> (b, n, h*d): softmax(q k^T * d^-1/2) v per head (attention.py:100-123)."""
    b, n, _ = q.shape
    split = lambda t: t.reshape(b, t.shape[1], heads, -1).transpose(1, 2)
    o = F.scaled_dot_product_attention(split(q), split(k), split(v))
    return o.transpose(1, 2).reshape(b, n, -1)


def _cross_attention(mod, x, context=None):
    """attention.py:81-144: self-attention when `context` is None; else text keys (the first 77 tokens,
    hard-coded) plus, for image_cross_attention, the image tokens behind them with scale 1.0."""
    q = mod.to_q(x)
    if context is None:
        out = _attend(q, mod.to_k(x), mod.to_v(x), mod.heads)
    elif mod.image_cross_attention:
        text, img = context[:, :77], context[:, 77:]
        out = _attend(q, mod.to_k(text), mod.to_v(text), mod.heads)
        out_ip = _attend(q, mod.to_k_ip(img), mod.to_v_ip(img), mod.heads)
        alpha = getattr(mod, "alpha", None)  # image_cross_attention_scale_learnable (attention.py:138-142): differentiable
        out = out + (1.0 * out_ip if alpha is None else 1.0 * out_ip * (torch.tanh(alpha) + 1))
    else:
        # (attention.py:96-99: without image cross-attention the context is still cut at the text length)
        text = context[:, :77]
        out = _attend(q, mod.to_k(text), mod.to_v(text), mod.heads)
    return mod.to_out(out)  # Linear + Dropout(0.0)


def _block(blk, x, context=None):
    """BasicTransformerBlock._forward (attention.py:242-246)."""
    x = _cross_attention(blk.attn1, blk.norm1(x)) + x
    x = _cross_attention(blk.attn2, blk.norm2(x), None if blk.attn2.self_attn else context) + x
    g = blk.ff.net[0].proj(blk.norm3(x))
    val, gate = g.chunk(2, dim=-1)
    return blk.ff.net[2](blk.ff.net[1](val * F.gelu(gate))) + x


def _spatial_transformer(mod, x, context):
    """attention.py:294-310 on (b*t, c, h, w)."""
    n, c, h, w = x.shape
    y = mod.norm(x).permute(0, 2, 3, 1).reshape(n, h * w, c)
    y = _lin(mod.proj_out, _block(mod.transformer_blocks[0], _lin(mod.proj_in, y), context))
    return y.reshape(n, h, w, c).permute(0, 3, 1, 2) + x


def _temporal_transformer(mod, x, batch):
    """attention.py:365-412 (+ the rearranges of TimestepEmbedSequential, openaimodel3d.py:43-46) on (b*t, c, h, w):
    GroupNorm over (t, h, w), attention along t at every pixel, attn1 and attn2 both self-attention."""
    n, c, h, w = x.shape
    t = n // batch
    x5 = x.reshape(batch, t, c, h, w).permute(0, 2, 1, 3, 4)  # b c t h w
    y = mod.norm(x5).permute(0, 3, 4, 2, 1).reshape(batch * h * w, t, c)
    y = _lin(mod.proj_out, _block(mod.transformer_blocks[0], _lin(mod.proj_in, y)))
    y = y.reshape(batch, h, w, t, c).permute(0, 4, 3, 1, 2) + x5
    return y.permute(0, 2, 1, 3, 4).reshape(n, c, h, w)


def _res_block(mod, x, emb, batch):
    """ResBlock._forward (openaimodel3d.py:213-239) + TemporalConvBlock.forward (:275-282)."""
    h = mod.in_layers(x)
    h = h + mod.emb_layers(emb).type(h.dtype)[:, :, None, None]
    h = mod.skip_connection(x) + mod.out_layers(h)
    if mod.use_temporal_conv:
        n, c, hh, ww = h.shape
        h5 = h.reshape(batch, n // batch, c, hh, ww).permute(0, 2, 1, 3, 4)
        tc = mod.temopral_conv
        h5 = h5 + tc.conv4(tc.conv3(tc.conv2(tc.conv1(h5))))
        h = h5.permute(0, 2, 1, 3, 4).reshape(n, c, hh, ww)
    return h


def _run(seq, h, emb, context, batch):
    for layer in seq:
        if isinstance(layer, U.ResBlock):
            h = _res_block(layer, h, emb, batch)
        elif isinstance(layer, U.SpatialTransformer):
            h = _spatial_transformer(layer, h, context)
        elif isinstance(layer, U.TemporalTransformer):
            h = _temporal_transformer(layer, h, batch)
        elif isinstance(layer, U.Downsample):
            h = layer.op(h)
        elif isinstance(layer, U.Upsample):  # nearest x2 in f32, then the conv (openaimodel3d.py:98-109)
            h = layer.conv(F.interpolate(h.float(), scale_factor=2, mode="nearest").to(h.dtype))
        else:  # the stem conv
            h = layer(h)
    return h


def forward(model, x, timesteps, context=None, fs=None, features_adapter=None):
    """`UNetModel.forward` with autograd (openaimodel3d.py:552-607): x (b, C_in, t, h, w), timesteps (b,),
    context (b, 77 + 16 t | L, D), fs (b,) -> (b, C_out, t, h, w)."""
    b, _, t, _, _ = x.shape
    pdt = model.time_embed[0].weight.dtype
    x = x.to(pdt)
    emb = model.time_embed(U.timestep_embedding(timesteps, model.model_channels).to(pdt))
    context = context.to(pdt)
    if context.shape[1] == 77 + t * 16:  # per-frame image conditioning (:559-564)
        text = context[:, :77].repeat_interleave(repeats=t, dim=0)
        img = context[:, 77:].reshape(b * t, 16, context.shape[-1])
        context = torch.cat([text, img], dim=1)
    else:
        context = context.repeat_interleave(repeats=t, dim=0)
    emb = emb.repeat_interleave(repeats=t, dim=0)
    if model.fs_condition:
        if fs is None:
            fs = torch.tensor([model.default_fs] * b, dtype=torch.long, device=x.device)
        fs_emb = model.fps_embedding(U.timestep_embedding(fs, model.model_channels).to(pdt))
        emb = emb + fs_emb.repeat_interleave(repeats=t, dim=0)
    h = x.permute(0, 2, 1, 3, 4).reshape(b * t, x.shape[1], x.shape[3], x.shape[4])
    hs = []
    for i, module in enumerate(model.input_blocks):
        h = _run(module, h, emb, context, b)
        if i == 0 and model.addition_attention:
            h = _run(model.init_attn, h, emb, context, b)
        if (i + 1) % 3 == 0 and features_adapter is not None:  # openaimodel3d.py:589-593
            h = h + features_adapter[i // 3].to(h.dtype)
        hs.append(h)
    if features_adapter is not None:
        assert len(features_adapter) == len(model.input_blocks) // 3, "Wrong features_adapter"
    h = _run(model.middle_block, h, emb, context, b)
    for module in model.output_blocks:
        h = _run(module, torch.cat([h, hs.pop()], dim=1), emb, context, b)
    y = model.out(h)
    return y.reshape(b, t, *y.shape[1:]).permute(0, 2, 1, 3, 4)
