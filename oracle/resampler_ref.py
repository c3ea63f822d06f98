"""ORACLE (test infrastructure, not product code): CPU f32 restatement of the image-context Resampler,
functional over a state_dict with the reference's key names.  Pinned against the real
lvdm.modules.encoders.resampler.Resampler (imported from /root/reference by oracle/make_golden.py
--resampler) through tests/golden/resampler.npz.

Reference (relative to /root/reference/DynamiCrafter/lvdm/modules/encoders/resampler.py):
  Resampler.forward          :131-144 (latents.repeat, proj_in, depth x (attn + ff, both residual), proj_out, norm_out)
  PerceiverAttention.forward :65-93   (norm1(x), norm2(latents), q from latents, k|v from cat(x, latents),
                                       scores (q*s)(k*s)^T with s = dim_head^-1/4, softmax in f32)
  FeedForward                :27-34   (LayerNorm, Linear no bias, erf GELU, Linear no bias)
"""
import torch
import torch.nn.functional as F


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"].float(), sd[p + ".bias"].float(), 1e-5)


def perceiver_attention(sd, p, x, latents, heads):
    x, latents = _ln(sd, p + ".norm1", x), _ln(sd, p + ".norm2", latents)
    b, l, _ = latents.shape
    q = latents @ sd[p + ".to_q.weight"].float().t()
    k, v = (torch.cat((x, latents), dim=-2) @ sd[p + ".to_kv.weight"].float().t()).chunk(2, dim=-1)
    split = lambda t: t.view(b, t.shape[1], heads, -1).transpose(1, 2)
    q, k, v = split(q), split(k), split(v)
    s = q.shape[-1] ** -0.25
    w = torch.softmax(((q * s) @ (k * s).transpose(-2, -1)).float(), dim=-1)
    out = (w @ v).permute(0, 2, 1, 3).reshape(b, l, -1)
    return out @ sd[p + ".to_out.weight"].float().t()


def feed_forward(sd, p, x):
    h = _ln(sd, p + ".0", x) @ sd[p + ".1.weight"].float().t()
    return F.gelu(h) @ sd[p + ".3.weight"].float().t()


def resampler_forward(sd, x, heads):
    """x [B, n1, embedding_dim] f32 -> [B, nq, output_dim]."""
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("layers."))
    x = x.float()
    latents = sd["latents"].float().repeat(x.size(0), 1, 1)
    x = x @ sd["proj_in.weight"].float().t() + sd["proj_in.bias"].float()
    for i in range(depth):
        latents = perceiver_attention(sd, f"layers.{i}.0", x, latents, heads) + latents
        latents = feed_forward(sd, f"layers.{i}.1", latents) + latents
    latents = latents @ sd["proj_out.weight"].float().t() + sd["proj_out.bias"].float()
    return _ln(sd, "norm_out", latents)
