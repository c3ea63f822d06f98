"""Image-context Resampler: drop-in for lvdm.modules.encoders.resampler.Resampler
(resampler.py:96-144, PerceiverAttention :48-93, FeedForward :27-34) on the MI355X op table.

Same constructor keywords and state-dict keys (`latents`, `proj_in`, `proj_out`, `norm_out`,
`layers.{i}.0.{norm1,norm2,to_q,to_kv,to_out}`, `layers.{i}.1.{0,1,3}`), so the `image_proj_stage_config`
target in the yaml (inference_512_v1.0.yaml:91-102) can simply point here; `bind(HipOps(...))` attaches
the kernels (no fallback: forward raises without an op table).

Graph (tokens as [rows, channels] matrices, f32 residual stream like the U-Net):
  x -> proj_in (pm_gemm); per layer: LayerNorm of x and of the latents (pm_layernorm) written behind one
  another = the keys of cat(x, latents) (resampler.py:76), ONE fused k|v projection over those rows, q from
  the latents, flash attention with head dim 64 over the 257 + 256 keys (pm_attention; the reference's
  q*s . k*s with s = 64^-1/4 is the same score as (q . k) * 64^-1/2), to_out + residual, LayerNorm ->
  Linear -> erf GELU (PM_ACT_GELU epilogue) -> Linear + residual; then proj_out and norm_out.
"""
import torch
import torch.nn as nn

from . import packing


class PerceiverAttention(nn.Module):
    def __init__(self, *, dim, dim_head=64, heads=8):
        super().__init__()
        self.dim_head, self.heads = dim_head, heads
        inner = dim_head * heads
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(dim, inner * 2, bias=False)
        self.to_out = nn.Linear(inner, dim, bias=False)


def FeedForward(dim, mult=4):
    inner = int(dim * mult)
    return nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, inner, bias=False), nn.GELU(), nn.Linear(inner, dim, bias=False))


class Resampler(packing.PackedWeights, nn.Module):
    def __init__(self, dim=1024, depth=8, dim_head=64, heads=16, num_queries=8, embedding_dim=768,
                 output_dim=1024, ff_mult=4, video_length=None):
        super().__init__()
        if dim_head != 64:
            raise NotImplementedError("pm_attention is built for head dim 64 (the shipped configs use 64)")
        self.num_queries, self.video_length = num_queries, video_length
        nq = num_queries * video_length if video_length is not None else num_queries
        self.latents = nn.Parameter(torch.randn(1, nq, dim) / dim ** 0.5)
        self.proj_in = nn.Linear(embedding_dim, dim)
        self.proj_out = nn.Linear(dim, output_dim)
        self.norm_out = nn.LayerNorm(output_dim)
        self.layers = nn.ModuleList([nn.ModuleList([PerceiverAttention(dim=dim, dim_head=dim_head, heads=heads),
                                                    FeedForward(dim=dim, mult=ff_mult)]) for _ in range(depth)])
        self.heads, self.ops = heads, None
        self._init_packed()

    def bind(self, ops):
        self.ops = ops
        self.invalidate_packed()
        return self

    def prepare(self):
        ops = self.ops
        dev = ops.device
        wt = lambda t: t.detach().to(device=dev, dtype=ops.dtype).contiguous()
        f32 = lambda t: None if t is None else t.detach().to(device=dev, dtype=torch.float32).contiguous()
        ln = lambda m: (f32(m.weight), f32(m.bias))
        W = {"proj_in": (wt(self.proj_in.weight), f32(self.proj_in.bias)),
             "proj_out": (wt(self.proj_out.weight), f32(self.proj_out.bias)),
             "norm_out": ln(self.norm_out), "latents": f32(self.latents)[0], "layers": []}
        for attn, ff in self.layers:
            W["layers"].append({"n1": ln(attn.norm1), "n2": ln(attn.norm2), "q": wt(attn.to_q.weight),
                                "kv": wt(attn.to_kv.weight), "o": wt(attn.to_out.weight),
                                "fn": ln(ff[0]), "f1": wt(ff[1].weight), "f2": wt(ff[3].weight)})
        self._packed = W
        return self

    @torch.no_grad()
    def forward(self, x):
        """x [B, n1, embedding_dim] image tokens -> [B, num_queries(*video_length), output_dim]."""
        if self.ops is None:
            raise RuntimeError("Resampler has no op table bound (Resampler.bind(HipOps(...)))")
        ops, W, heads = self.ops, self.packed(), self.heads
        B, n1, E = x.shape
        nq, D = W["latents"].shape
        inner = heads * 64
        xs = ops.gemm(x.reshape(B * n1, E).to(device=ops.device, dtype=ops.dtype).contiguous(), *W["proj_in"], stream=True)
        lat = W["latents"].unsqueeze(0).expand(B, nq, D).reshape(B * nq, D).contiguous()  # f32 residual stream
        for L in W["layers"]:
            xn = ops.layernorm(xs, *L["n1"]).view(B, n1, D)
            ln = ops.layernorm(lat, *L["n2"])
            q = ops.gemm(ln, L["q"]).view(B, nq, inner)
            kv_in = torch.cat([xn, ln.view(B, nq, D)], 1).reshape(B * (n1 + nq), D)  # keys: image tokens, then latents
            kv = ops.gemm(kv_in, L["kv"]).view(B, n1 + nq, 2 * inner)
            a = ops.attention(q, kv[..., :inner], kv[..., inner:], heads)
            lat = ops.gemm(a.reshape(B * nq, inner), L["o"], None, residual=lat, stream=True)
            h = ops.gemm(ops.layernorm(lat, *L["fn"]), L["f1"], act="gelu")
            lat = ops.gemm(h, L["f2"], None, residual=lat, stream=True)
        out = ops.gemm(lat, *W["proj_out"], stream=True)
        return ops.layernorm(out, *W["norm_out"]).view(B, nq, -1)
