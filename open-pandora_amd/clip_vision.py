"""OpenCLIP ViT-H/14 image tower of the conditioning tail: drop-in for
lvdm.modules.encoders.condition.FrozenOpenCLIPImageEmbedderV2 (condition.py:300-382) on the MI355X op table.

The reference builds `open_clip.create_model_and_transforms("ViT-H-14", pretrained="laion2b_s32b_b79k")`
(third-party `open_clip_torch`, pinned 2.22.0 in DynamiCrafter/requirements.txt:22, absent from this image),
deletes the text transformer and returns ALL 257 tokens of the vision transformer before `ln_post` / `proj`
(`encode_with_vision_transformer`, :350-382).  This module restates that published architecture -
`open_clip.transformer.VisionTransformer`: 14x14 patch conv without bias, class + positional embedding, ln_pre,
32 pre-LN residual blocks of width 1280 (nn.MultiheadAttention with 16 heads of 80 channels; MLP 5120, erf GELU) -
with the same parameter names under `model.visual.*`, so the `embedder.*` tensors of a DynamiCrafter checkpoint load
(`load_state_dict(strict=False)`: the CLIP text-side leftovers `model.token_embedding` ... are not used and not kept).

Graph: image -> resize 224 + CLIP normalisation (torch, as the reference's torchvision transforms) -> patches as a
GEMM on the unfolded image (K = 3*14*14 = 588 zero-padded to 640) with the positional embedding as its residual ->
ln_pre (pm_layernorm) -> per block: LayerNorm, fused in_proj GEMM (+bias), pm_attention_generic (head dim 80),
out_proj GEMM + residual, LayerNorm, c_fc GEMM with the erf-GELU epilogue, c_proj GEMM + residual; f32 residual stream.
"""
import collections

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import packing

VIT_H_14 = dict(image_size=224, patch_size=14, width=1280, layers=32, heads=16, mlp_ratio=4.0, output_dim=1024)


class ResidualAttentionBlock(nn.Module):
    def __init__(self, d_model, n_head, mlp_ratio):
        super().__init__()
        self.ln_1 = nn.LayerNorm(d_model)
        self.attn = nn.MultiheadAttention(d_model, n_head)  # parameter container: in_proj_weight/bias, out_proj
        self.ln_2 = nn.LayerNorm(d_model)
        mlp_width = int(d_model * mlp_ratio)
        self.mlp = nn.Sequential(collections.OrderedDict([("c_fc", nn.Linear(d_model, mlp_width)), ("gelu", nn.GELU()),
                                                          ("c_proj", nn.Linear(mlp_width, d_model))]))


class _Transformer(nn.Module):
    def __init__(self, width, layers, heads, mlp_ratio):
        super().__init__()
        self.resblocks = nn.ModuleList([ResidualAttentionBlock(width, heads, mlp_ratio) for _ in range(layers)])


class VisionTransformer(nn.Module):
    def __init__(self, image_size, patch_size, width, layers, heads, mlp_ratio, output_dim):
        super().__init__()
        self.grid_size = (image_size // patch_size, image_size // patch_size)
        self.patch_size = (patch_size, patch_size)
        self.image_size, self.width, self.heads = image_size, width, heads
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        scale = width ** -0.5
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn(self.grid_size[0] * self.grid_size[1] + 1, width))
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = _Transformer(width, layers, heads, mlp_ratio)
        self.ln_post = nn.LayerNorm(width)                               # (not on this path: kept for the key set)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))  # (idem)


class _ClipShell(nn.Module):
    def __init__(self, **cfg):
        super().__init__()
        self.visual = VisionTransformer(**cfg)


class FrozenOpenCLIPImageEmbedderV2(packing.PackedWeights, nn.Module):
    def __init__(self, arch="ViT-H-14", version="laion2b_s32b_b79k", device="cuda", freeze=True, layer="pooled",
                 antialias=True, vision_cfg=None):
        super().__init__()
        if arch != "ViT-H-14" and vision_cfg is None:
            raise NotImplementedError(f"arch {arch!r}: the shipped configs use ViT-H-14 (pass vision_cfg for others)")
        if layer == "penultimate":
            raise NotImplementedError()  # as the reference (condition.py:317-319)
        self.model = _ClipShell(**dict(vision_cfg or VIT_H_14))
        self.antialias = antialias
        self.mean, self.std = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)
        self.ops = None
        self._init_packed()
        if freeze:
            self.eval()
            for p in self.parameters():
                p.requires_grad = False

    def bind(self, ops):
        self.ops = ops
        self.invalidate_packed()
        return self

    def preprocess(self, x):
        """transforms.Resize((224, 224)) + Normalize(mean, std) (condition.py:331-343): bilinear, antialiased."""
        s = self.model.visual.image_size
        x = F.interpolate(x.float(), size=(s, s), mode="bilinear", antialias=self.antialias, align_corners=False)
        mean = torch.tensor(self.mean, device=x.device)[None, :, None, None]
        return (x - mean) / torch.tensor(self.std, device=x.device)[None, :, None, None]

    def prepare(self):
        ops, vis = self.ops, self.model.visual
        dev = ops.device
        wt = lambda t: t.detach().to(device=dev, dtype=ops.dtype).contiguous()
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        k = vis.conv1.weight.shape[1] * vis.patch_size[0] * vis.patch_size[1]
        kpad = (k + 63) // 64 * 64
        wp = torch.zeros(vis.width, kpad)
        wp[:, :k] = vis.conv1.weight.detach().float().reshape(vis.width, k)
        W = {"patch": wt(wp), "kpad": kpad, "k": k, "pos_patches": f32(vis.positional_embedding[1:]),
             "cls": f32(vis.class_embedding + vis.positional_embedding[0]), "ln_pre": (f32(vis.ln_pre.weight), f32(vis.ln_pre.bias)),
             "blocks": []}
        for blk in vis.transformer.resblocks:
            W["blocks"].append({"ln1": (f32(blk.ln_1.weight), f32(blk.ln_1.bias)),
                                "qkv": (wt(blk.attn.in_proj_weight), f32(blk.attn.in_proj_bias)),
                                "out": (wt(blk.attn.out_proj.weight), f32(blk.attn.out_proj.bias)),
                                "ln2": (f32(blk.ln_2.weight), f32(blk.ln_2.bias)),
                                "fc": (wt(blk.mlp.c_fc.weight), f32(blk.mlp.c_fc.bias)),
                                "proj": (wt(blk.mlp.c_proj.weight), f32(blk.mlp.c_proj.bias))})
        self._packed = W
        return self

    @torch.no_grad()
    def forward(self, image, no_dropout=False):
        """image (b, 3, H, W) -> (b, 257, 1280) tokens (all of them, before ln_post / proj)."""
        if self.ops is None:
            raise RuntimeError("FrozenOpenCLIPImageEmbedderV2.bind(ops) must be called first (no implicit CPU fallback)")
        ops, W, vis = self.ops, self.packed(), self.model.visual
        B = image.shape[0]
        g, ps, C, heads = vis.grid_size, vis.patch_size[0], vis.width, vis.heads
        x = self.preprocess(image.to(ops.device))
        # unfold: (b, 3, g*ps, g*ps) -> (b*g*g, 3*ps*ps), channel-major inside a patch like conv1's weight
        pt = x.reshape(B, 3, g[0], ps, g[1], ps).permute(0, 2, 4, 1, 3, 5).reshape(B * g[0] * g[1], W["k"])
        a = torch.zeros(pt.shape[0], W["kpad"], dtype=ops.dtype, device=ops.device)
        a[:, :W["k"]] = pt.to(ops.dtype)
        pos = W["pos_patches"].repeat(B, 1)
        patches = ops.gemm(a, W["patch"], None, residual=pos, stream=True).view(B, g[0] * g[1], C)
        n = g[0] * g[1] + 1
        h = torch.cat([W["cls"].expand(B, 1, C), patches], 1).reshape(B * n, C).contiguous()
        h = ops.layernorm(h, *W["ln_pre"]).float()  # (the reference's stream starts at ln_pre's output)
        for L in W["blocks"]:
            qkv = ops.gemm(ops.layernorm(h, *L["ln1"]), *L["qkv"]).view(B, n, 3 * C)
            att = ops.attention_generic(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads)
            h = ops.gemm(att.view(B * n, C), *L["out"], residual=h, stream=True)
            y = ops.gemm(ops.layernorm(h, *L["ln2"]), *L["fc"], act="gelu")
            h = ops.gemm(y, *L["proj"], residual=h, stream=True)
        return h.view(B, n, C)

    def encode(self, image):
        return self(image)
