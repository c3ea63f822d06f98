"""ORACLE (test infrastructure, not product code): CPU f32 restatement of the OpenCLIP ViT image tower as
FrozenOpenCLIPImageEmbedderV2 runs it (condition.py:300-382), functional over a state_dict with open_clip's key
names under `model.visual.`.

PARITY of this module: the arithmetic lives in the third-party `open_clip_torch` (pinned 2.22.0 in
DynamiCrafter/requirements.txt:22; absent from this image and from /root/reference): there is no open_clip output to pin
against - "parity unpinned" against THAT package.  Since r05 the restatement is pinned against a third-party
implementation of the same architecture that IS in the image: HF `transformers.CLIPVisionModel.last_hidden_state` (the
class that loads the laion/CLIP-ViT-H-14-laion2B-s32B-b79K conversion of the checkpoint the reference pulls through
open_clip) on seeded weights in open_clip's key layout, reduced and full ViT-H/14 size, 2e-5
(tests/golden/clip_vision_hf.npz, oracle/make_golden.py --clip-hf, tests/test_clip_vision_cpu.py).  Restated from the
published architecture - open_clip/transformer.py `VisionTransformer`
(conv1 patchify without bias, class token + learned positions, ln_pre, pre-LN `ResidualAttentionBlock`s built on
nn.MultiheadAttention and Linear-GELU-Linear) - and anchored on the reference's own call site: all tokens of the
transformer output, no ln_post, no projection (condition.py:353-382); preprocessing = torchvision Resize((224, 224))
(bilinear, antialias) + Normalize(mean, std) (:331-343).
"""
import torch
import torch.nn.functional as F


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"].float(), sd[p + ".bias"].float(), 1e-5)


def preprocess(x, image_size=224):
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073])[None, :, None, None]
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711])[None, :, None, None]
    x = F.interpolate(x.float(), size=(image_size, image_size), mode="bilinear", antialias=True, align_corners=False)
    return (x - mean) / std


def vision_tower_forward(sd, image, heads, image_size=224):
    """image (b, 3, H, W) -> (b, grid^2 + 1, width): the embedder's output."""
    v = "model.visual."
    x = preprocess(image, image_size)
    w = sd[v + "conv1.weight"].float()
    x = F.conv2d(x, w, None, stride=w.shape[-1])                        # (b, width, g, g)
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)          # (b, g*g, width)
    cls = sd[v + "class_embedding"].float() + torch.zeros(x.shape[0], 1, x.shape[-1])
    x = torch.cat([cls, x], 1) + sd[v + "positional_embedding"].float()
    x = _ln(sd, v + "ln_pre", x)
    depth = 1 + max(int(k.split(".")[4]) for k in sd if k.startswith(v + "transformer.resblocks."))
    b, n, c = x.shape
    d = c // heads
    for i in range(depth):
        p = f"{v}transformer.resblocks.{i}."
        y = _ln(sd, p + "ln_1", x)
        qkv = y @ sd[p + "attn.in_proj_weight"].float().t() + sd[p + "attn.in_proj_bias"].float()
        q, k, val = (t.reshape(b, n, heads, d).transpose(1, 2) for t in qkv.chunk(3, -1))
        att = torch.softmax((q * d ** -0.5) @ k.transpose(-1, -2), -1) @ val
        att = att.transpose(1, 2).reshape(b, n, c)
        x = x + att @ sd[p + "attn.out_proj.weight"].float().t() + sd[p + "attn.out_proj.bias"].float()
        y = _ln(sd, p + "ln_2", x)
        y = F.gelu(y @ sd[p + "mlp.c_fc.weight"].float().t() + sd[p + "mlp.c_fc.bias"].float())
        x = x + y @ sd[p + "mlp.c_proj.weight"].float().t() + sd[p + "mlp.c_proj.bias"].float()
    return x
