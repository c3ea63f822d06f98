"""ORACLE (test infrastructure, not product code): CPU restatement of the first-stage decoder
(AutoencoderKL.decode + LatentDiffusion.decode_first_stage), functional over a state_dict with the
reference's key names.  Pinned against the real AutoencoderKL (imported from /root/reference by
oracle/make_golden.py) through tests/golden/ae_decode_small.npz.

Reference call sites (relative to /root/reference/DynamiCrafter/lvdm):
  decode_first_stage   models/ddpm3d.py:630-655 (1/scale_factor, per-frame loop = plain batch)
  AutoencoderKL.decode models/autoencoder.py:103-106 (post_quant_conv, decoder)
  Decoder.forward      modules/networks/ae_modules.py:539-578
  ResnetBlock.forward  modules/networks/ae_modules.py:194-215 (temb is None in the decoder)
  AttnBlock.forward    modules/networks/ae_modules.py:52-75 (single head over h*w, scale c^-0.5)
  Upsample.forward     modules/networks/ae_modules.py:119-123 (nearest x2 then conv)
  Encoder.forward      modules/networks/ae_modules.py:430-464; Downsample :99-103 (pad (0,1,0,1), stride 2)
"""
import torch
import torch.nn.functional as F


def _norm(sd, p, x):
    return F.group_norm(x, 32, sd[p + ".weight"].float(), sd[p + ".bias"].float(), 1e-6)


def _conv(sd, p, x, padding=1):
    return F.conv2d(x, sd[p + ".weight"].float(), sd[p + ".bias"].float(), padding=padding)


def _swish(x):
    return x * torch.sigmoid(x)


def resnet_block(sd, p, x):
    h = _conv(sd, p + ".conv1", _swish(_norm(sd, p + ".norm1", x)))
    h = _conv(sd, p + ".conv2", _swish(_norm(sd, p + ".norm2", h)))
    if (p + ".nin_shortcut.weight") in sd:
        x = _conv(sd, p + ".nin_shortcut", x, padding=0)
    return x + h


def attn_block(sd, p, x):
    h = _norm(sd, p + ".norm", x)
    q, k, v = (_conv(sd, f"{p}.{n}", h, padding=0) for n in ("q", "k", "v"))
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, hh * ww)
    w_ = torch.softmax(torch.bmm(q, k) * (int(c) ** -0.5), dim=2)
    v = v.reshape(b, c, hh * ww)
    h = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(sd, p + ".proj_out", h, padding=0)


@torch.no_grad()
def ae_encode_moments(sd, x, prefix=""):
    """x (n, 3, H, W) pixels -> (n, 8, H/8, W/8) posterior moments [mean | logvar]
    (Encoder.forward ae_modules.py:430-464, Downsample :99-103, quant_conv autoencoder.py:98-101)."""
    sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    h = _conv(sd, "encoder.conv_in", x.float())
    levels = sorted({int(k.split(".")[2]) for k in sd if k.startswith("encoder.down.")})
    for lvl in levels:
        i = 0
        while f"encoder.down.{lvl}.block.{i}.norm1.weight" in sd:
            h = resnet_block(sd, f"encoder.down.{lvl}.block.{i}", h)
            i += 1
        if f"encoder.down.{lvl}.downsample.conv.weight" in sd:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)
            h = F.conv2d(h, sd[f"encoder.down.{lvl}.downsample.conv.weight"].float(),
                         sd[f"encoder.down.{lvl}.downsample.conv.bias"].float(), stride=2)
    h = resnet_block(sd, "encoder.mid.block_1", h)
    h = attn_block(sd, "encoder.mid.attn_1", h)
    h = resnet_block(sd, "encoder.mid.block_2", h)
    h = _conv(sd, "encoder.conv_out", _swish(_norm(sd, "encoder.norm_out", h)))
    return _conv(sd, "quant_conv", h, padding=0)


def ae_sample_latent(moments, noise, scale_factor=0.18215):
    """DiagonalGaussianDistribution.sample (distributions.py:24-42) + get_first_stage_encoding
    (ddpm3d.py:596-604): scale_factor * (mean + exp(0.5 clamp(logvar)) * noise)."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    return scale_factor * (mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * noise)


@torch.no_grad()
def ae_decode(sd, z, scale_factor=0.18215, prefix=""):
    """z (b, 4, t, h, w) scaled latents -> (b, 3, t, 8h, 8w) pixels in [-1, 1] (not clamped)."""
    sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    b, c, t, hh, ww = z.shape
    x = (1.0 / scale_factor) * z.permute(0, 2, 1, 3, 4).reshape(b * t, c, hh, ww).float()
    x = _conv(sd, "post_quant_conv", x, padding=0)
    h = _conv(sd, "decoder.conv_in", x)
    h = resnet_block(sd, "decoder.mid.block_1", h)
    h = attn_block(sd, "decoder.mid.attn_1", h)
    h = resnet_block(sd, "decoder.mid.block_2", h)
    levels = sorted({int(k.split(".")[2]) for k in sd if k.startswith("decoder.up.")})
    for lvl in reversed(levels):
        i = 0
        while f"decoder.up.{lvl}.block.{i}.norm1.weight" in sd:
            h = resnet_block(sd, f"decoder.up.{lvl}.block.{i}", h)
            i += 1
        if f"decoder.up.{lvl}.upsample.conv.weight" in sd:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(sd, f"decoder.up.{lvl}.upsample.conv", h)
    h = _conv(sd, "decoder.conv_out", _swish(_norm(sd, "decoder.norm_out", h)))
    return h.reshape(b, t, *h.shape[1:]).permute(0, 2, 1, 3, 4)
