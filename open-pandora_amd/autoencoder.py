"""First-stage decoder on the MI355X kernels: `AutoencoderKL` with the reference's
constructor (`ddconfig`, `embed_dim`) and state_dict keys (`decoder.*`, `post_quant_conv.*`,
`encoder.*`, `quant_conv.*`; lvdm/models/autoencoder.py:13-45, ae_modules.py:466-537), and
`decode_first_stage(z)` with the semantics of LatentDiffusion.decode_first_stage (ddpm3d.py:630-655).

The decoder is run channels-last on the same op table as the U-Net: 3x3 convs (and nearest-x2
upsample + conv) on pm_conv2d_3x3, GroupNorm(eps 1e-6)+swish on pm_groupnorm_*, 1x1 convs on pm_gemm,
the f32 residual stream convention of unet.py.  The single-head 512-channel mid attention
(ae_modules.py:52-75, head dim 512 - not the 64 of pm_attention) is QK^T and PV on pm_gemm around
pm_softmax_rows, one frame at a time; V's bias is added after PV (softmax rows sum to one).
`encode_first_stage(x)` (1-4 conditioning frames per generate, model.py:690-701) runs the encoder the
same way: its stride-2 Downsample uses the (0,1,0,1) zero padding through pm_conv2d_3x3's pad_lo = 0;
quant_conv (1x1 after conv_out) is folded into conv_out's weights at pack time (exact: both linear,
no padding between them); the posterior sample mean + std * noise is taken on the (n, 4, h, w) latent.
"""
import torch
import torch.nn as nn

from . import packing


class _ResnetBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(32, cout, eps=1e-6)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        if cin != cout:
            self.nin_shortcut = nn.Conv2d(cin, cout, 1)


class _AttnBlock(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.norm = nn.GroupNorm(32, c, eps=1e-6)
        self.q, self.k, self.v = nn.Conv2d(c, c, 1), nn.Conv2d(c, c, 1), nn.Conv2d(c, c, 1)
        self.proj_out = nn.Conv2d(c, c, 1)


class _Resample(nn.Module):
    def __init__(self, c, down):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2 if down else 1, padding=0 if down else 1)


class _Level(nn.Module):
    pass


def _mid(c):
    m = nn.Module()
    m.block_1, m.attn_1, m.block_2 = _ResnetBlock(c, c), _AttnBlock(c), _ResnetBlock(c, c)
    return m


class Decoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, z_channels, **ignored):
        super().__init__()
        nres = len(ch_mult)
        block_in = ch * ch_mult[-1]
        self.conv_in = nn.Conv2d(z_channels, block_in, 3, padding=1)
        self.mid = _mid(block_in)
        self.up = nn.ModuleList()
        for lvl in reversed(range(nres)):
            up = _Level()
            up.block, up.attn = nn.ModuleList(), nn.ModuleList()
            block_out = ch * ch_mult[lvl]
            for _ in range(num_res_blocks + 1):
                up.block.append(_ResnetBlock(block_in, block_out))
                block_in = block_out
            if lvl != 0:
                up.upsample = _Resample(block_in, down=False)
            self.up.insert(0, up)
        self.norm_out = nn.GroupNorm(32, block_in, eps=1e-6)
        self.conv_out = nn.Conv2d(block_in, out_ch, 3, padding=1)


class Encoder(nn.Module):
    """Parameter container only (checkpoint keys); see the module docstring."""

    def __init__(self, *, ch, in_channels, ch_mult=(1, 2, 4, 8), num_res_blocks, z_channels, double_z=True, **ignored):
        super().__init__()
        self.conv_in = nn.Conv2d(in_channels, ch, 3, padding=1)
        in_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        block_in = ch
        for lvl in range(len(ch_mult)):
            d = _Level()
            d.block, d.attn = nn.ModuleList(), nn.ModuleList()
            block_in, block_out = ch * in_mult[lvl], ch * ch_mult[lvl]
            for _ in range(num_res_blocks):
                d.block.append(_ResnetBlock(block_in, block_out))
                block_in = block_out
            if lvl != len(ch_mult) - 1:
                d.downsample = _Resample(block_in, down=True)
            self.down.append(d)
        self.mid = _mid(block_in)
        self.norm_out = nn.GroupNorm(32, block_in, eps=1e-6)
        self.conv_out = nn.Conv2d(block_in, 2 * z_channels if double_z else z_channels, 3, padding=1)


DDCONFIG = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
                ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0)  # inference yaml :57-76


class AutoencoderKL(packing.PackedWeights, nn.Module):
    def __init__(self, ddconfig=None, lossconfig=None, embed_dim=4, scale_factor=0.18215, **ignored):
        super().__init__()
        dd = dict(ddconfig or DDCONFIG)
        assert not dd.get("attn_resolutions"), "only the mid attention of the shipped config is built"
        self.encoder = Encoder(**dd)
        self.decoder = Decoder(**dd)
        self.quant_conv = nn.Conv2d(2 * dd["z_channels"], 2 * embed_dim, 1)
        self.post_quant_conv = nn.Conv2d(embed_dim, dd["z_channels"], 1)
        self.embed_dim, self.scale_factor = embed_dim, scale_factor
        self.ops = None
        self._init_packed()

    def bind(self, ops):
        self.ops = ops
        self.invalidate_packed()
        return self

    @torch.no_grad()
    def encode_moments(self, x):
        """x (n, 3, H, W) pixels in [-1, 1] -> posterior moments (n, 8, H/8, W/8) f32 [mean | logvar]."""
        if self.ops is None:
            raise RuntimeError("AutoencoderKL.bind(ops) must be called first (no implicit CPU fallback)")
        ops, W = self.ops, self.packed()
        n, c, H, Wd = x.shape
        xs = x.to(device=ops.device, dtype=torch.float32).permute(1, 0, 2, 3).reshape(c, n, H * Wd).contiguous()
        pad = torch.zeros(8 - c, n, H * Wd, dtype=torch.float32, device=ops.device)
        h = ops.pack_input(xs, pad)                                   # [n*H*W, 8] (3 channels + zero pad)
        h = ops.conv3x3(h, *W["enc.conv_in"], n, H, Wd, stream=True)
        enc = self.encoder
        for lvl in range(len(enc.down)):
            for i in range(len(enc.down[lvl].block)):
                h = self._res(W[f"enc.down.{lvl}.block.{i}"], h, n, H, Wd)
            if hasattr(enc.down[lvl], "downsample"):
                h = ops.conv3x3(h, *W[f"enc.down.{lvl}.downsample"], n, H, Wd, stride=2, pad_lo=0, stream=True)
                H, Wd = H // 2, Wd // 2
        h = self._res(W["enc.mid.block_1"], h, n, H, Wd)
        h = self._attn(W["enc.mid.attn_1"], h, n, H * Wd)
        h = self._res(W["enc.mid.block_2"], h, n, H, Wd)
        h = ops.groupnorm(h, *W["enc.norm_out"], 1e-6, n, True)
        m = ops.conv3x3(h, *W["enc.conv_out_q"], n, H, Wd, stream=True)  # conv_out with quant_conv folded in
        return ops.unpack_output(m, n, H * Wd).reshape(m.shape[1], n, H, Wd).permute(1, 0, 2, 3)

    def encode(self, x, **kwargs):
        return self.encode_moments(x)

    @torch.no_grad()
    def encode_first_stage(self, x, noise=None):
        """LatentDiffusion.encode_first_stage + get_first_stage_encoding (ddpm3d.py:596-628):
        x (b, 3, t, H, W) or (n, 3, H, W) -> scale_factor * posterior sample."""
        five = x.dim() == 5
        if five:
            b, c, t, H, Wd = x.shape
            x = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, H, Wd)
        mom = self.encode_moments(x)
        mean, logvar = torch.chunk(mom, 2, dim=1)
        noise = torch.randn_like(mean) if noise is None else noise.to(mean)
        z = self.scale_factor * (mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * noise)
        if five:
            z = z.reshape(b, t, *z.shape[1:]).permute(0, 2, 1, 3, 4)
        return z

    # ---- kernel-side weights --------------------------------------------------------------------
    def prepare(self):
        ops = self.ops
        dev, dt = ops.device, ops.dtype
        wt = lambda t: t.detach().to(device=dev, dtype=dt).contiguous()
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        conv = lambda m: (wt(packing.pack_conv3x3(m.weight)), f32(m.bias))
        lin = lambda m: (wt(m.weight.reshape(m.weight.shape[0], -1)), f32(m.bias))
        gn = lambda m: (f32(m.weight), f32(m.bias))
        W = {}
        for name, m in self.decoder.named_modules():
            if isinstance(m, _ResnetBlock):
                W[name] = dict(n1=gn(m.norm1), c1=conv(m.conv1), n2=gn(m.norm2), c2=conv(m.conv2),
                               skip=lin(m.nin_shortcut) if hasattr(m, "nin_shortcut") else None)
            elif isinstance(m, _AttnBlock):
                wv, bv = lin(m.v)
                W[name] = dict(n=gn(m.norm), q=lin(m.q), k=lin(m.k), v=(wv, bv), o=lin(m.proj_out))
            elif isinstance(m, _Resample):
                W[name] = conv(m.conv)
        for name, m in self.encoder.named_modules():
            if isinstance(m, _ResnetBlock):
                W["enc." + name] = dict(n1=gn(m.norm1), c1=conv(m.conv1), n2=gn(m.norm2), c2=conv(m.conv2),
                                        skip=lin(m.nin_shortcut) if hasattr(m, "nin_shortcut") else None)
            elif isinstance(m, _AttnBlock):
                W["enc." + name] = dict(n=gn(m.norm), q=lin(m.q), k=lin(m.k), v=lin(m.v), o=lin(m.proj_out))
            elif isinstance(m, _Resample):
                W["enc." + name] = conv(m.conv)
        e = self.encoder
        w_in = torch.zeros(e.conv_in.weight.shape[0], 8, 3, 3)
        w_in[:, :e.conv_in.weight.shape[1]] = e.conv_in.weight.detach().float().cpu()
        W["enc.conv_in"] = (wt(packing.pack_conv3x3(w_in)), f32(e.conv_in.bias))
        W["enc.norm_out"] = gn(e.norm_out)
        # quant_conv (1x1) o conv_out (3x3): W' = Wq . Wc, b' = Wq . bc + bq (f64 on the host)
        wq = self.quant_conv.weight.detach().double().cpu().reshape(self.quant_conv.weight.shape[0], -1)
        wc, bc = e.conv_out.weight.detach().double().cpu(), e.conv_out.bias.detach().double().cpu()
        w_f = torch.einsum("oc,cikl->oikl", wq, wc).float()
        b_f = (wq @ bc + self.quant_conv.bias.detach().double().cpu()).float()
        W["enc.conv_out_q"] = (wt(packing.pack_conv3x3(w_f)), f32(b_f))
        d = self.decoder
        # conv_in reads the latent padded from z_channels to 8 channels (zero weights on the padding)
        w_in = torch.zeros(d.conv_in.weight.shape[0], 8, 3, 3)
        w_in[:, :d.conv_in.weight.shape[1]] = d.conv_in.weight.detach().float().cpu()
        W["conv_in"] = (wt(packing.pack_conv3x3(w_in)), f32(d.conv_in.bias))
        W["norm_out"], W["conv_out"] = gn(d.norm_out), conv(d.conv_out)
        W["post_quant"] = (f32(self.post_quant_conv.weight.reshape(self.post_quant_conv.weight.shape[0], -1)),
                           f32(self.post_quant_conv.bias))
        self._packed = W
        return self

    # ---- graph ----------------------------------------------------------------------------------
    def _res(self, W, x, F, H, Wd):
        ops = self.ops
        h = ops.groupnorm(x, *W["n1"], 1e-6, F, True)
        h = ops.conv3x3(h, *W["c1"], F, H, Wd, stream=True)
        h = ops.groupnorm(h, *W["n2"], 1e-6, F, True)
        skip = x if W["skip"] is None else ops.gemm(x, *W["skip"], stream=True)
        return ops.conv3x3(h, *W["c2"], F, H, Wd, residual=skip, stream=True)

    def _attn(self, W, x, F, P):
        ops = self.ops
        C = x.shape[1]
        h = ops.groupnorm(x, *W["n"], 1e-6, F, False)
        q, k = ops.gemm(h, *W["q"]), ops.gemm(h, *W["k"])
        if h.shape[1] != C:  # parity op table: the norm output is [hi | lo]; V^T below takes it as the (contiguous) W operand: hi only
            h = h[:, :C].contiguous()
        out = ops.empty(F * P, C)
        for f in range(F):  # one single-head (h*w x h*w) attention per frame
            sl = slice(f * P, (f + 1) * P)
            s = ops.gemm(q[sl], k[sl], stream=True)                 # scores f32 [P, P]
            pmat = ops.softmax_rows(s, C ** -0.5)
            vt = ops.gemm(W["v"][0], h[sl])                          # V^T without bias [C, P]
            ops.gemm(pmat, vt, W["v"][1], out=out[sl])               # P V + b_v (rows of P sum to 1)
        return ops.gemm(out, *W["o"], residual=x, stream=True)

    @torch.no_grad()
    def decode(self, z, scaled=False, **kwargs):
        """z (n, 4, h, w) latents (already divided by scale_factor unless scaled=True) -> (n, 3, 8h, 8w)."""
        if self.ops is None:
            raise RuntimeError("AutoencoderKL.bind(ops) must be called first (no implicit CPU fallback)")
        ops, W = self.ops, self.packed()
        n, c, hh, ww = z.shape
        # the kernels address one operand with 32-bit byte offsets: keep the largest activation
        # (f32, 128 channels at full pixel resolution) under 2 GiB by decoding a few frames at a time
        per_frame = 64 * hh * ww * self.decoder.conv_out.weight.shape[1] * 4
        chunk = max(1, (1 << 31) // per_frame)
        if n > chunk:
            return torch.cat([self.decode(z[i:i + chunk], scaled) for i in range(0, n, chunk)], 0)
        F, H, Wd = n, hh, ww
        zx = z.to(device=ops.device, dtype=torch.float32).permute(1, 0, 2, 3).reshape(c, n, hh * ww).contiguous()
        x = ops.latent_affine(zx, *W["post_quant"], (1.0 / self.scale_factor) if scaled else 1.0)
        h = ops.conv3x3(x, *W["conv_in"], F, H, Wd, stream=True)
        h = self._res(W["mid.block_1"], h, F, H, Wd)
        h = self._attn(W["mid.attn_1"], h, F, H * Wd)
        h = self._res(W["mid.block_2"], h, F, H, Wd)
        for lvl in reversed(range(len(self.decoder.up))):
            up = self.decoder.up[lvl]
            for i in range(len(up.block)):
                h = self._res(W[f"up.{lvl}.block.{i}"], h, F, H, Wd)
            if hasattr(up, "upsample"):
                h = ops.conv3x3(h, *W[f"up.{lvl}.upsample"], F, H, Wd, upsample=True, stream=True)
                H, Wd = 2 * H, 2 * Wd
        h = ops.groupnorm(h, *W["norm_out"], 1e-6, F, True)
        y = ops.conv3x3(h, *W["conv_out"], F, H, Wd, stream=True)           # [F*H*W, 3] f32
        return ops.unpack_output(y, F, H * Wd).reshape(3, n, H, Wd).permute(1, 0, 2, 3)

    @torch.no_grad()
    def decode_first_stage(self, z):
        """LatentDiffusion.decode_first_stage (ddpm3d.py:630-655): z (b, 4, t, h, w) *scaled* latents
        -> (b, 3, t, 8h, 8w)."""
        b, c, t, hh, ww = z.shape
        x = z.permute(0, 2, 1, 3, 4).reshape(b * t, c, hh, ww)
        y = self.decode(x, scaled=True)
        return y.reshape(b, t, *y.shape[1:]).permute(0, 2, 1, 3, 4)
