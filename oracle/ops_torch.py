"""ORACLE (test infrastructure, not product code): per-op CPU restatement of the hot-path operators.

TorchOps mirrors the method table of open-pandora_amd/ops_hip.py (same names, argument meaning and
packed-weight conventions) but evaluates every op with plain PyTorch f32 math in the REFERENCE's own
formulation (NCHW convs, einsum attention with a materialised softmax, nn.functional norms), so a
HIP kernel and its layout/packing conventions are checked against an independent computation.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Reference call sites restated here (relative to /root/reference/DynamiCrafter/lvdm):
  gemm/geglu      modules/attention.py:53-57,86-99,144,415-442
  conv3x3         modules/networks/openaimodel3d.py:51-109,154-183 (Conv2d 3x3 pad 1; nearest x2 in f32)
  conv_t3         modules/networks/openaimodel3d.py:252-269 (Conv3d (3,1,1) pad (1,0,0))
  groupnorm       basics.py:76-88 (native-dtype nn.GroupNorm), openaimodel3d.py:258-269, attention.py:265,331
  layernorm       modules/attention.py:225-227
  attention       modules/attention.py:101-142 (scale 64^-0.5, softmax over keys, text + image branches)
  ddim_update     models/samplers/ddim.py:238-288, models/ddpm3d.py:235-247
"""
import torch
import torch.nn.functional as F_


def _f(t):
    return None if t is None else t.float()


class TorchOps:
    name = "torch-oracle"

    def __init__(self, dtype=torch.float32, device="cpu", model_16bit=False):
        """model_16bit (with a 16-bit dtype): the IDEAL 16-bit-operand machine - exact f32 arithmetic, but
        every tensor the HIP path stores as 16 bit (the MFMA operands: normalised activations, q/k/v, the
        attention probabilities P, attention outputs, GEGLU products) is rounded to `dtype`, and what it keeps
        in f32 (`stream=True` outputs: the residual stream) stays f32.  Its distance from the f32 reference is
        the error inherent to 16-bit matrix operands, independent of any kernel
        (tests/test_error_budget_gpu.py compares the HIP kernels with it)."""
        self.dtype = dtype
        self.device = torch.device(device)
        self.model_16bit = bool(model_16bit) and dtype != torch.float32

    def empty(self, *shape, dtype=None):
        return torch.empty(*shape, dtype=dtype or self.dtype, device=self.device)

    @staticmethod
    def _with_stats(y, stats):
        if stats is None:
            return y
        NI, groups = stats[0], stats[1]  # (a third element, "lazy", is a HipOps hint)
        yg = y.float().reshape(NI, y.shape[0] // NI, groups, -1)
        return y, torch.stack([yg.sum((1, 3)), (yg * yg).sum((1, 3))], -1)

    def _a(self, t):
        """A operand of a GEMM-family op: the MFMA takes it as 16 bit even when it is read from the f32 residual
        stream (1x1 skip convs, Down/Upsample convs: PM_FLAG_A_F32 rounds while staging)."""
        return t.to(self.dtype).float() if self.model_16bit else t.float()

    def _out(self, t, out, stream=False):
        t = t.to(torch.float32 if (stream and self.model_16bit) else self.dtype)
        if out is not None:
            out.copy_(t)
            return out
        return t

    # -- GEMM family -----------------------------------------------------------------------------
    def gemm(self, a, w, bias=None, residual=None, act="none", out=None, stream=False, stats=None, col_scale=None,
             split_a=False):
        if split_a and self.model_16bit and a.dtype == torch.float32:  # hi + lo: two 16-bit operands carry a
            hi = a.to(self.dtype).float()
            af = hi + (a - hi).to(self.dtype).float()
        else:
            af = self._a(a)
        y = af @ _f(w).t()
        if col_scale is not None:
            y = y * _f(col_scale)
        if bias is not None:
            y = y + _f(bias)
        if act == "silu":
            y = F_.silu(y)
        elif act == "gelu":
            y = F_.gelu(y)
        elif act == "geglu":
            # packed rows: per 32-row group [16 value rows | 16 gate rows]
            n = y.shape[1]
            yy = y.reshape(y.shape[0], n // 32, 2, 16)
            y = (yy[:, :, 0] * F_.gelu(yy[:, :, 1])).reshape(y.shape[0], n // 2)
        if residual is not None:
            y = y + _f(residual)
        return self._with_stats(self._out(y, out, stream), stats)

    def conv3x3(self, x, wp, bias, F, H, W, stride=1, upsample=False, residual=None, out=None, stream=False,
                pad_lo=1, stats=None, presplit_upsample=True):
        cin, cout = x.shape[1], wp.shape[0]
        xi = self._a(x).reshape(F, H, W, cin).permute(0, 3, 1, 2)
        if upsample:
            xi = F_.interpolate(xi, scale_factor=2, mode="nearest")
        w = _f(wp).reshape(cout, 3, 3, cin).permute(0, 3, 1, 2)  # -> [Cout, Cin, ky, kx]
        xi = F_.pad(xi, (pad_lo, 1, pad_lo, 1))  # (left, right, top, bottom)
        y = F_.conv2d(xi, w, _f(bias), stride=stride, padding=0)
        y = y.permute(0, 2, 3, 1).reshape(-1, cout)
        if residual is not None:
            y = y + _f(residual)
        return self._with_stats(self._out(y, out, stream), stats)

    def conv_t3(self, x, wp, bias, F, P, residual=None, halo_lo=None, halo_hi=None, out=None, stream=False,
                stats=None):
        cin, cout = x.shape[1], wp.shape[0]
        xi = self._a(x).reshape(F, P, cin)
        lo = torch.zeros(1, P, cin) if halo_lo is None else _f(halo_lo).reshape(1, P, cin)
        hi = torch.zeros(1, P, cin) if halo_hi is None else _f(halo_hi).reshape(1, P, cin)
        xe = torch.cat([lo, xi, hi], 0)  # frames -1 .. F
        # Conv3d with kernel (3,1,1): [1, Cin, F+2, P, 1]
        xc = xe.permute(2, 0, 1)[None, :, :, :, None]
        w = _f(wp).reshape(cout, 3, cin).permute(0, 2, 1)[:, :, :, None, None]
        y = F_.conv3d(xc, w, _f(bias))  # valid conv over the extended frame axis
        y = y[0, :, :, :, 0].permute(1, 2, 0).reshape(F * P, cout)
        if residual is not None:
            y = y + _f(residual)
        return self._with_stats(self._out(y, out, stream), stats)

    def gemv(self, w, x, bias=None, silu_in=False, act="none"):
        xv = _f(x)
        if silu_in:
            xv = F_.silu(xv)
        y = _f(w) @ xv
        if bias is not None:
            y = y + _f(bias)
        if act == "silu":
            y = F_.silu(y)
        return y

    # -- normalisation ---------------------------------------------------------------------------
    def groupnorm(self, x, gamma, beta, eps, NI, silu, groups=32, stats_reduce=None, out=None, totals=None):
        M, C = x.shape
        P = M // NI
        xi = _f(x).reshape(NI, P, C).permute(0, 2, 1)  # [NI, C, P]
        if totals is not None and stats_reduce is None:  # statistics handed over by the producing op
            cnt = P * (C // groups)
            mean = totals[..., 0] / cnt
            var = (totals[..., 1] / cnt - mean * mean).clamp_min(0)
            xg = xi.reshape(NI, groups, -1)
            y = ((xg - mean[..., None]) * torch.rsqrt(var + eps)[..., None]).reshape(NI, C, P)
            y = y * _f(gamma)[None, :, None] + _f(beta)[None, :, None]
        elif stats_reduce is None:
            y = F_.group_norm(xi, groups, _f(gamma), _f(beta), eps)
        else:
            xg = xi.reshape(NI, groups, -1)
            part = totals if totals is not None else torch.stack([xg.sum(-1), (xg * xg).sum(-1)], -1)
            tot, count = stats_reduce(part, P * (C // groups))
            mean = tot[..., 0] / count
            var = (tot[..., 1] / count - mean * mean).clamp_min(0)
            y = (xg - mean[..., None]) * torch.rsqrt(var + eps)[..., None]
            y = y.reshape(NI, C, P) * _f(gamma)[None, :, None] + _f(beta)[None, :, None]
        if silu:
            y = F_.silu(y)
        return self._out(y.permute(0, 2, 1).reshape(M, C), out)

    def layernorm(self, x, gamma, beta, eps=1e-5, out=None):
        return self._out(F_.layer_norm(_f(x), (x.shape[1],), _f(gamma), _f(beta), eps), out)

    def ln_gemm(self, x, gamma, beta, w, bias=None, act="none", col_scale=None, eps=1e-5):
        return self.gemm(self.layernorm(x, gamma, beta, eps), w, bias, act=act, col_scale=col_scale)

    # -- attention -------------------------------------------------------------------------------
    def _attn(self, q, k, v, heads):
        B, Nq, C = q.shape
        if k.shape[0] == 1 and B > 1:
            k, v = k.expand(B, -1, -1), v.expand(B, -1, -1)
        sp = lambda t: t.reshape(t.shape[0], t.shape[1], heads, 64).permute(0, 2, 1, 3)
        qh, kh, vh = sp(q), sp(k), sp(v)
        out = torch.empty_like(qh)
        for h in range(heads):  # per head to bound the score tensor
            sim = torch.einsum("bid,bjd->bij", qh[:, h], kh[:, h]) * (64 ** -0.5)
            if self.model_16bit:  # P (unnormalised, row maximum 1) is an MFMA operand: rounded; the row sum is f32
                pu = torch.exp(sim - sim.amax(-1, keepdim=True))
                out[:, h] = torch.einsum("bij,bjd->bid", pu.to(self.dtype).float(), vh[:, h]) / pu.sum(-1, keepdim=True)
                continue
            out[:, h] = torch.einsum("bij,bjd->bid", sim.softmax(dim=-1), vh[:, h])
        return out.permute(0, 2, 1, 3).reshape(B, Nq, C)

    def attention(self, q, k1, v1, heads, k2=None, v2=None, w2=1.0, out=None, prescaled=False):
        assert not prescaled, "TorchOps has no q_prescale: callers fold nothing"
        y = self._attn(_f(q), _f(k1), _f(v1), heads)
        if k2 is not None:
            y = y + w2 * self._attn(_f(q), _f(k2), _f(v2), heads)
        return self._out(y, out)

    def attention_generic(self, q, k, v, heads, out=None):
        B, Nq, C = q.shape
        D = C // heads
        sp = lambda t: _f(t).reshape(t.shape[0], t.shape[1], heads, D).permute(0, 2, 1, 3)
        w = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) * D ** -0.5, -1)
        return self._out((w @ sp(v)).permute(0, 2, 1, 3).reshape(B, Nq, C), out)

    def attention_temporal(self, q, k, v, heads, out=None):
        # (Fq, P, C) -> batch over pixels: (P, Fq, C)
        y = self._attn(_f(q).permute(1, 0, 2), _f(k).permute(1, 0, 2), _f(v).permute(1, 0, 2), heads)
        return self._out(y.permute(1, 0, 2), out)

    # -- path boundary ---------------------------------------------------------------------------
    def ddim_update(self, x, e_c, e_u, noise, cfg, sqrt_ac, sqrt_1mac, rescale, sqrt_a_prev,
                    dir_coef, sigma, want_x0=True):
        f32 = lambda s: torch.tensor(s, dtype=torch.float32)
        ec = _f(e_c).reshape(x.shape)
        v = ec if e_u is None else _f(e_u).reshape(x.shape) + f32(cfg) * (ec - _f(e_u).reshape(x.shape))
        eps = f32(sqrt_ac) * v + f32(sqrt_1mac) * x
        x0 = (f32(sqrt_ac) * x - f32(sqrt_1mac) * v) * f32(rescale)
        xp = f32(sqrt_a_prev) * x0 + f32(dir_coef) * eps
        if noise is not None:
            xp = xp + f32(sigma) * noise
        return xp, (x0 if want_x0 else None)

    def pack_input(self, x, cond):
        xs = x if cond is None else torch.cat([x, cond], 0)  # [C, F, P]
        C, F, P = xs.shape
        return xs.permute(1, 2, 0).reshape(F * P, C).to(self.dtype)

    def latent_affine(self, x, W, b, inv_scale, cpad=8):
        C, F, P = x.shape
        y = torch.einsum("oc,cfp->fpo", _f(W), _f(x) * inv_scale) + _f(b)
        out = torch.zeros(F * P, cpad)
        out[:, :C] = y.reshape(F * P, C)
        return out.to(self.dtype)

    def softmax_rows(self, x, scale):
        return torch.softmax(_f(x) * scale, dim=-1).to(self.dtype)

    def unpack_output(self, y, F, P):
        return y.reshape(F, P, -1).permute(2, 0, 1).contiguous()
