"""What would the scale operands of v_mfma_scale_f32_32x32x64_f8f6f4 buy pm_attention_fp8?  (VERDICT r02 #7.)

Exact-arithmetic emulation on the CPU (products and sums in f64, only the four e4m3 roundings of the kernel applied:
q' = q * scale * log2 e, k, P = 2^(s - m), v), with and without MX block scales: one power-of-two scale per 32 elements
along the contraction axis (per (row, 32-d block) of Q' and K for Q'K^T; per (key block of 32, d) of V and per (query,
32-key block) of P for PV), chosen so that the block's absmax lands just below e4m3's 448.

    python tools/fp8_scale_study.py
"""
import math

import torch

F8 = torch.float8_e4m3fn


def q8(x):
    return x.float().clamp(-448, 448).to(F8).double()


def q8_blocks(x, axis, blk=32):
    """e4m3 with one power-of-two scale per `blk` elements along `axis` (MX style), returned de-scaled."""
    x = x.double().movedim(axis, -1)
    shp = x.shape
    xb = x.reshape(*shp[:-1], shp[-1] // blk, blk)
    amax = xb.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    e = torch.floor(torch.log2(448.0 / amax))
    y = q8(xb * 2.0 ** e) / 2.0 ** e
    return y.reshape(shp).movedim(-1, axis)


def attention(q, k, v, mode):
    d = q.shape[-1]
    qs = q.double() * (d ** -0.5) * math.log2(math.e)
    if mode == "f64":
        s = qs @ k.double().T
        p = torch.exp2(s - s.amax(-1, keepdim=True))
        return (p @ v.double()) / p.sum(-1, keepdim=True)
    blk = mode == "block"
    qq = q8_blocks(qs, -1) if blk else q8(qs)
    kk = q8_blocks(k, -1) if blk else q8(k)
    s = qq @ kk.T
    m = s.amax(-1, keepdim=True) - 4.0  # the kernel's stale maximum keeps p <= 2^4
    p = torch.exp2(s - m)
    l = p.sum(-1, keepdim=True)  # (row sums are taken in f32 before the rounding, as in the kernel)
    pp = q8_blocks(p, -1) if blk else q8(p)
    vv = q8_blocks(v, 0) if blk else q8(v)
    return (pp @ vv) / l


def rel(a, b):
    return ((a - b).norm() / b.norm()).item()


def main():
    torch.manual_seed(0)
    N, d = 2048, 64
    print(f"{'data':44s} {'unit scales':>12s} {'block scales':>13s}")
    for tag, gen in (("q, k, v ~ N(0, 1)  (the parity tests' data)", lambda: (torch.randn(N, d), torch.randn(N, d), torch.randn(N, d))),
                     ("k, v x 8 (large operands)", lambda: (torch.randn(N, d), 8 * torch.randn(N, d), 8 * torch.randn(N, d))),
                     ("q, k, v x 1/32 (subnormal range of e4m3)", lambda: (torch.randn(N, d), torch.randn(N, d) / 32, torch.randn(N, d) / 32)),
                     ("per-channel spread: k, v x 2^U(-6, 2) per d", lambda: (torch.randn(N, d), torch.randn(N, d) * 2 ** (8 * torch.rand(d) - 6),
                                                                          torch.randn(N, d) * 2 ** (8 * torch.rand(d) - 6)))):
        q, k, v = gen()
        ref = attention(q, k, v, "f64")
        print(f"{tag:44s} {rel(attention(q, k, v, 'unit'), ref):12.2e} {rel(attention(q, k, v, 'block'), ref):13.2e}")


if __name__ == "__main__":
    main()
