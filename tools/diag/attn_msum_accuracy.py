"""attn_self16_kernel: row sums on the matrix pipe (variant 16, sums the 16-bit P) against row sums on the vector pipe (variant 18,
sums the f32 p) - error of each against an f64 softmax attention on the same 16-bit inputs, and the row-wise scale between the two.
usage: python tools/diag/attn_msum_accuracy.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from open_pandora_amd.ops_hip import HipOps  # noqa: E402

for dt in (torch.bfloat16, torch.float16):
    ops = HipOps(dt, "cuda:0", diag=True)
    torch.manual_seed(1)
    N, heads, F = 9216, 5, 2
    C = heads * 64
    qkv = torch.randn(F, N, 3 * C, device="cuda", dtype=dt)
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    out = {}
    for var in (1, 18, 16):
        ops.lib.pm_debug_attn_variant(var)
        out[var] = ops.attention(q, k, v, heads).double()
    ops.lib.pm_debug_attn_variant(0)
    qd = q.double().view(F, N, heads, 64).permute(0, 2, 1, 3)[:, :, :1024]  # first 1024 query rows of every (frame, head)
    kd = k.double().view(F, N, heads, 64).permute(0, 2, 1, 3)
    vd = v.double().view(F, N, heads, 64).permute(0, 2, 1, 3)
    ref = torch.softmax(qd @ kd.transpose(-1, -2) / 8.0, -1) @ vd  # [F, heads, 1024, 64]
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    for var in (1, 18, 16):
        o = out[var].view(F, N, heads, 64).permute(0, 2, 1, 3)[:, :, :1024]
        # best row-wise scale of o against ref (a denominator error is a pure row scale)
        sc = (o * ref).sum(-1) / (ref * ref).sum(-1)
        print(f"{dt} variant {var:2d}: rel err vs f64 {rel(o, ref):.3e} (16-bit output rounding included); row scale - 1: mean {(sc - 1).mean().item():+.2e} rms {(sc - 1).pow(2).mean().sqrt().item():.2e}")
