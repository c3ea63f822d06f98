"""A/B of the spatial self-attention kernel variants in ONE process (interleaved rounds, random data).
usage: python tools/attn_bench.py [--dtype bf16|f16] [--rounds 5]
variants: 9 = round-1 kernel (32 rows/wave), 1 = attn_self_kernel 32 rows/wave, 2 = 64 rows/wave,
3 = 64 rows/wave with the P.V of block 0 issued behind the S' chain of block 1;
11 / 12 / 13 = CEILING PROBES of the production kernel (no global traffic | + no softmax | + no LDS reads): timing only,
their output is not an attention result (the agreement check skips them)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pandora_amd.ops_hip import HipOps  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--variants", default="9,1,3,5,101")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    ops = HipOps(dt, "cuda:0", diag=True)  # (variant overrides: the diagnostics build)
    # 101 / 102: pm_attention_fp8 (block-scaled e4m3 MFMA) with 32 / 64 query rows per wave
    variants = [int(v) for v in a.variants.split(",")]
    F = 16
    for (N, heads) in [(9216, 5), (2304, 10), (2560, 5), (640, 10), (576, 20)]:
        C = heads * 64
        qkv = torch.randn(F, N, 3 * C, device="cuda", dtype=dt)
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        fl = 4.0 * N * N * 64 * heads * F
        res = {vv: [] for vv in variants}
        ref = None
        for r in range(a.rounds + 1):
            for vv in variants:
                ops.lib.pm_debug_attn_variant(vv % 100)
                call = (lambda: ops.attention_fp8(q, k, v, heads)) if vv >= 100 else (lambda: ops.attention(q, k, v, heads))
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = 10
                e0.record()
                for _ in range(n):
                    o = call()
                e1.record()
                torch.cuda.synchronize()
                if r:
                    res[vv].append(e0.elapsed_time(e1) / n)
                elif vv in (11, 12, 13):
                    pass  # probes: not a result
                else:  # agreement of the variants (first round)
                    if ref is None:
                        ref = o.float()
                    else:
                        err = ((o.float() - ref).norm() / ref.norm()).item()
                        print(f"   N={N} variant {vv} vs {variants[0]}: rel diff {err:.2e}")
        for vv in variants:
            t = sorted(res[vv])
            med, mn = t[len(t) // 2], t[0]
            print(f"N={N:5d} heads={heads:2d} variant {vv}: median {med:.3f} ms ({fl / med / 1e9:7.1f} TF/s)  min {mn:.3f} ms ({fl / mn / 1e9:7.1f} TF/s)")
    ops.lib.pm_debug_attn_variant(0)


if __name__ == "__main__":
    main()
