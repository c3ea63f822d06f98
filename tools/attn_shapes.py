"""The two bf16 MFMA shapes of the spatial self-attention at the SAME per-wave output tile (cdna guide
rule 28): variant 1 = attn_self_kernel on v_mfma_f32_32x32x16, variant 16 = attn_self16_kernel on v_mfma_f32_16x16x32
(csrc/attn16.hip).  Interleaved rounds in ONE process on random data; per variant: wall time / TFLOP/s AND the in-kernel
clock the chip holds under it (d s_memtime / d s_memrealtime x 100 MHz stamped around every workgroup's body by the
DIAGNOSTICS build, after >= 2 s of back-to-back launches; median over workgroups), i.e. cycles AND wall.
usage: python tools/attn_shapes.py [--dtype bf16] [--rounds 7] [--variants 0,16]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pandora_amd.ops_hip import HipOps  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--variants", default="1,16")
    ap.add_argument("--shapes", default="9216x5,2304x10,2560x5,640x10,576x20")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    ops = HipOps(dt, "cuda:0", diag=True)
    lib = ops.lib
    variants = [int(v) for v in a.variants.split(",")]
    F = 16
    print(f"# {torch.cuda.get_device_name(0)}  dtype {a.dtype}  random N(0,1) q|k|v, {F} frames; interleaved rounds in one process")
    for shp in a.shapes.split(","):
        N, heads = (int(x) for x in shp.split("x"))
        C = heads * 64
        qkv = torch.randn(F, N, 3 * C, device="cuda", dtype=dt)
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        fl = 4.0 * N * N * 64 * heads * F
        grid = ((N + 127) // 128) * F * heads
        stamps = torch.zeros(2 * grid, dtype=torch.int64, device="cuda")
        res = {vv: [] for vv in variants}
        clk = {vv: [] for vv in variants}
        ref = None
        n = 10 if N >= 2000 else 50
        for r in range(a.rounds + 1):
            for vv in variants:
                lib.pm_debug_attn_variant(vv)
                lib.pm_debug_attn_stamps(None)
                if r == 0:  # agreement of the shapes
                    o = ops.attention(q, k, v, heads).float()
                    if ref is None:
                        ref = o
                    else:
                        print(f"   N={N} variant {vv} vs {variants[0]}: rel diff {((o - ref).norm() / ref.norm()).item():.2e}")
                    continue
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    ops.attention(q, k, v, heads)
                e1.record()
                torch.cuda.synchronize()
                res[vv].append(e0.elapsed_time(e1) / n)
        # in-kernel clock: >= 2 s of back-to-back launches of the variant, then one stamped launch (its stamps are the last
        # thing written), repeated 3 times
        for vv in variants:
            lib.pm_debug_attn_variant(vv)
            for rep in range(3):
                lib.pm_debug_attn_stamps(None)
                t0 = time.time()
                while time.time() - t0 < 2.0:
                    for _ in range(20):
                        ops.attention(q, k, v, heads)
                    torch.cuda.synchronize()
                for _ in range(20):
                    ops.attention(q, k, v, heads)
                lib.pm_debug_attn_stamps(stamps.data_ptr())
                ops.attention(q, k, v, heads)
                lib.pm_debug_attn_stamps(None)
                torch.cuda.synchronize()
                st = stamps.view(grid, 2).double()
                ghz = (st[:, 0] / st[:, 1].clamp_min(1.0) * 0.1)
                clk[vv].append((ghz.median().item(), st[:, 0].median().item()))
        for vv in variants:
            t = sorted(res[vv])
            med, mn = t[len(t) // 2], t[0]
            c = sorted(clk[vv])[1]
            print(f"N={N:5d} heads={heads:2d} variant {vv:2d} ({'32x32x16, 3 waves/SIMD' if vv == 1 else '16x16x32, 4 waves/SIMD' if vv == 16 else 'auto' if vv == 0 else '16x16x32 with per-lane K/V addresses' if vv == 17 else '16x16x32 with the row sums on the vector pipe' if vv == 18 else '16x16x32 with the compiler-chosen PV issue order' if vv == 19 else 'other'}): "
                  f"median {med:.3f} ms ({fl / med / 1e9:7.1f} TF/s = {fl / med / 1e9 / 2500:.3f} of 2.5 PF)  "
                  f"min {mn:.3f} ms ({fl / mn / 1e9:7.1f} TF/s)  in-kernel clock {c[0]:.3f} GHz, "
                  f"{c[1] / 1e3:.1f} k cycles per workgroup")
    lib.pm_debug_attn_variant(0)
    lib.pm_debug_attn_stamps(None)


if __name__ == "__main__":
    main()
