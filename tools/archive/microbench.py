"""Per-kernel timings at the real shapes of the U-Net (run on the GPU box).
usage: python tools/microbench.py [--res 320x512|576x1024] [--dtype bf16|f16]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pandora_amd.ops_hip import HipOps  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters  # ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", default="320x512")
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    h, w = [int(v) // 8 for v in a.res.split("x")]
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    ops = HipOps(dt, "cuda:0")
    F = 16
    r = lambda *s: torch.randn(*s, device="cuda", dtype=dt)
    print(f"# latent {h}x{w}, F={F}, dtype={a.dtype}")
    for lvl, (C, div) in enumerate([(320, 1), (640, 2), (1280, 4), (1280, 8)]):
        H, W = h // div, w // div
        M = F * H * W
        heads = C // 64
        # conv3x3 C->C
        x = r(M, C); wp = r(C, 9 * C) * 0.02; b = torch.zeros(C, device="cuda")
        t = timeit(lambda: ops.conv3x3(x, wp, b, F, H, W))
        fl = 2.0 * M * C * 9 * C
        print(f"L{lvl} conv3x3 {C}->{C} M={M}: {t:.3f} ms  {fl / t / 1e9:.1f} TF/s")
        wt = r(C, 3 * C) * 0.02
        t = timeit(lambda: ops.conv_t3(x, wt, b, F, H * W))
        fl = 2.0 * M * C * 3 * C
        print(f"L{lvl} conv_t3 {C}: {t:.3f} ms  {fl / t / 1e9:.1f} TF/s")
        # linear C->C, ff
        wl = r(C, C) * 0.02
        t = timeit(lambda: ops.gemm(x, wl, b, x))
        print(f"L{lvl} gemm {C}x{C}: {t:.3f} ms  {2.0 * M * C * C / t / 1e9:.1f} TF/s")
        wf = r(8 * C, C) * 0.02; bf = torch.zeros(8 * C, device="cuda")
        t = timeit(lambda: ops.gemm(x, wf, bf, act="geglu"))
        print(f"L{lvl} gemm geglu {C}->{8*C}: {t:.3f} ms  {2.0 * M * C * 8 * C / t / 1e9:.1f} TF/s")
        g4 = r(M, 4 * C); w2 = r(C, 4 * C) * 0.02
        t = timeit(lambda: ops.gemm(g4, w2, b, x))
        print(f"L{lvl} gemm ff2 {4*C}->{C}: {t:.3f} ms  {2.0 * M * C * 4 * C / t / 1e9:.1f} TF/s")
        # norms
        ga = torch.ones(C, device="cuda"); be = torch.zeros(C, device="cuda")
        t = timeit(lambda: ops.groupnorm(x, ga, be, 1e-5, F, True))
        print(f"L{lvl} groupnorm+silu per-frame: {t:.3f} ms  {3 * M * C * 2 / t / 1e6:.0f} GB/s (3 passes)")
        t = timeit(lambda: ops.groupnorm(x, ga, be, 1e-5, 1, True))
        print(f"L{lvl} groupnorm+silu (T,H,W): {t:.3f} ms  {3 * M * C * 2 / t / 1e6:.0f} GB/s")
        t = timeit(lambda: ops.layernorm(x, ga, be))
        print(f"L{lvl} layernorm: {t:.3f} ms  {2 * M * C * 2 / t / 1e6:.0f} GB/s")
        if lvl < 3 or True:
            N = H * W
            qkv = r(F, N, 3 * C)
            t = timeit(lambda: ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads))
            fl = 4.0 * N * N * 64 * heads * F
            print(f"L{lvl} spatial attn N={N} heads={heads}: {t:.3f} ms  {fl / t / 1e9:.1f} TF/s")
            t = timeit(lambda: ops.attention_temporal(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads))
            print(f"L{lvl} temporal attn: {t:.3f} ms  {4 * M * C * 2 / t / 1e6:.0f} GB/s")
            kt = r(1, 77, C); ki = r(F, 16, C); q = r(F, N, C)
            t = timeit(lambda: ops.attention(q, kt, kt, heads, ki, ki, 1.0))
            print(f"L{lvl} cross attn 77+16: {t:.3f} ms")


if __name__ == "__main__":
    main()
