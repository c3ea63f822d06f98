"""A/B of pm_ln_gemm (LayerNorm + projection, one panel kernel) against the pm_layernorm + pm_gemm pair on the
shapes of the U-Net's shallowest level (K = 320) at both BASELINE resolutions.  Interleaved in one process (MI355X
devices and clocks differ between runs), random data, HIP events over `reps` back-to-back launches.

    python tools/lngemm_bench.py [--reps 50] [--nsplit 0]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pandora_amd.ops_hip import HipOps  # noqa: E402

# (label, M, K, N, mode)
SHAPES = [
    ("320x512 L0 qkv", 40960, 320, 960, "scale"), ("320x512 L0 a2_q", 40960, 320, 320, "plain"),
    ("320x512 L0 geglu", 40960, 320, 2560, "geglu"),
    ("576x1024 L0 qkv", 147456, 320, 960, "scale"), ("576x1024 L0 a2_q", 147456, 320, 320, "plain"),
    ("576x1024 L0 geglu", 147456, 320, 2560, "geglu"),
]


def time_ms(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--nsplit", type=str, default="0", help="comma list of forced column splits to try (0 = host choice)")
    ap.add_argument("--dtype", default="bf16")
    args = ap.parse_args()
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    ops = HipOps(dtype, "cuda:0")
    print(f"# {torch.cuda.get_device_name(0)} dtype={args.dtype} reps={args.reps}")
    print(f"{'shape':22s} {'M':>7s} {'K':>4s} {'N':>5s}  {'LN us':>7s} {'gemm us':>8s} {'pair us':>8s} "
          f"{'fused us':>9s} {'speedup':>8s} {'fused TF/s':>10s}  nsplit")
    tot_pair = tot_fused = 0.0
    for label, M, K, N, mode in SHAPES:
        x = torch.randn(M, K, device="cuda") * 1.5
        g, b = torch.rand(K, device="cuda") + 0.5, torch.randn(K, device="cuda") * 0.1
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dtype)
        bias = torch.randn(N, device="cuda") if mode == "geglu" else None
        scale = torch.ones(N, device="cuda") if mode == "scale" else None
        act = "geglu" if mode == "geglu" else "none"
        y = ops.layernorm(x, g, b)
        t_ln = time_ms(lambda: ops.layernorm(x, g, b), args.reps)
        t_g = time_ms(lambda: ops.gemm(y, w, bias, act=act, col_scale=scale), args.reps)
        os.environ.pop("PANDORA_LNGEMM_NSPLIT", None)
        best = None
        for ns in [int(v) for v in args.nsplit.split(",")]:
            if ns:
                os.environ["PANDORA_LNGEMM_NSPLIT"] = str(ns)
            else:
                os.environ.pop("PANDORA_LNGEMM_NSPLIT", None)
            t_f = time_ms(lambda: ops.ln_gemm(x, g, b, w, bias, act=act, col_scale=scale), args.reps)
            if best is None or t_f < best[0]:
                best = (t_f, ns)
            if len(args.nsplit.split(",")) > 1:
                print(f"    nsplit={ns}: {t_f * 1e3:8.1f} us")
        os.environ.pop("PANDORA_LNGEMM_NSPLIT", None)
        t_f, ns = best
        tf = 2.0 * M * N * K / (t_f * 1e-3) / 1e12
        print(f"{label:22s} {M:7d} {K:4d} {N:5d}  {t_ln * 1e3:7.1f} {t_g * 1e3:8.1f} {(t_ln + t_g) * 1e3:8.1f} "
              f"{t_f * 1e3:9.1f} {(t_ln + t_g) / t_f:8.2f} {tf:10.0f}  {ns}")
        tot_pair += t_ln + t_g
        tot_fused += t_f
    print(f"# sum: pair {tot_pair:.3f} ms, fused {tot_fused:.3f} ms")


if __name__ == "__main__":
    main()
