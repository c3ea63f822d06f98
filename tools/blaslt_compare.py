"""How far are the hand-written GEMM kernels from the vendor library on this chip?  torch.matmul (hipBLASLt / rocBLAS on
ROCm) against pm_gemm on the U-Net's dense shapes, plain product only (no fused epilogue on either side), bf16.
usage: python tools/blaslt_compare.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pandora_amd.ops_hip import HipOps  # noqa: E402

ops = HipOps(torch.bfloat16, "cuda:0", diag=True)  # (diagnostics build: pm_debug_gemm_wide forces / forbids the 256x256 assembly-loop kernel)
SHAPES = [(147456, 960, 320), (36864, 1920, 640), (9216, 3840, 1280), (2304, 3840, 1280),
          (147456, 320, 1280), (36864, 640, 2560), (9216, 1280, 5120),
          (147456, 2560, 320), (36864, 5120, 640), (9216, 10240, 1280),
          (40960, 960, 320), (10240, 1920, 640), (2560, 3840, 1280), (40960, 2560, 320), (2560, 1280, 5120),
          (8192, 8192, 8192)]


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


for M, N, K in SHAPES:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * K ** -0.5
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.lib.pm_debug_gemm_wide(1)  # the library's own choice
    t_pm = bench(lambda: ops.gemm(a, w, out=out))
    ops.lib.pm_debug_gemm_wide(2)  # gemm_wide wherever it is legal
    t_wd = bench(lambda: ops.gemm(a, w, out=out))
    ops.lib.pm_debug_gemm_wide(1)
    wt = w.t()
    t_lt = bench(lambda: torch.matmul(a, wt, out=out))
    fl = 2.0 * M * N * K
    print(f"M={M:6d} N={N:5d} K={K:5d}: pm_gemm {t_pm:8.1f} us {fl / t_pm / 1e6:6.0f} TF/s | gemm_wide forced {t_wd:8.1f} us {fl / t_wd / 1e6:6.0f} TF/s"
          f" | torch.matmul {t_lt:8.1f} us {fl / t_lt / 1e6:6.0f} TF/s | pm / lib time {t_pm / t_lt:5.2f} | best pm / lib {min(t_pm, t_wd) / t_lt:5.2f}")
