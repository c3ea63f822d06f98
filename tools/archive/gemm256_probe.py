"""Isolated timing of dense GEMM shapes: 128x128 kernels (PANDORA_GEMM256=0) vs the 256x256 kernel, set per process through
the environment (the choice is read once).  usage: PANDORA_GEMM256=0|1|2 [PANDORA_GEMM256_NODMA=1] python tools/gemm256_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pandora_amd.ops_hip import HipOps  # noqa: E402

ops = HipOps(torch.bfloat16, "cuda:0")
shapes = [(36864, 1920, 640, "none"), (9216, 3840, 1280, "none"), (36864, 5120, 640, "geglu"), (9216, 10240, 1280, "geglu"),
          (147456, 960, 320, "none"), (9216, 1280, 5120, "none"), (36864, 2560, 2560, "none"), (8192, 8192, 8192, "none")]
for M, N, K, act in shapes:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * K ** -0.5
    b = torch.zeros(N, device="cuda") if act == "geglu" else None
    for _ in range(3):
        ops.gemm(a, w, b, act=act)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        ops.gemm(a, w, b, act=act)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / n
    ch = ops.lib.pm_gemm_kernel_choice(M, N, K, 2 if act == "geglu" else 0, 0, ops.ws_bytes)
    print(f"M={M:6d} N={N:5d} K={K:5d} {act:5s} kernel {ch}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.0f} TF/s")
