"""FETCH_SIZE calibration for the GEMM loaders' access pattern (8 lanes x 16 B = one 128-B line per row):
a GEMM that reads every A byte exactly once (one column tile) - does the counter see the bytes or half?
usage: rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -o f -- python3 tools/fetch_calib.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pandora_amd.ops_hip import HipOps
ops = HipOps(torch.bfloat16, "cuda:0")
for M, N, K in ((40960, 128, 320), (40960, 128, 1280)):
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    torch.cuda.synchronize()
    for _ in range(3):
        ops.gemm(a, w)
    torch.cuda.synchronize()
    print(f"M={M} N={N} K={K}: A bytes {M * K * 2 / 1024:.0f} KiB, W {N * K * 2 / 1024:.0f} KiB")
