"""Does the row stride of A / W matter (L2 channel camping)?  Dense pm_gemm on slices of wider buffers.
usage: python tools/stride_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pandora_amd.ops_hip import HipOps  # noqa: E402
from tools.microbench import timeit  # noqa: E402

ops = HipOps(torch.bfloat16, "cuda:0")
for M, N, K in ((2560, 1280, 3840), (2560, 1280, 1280), (640, 1280, 5120), (10240, 640, 2560), (40960, 320, 1280), (2560, 1280, 11520)):
    for pad_a, pad_w in ((0, 0), (64, 0), (0, 64), (64, 64), (32, 32), (8, 8), (128, 128)):
        a = torch.randn(M, K + pad_a, device="cuda", dtype=torch.bfloat16)[:, :K]
        w = (torch.randn(N, K + pad_w, device="cuda", dtype=torch.bfloat16) * 0.02)[:, :K]
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        lib, P = ops.lib, lambda t: t.data_ptr()

        def run():
            rc = lib.pm_gemm(P(a), a.stride(0), P(w), w.stride(0), None, None, 0, P(out), N, M, N, K, 0, 0, ops.dt,
                             P(ops.workspace), ops.ws_bytes, None, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
        t = timeit(run)
        print(f"M={M} N={N} K={K} lda=K+{pad_a} ldw=K+{pad_w}: {t * 1e3:7.1f} us  {2.0 * M * N * K / t / 1e9:6.0f} TF/s", flush=True)
