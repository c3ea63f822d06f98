"""A few launches of the dense GEMM kernels on the shapes the stream kernel serves, for rocprofv3 --pmc (VERDICT r05 #1d):
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY \
      SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d out -- python3 tools/gemm_pmc.py
then python tools/pmc_table.py out  (per kernel and grid: clock, MFMA pipe busy of SIMD cycles)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pandora_amd.ops_hip import HipOps  # noqa: E402

ops = HipOps(torch.bfloat16, "cuda:0")
g = torch.Generator(device="cuda").manual_seed(1)
SHAPES = [(8192, 8192, 8192, "none"), (9216, 10240, 1280, "geglu"), (36864, 5120, 640, "geglu"), (147456, 2560, 320, "geglu"),
          (40960, 2560, 320, "geglu"), (36864, 640, 2560, "none"), (147456, 320, 1280, "none")]
for M, N, K, act in SHAPES:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16, generator=g)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16, generator=g) * 0.05
    b = torch.randn(N, device="cuda", dtype=torch.float32, generator=g)
    for _ in range(4):
        ops.gemm(a, w, b, act=act)
    torch.cuda.synchronize()
    del a, w
