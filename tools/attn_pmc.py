"""A few launches of every spatial self-attention variant at N = 9216 (16 frames x 5 heads), for rocprofv3 --pmc:
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY \
      SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d out -- python3 tools/attn_pmc.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pandora_amd.ops_hip import HipOps  # noqa: E402

ops = HipOps(torch.bfloat16, "cuda:0", diag=True)  # (variant overrides: the diagnostics build)
N, heads, F = (int(sys.argv[1]) if len(sys.argv) > 1 else 9216), 5, 16
C = heads * 64
qkv = torch.randn(F, N, 3 * C, device="cuda", dtype=torch.bfloat16)
variants = (0,) if (len(sys.argv) > 2 and sys.argv[2] == "prod") else (1, 16, 3, 5)  # 0 = the production choice; 1 / 16 = the two MFMA shapes
for vv in variants:
    ops.lib.pm_debug_attn_variant(vv)
    for _ in range(4):
        ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads)
    torch.cuda.synchronize()
