"""A few launches of the dominant kernels at the U-Net's shapes, for rocprofv3 --pmc passes (HBM-side traffic:
FETCH_SIZE and WRITE_SIZE in separate runs; SQ counters in a third).  Shapes: the dense GEMM family's heaviest
(GEGLU ff1 at level 0 / level 1, the q|k|v projection, the f32-stream out-projection with residual), the level-0
3x3 conv, the LayerNorm + projection panel kernel (pm_ln_gemm), and the spatial self-attention at N = 2560 (320x512) and N = 9216 (576x1024)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pandora_amd.ops_hip import HipOps  # noqa: E402

ops = HipOps(torch.bfloat16, "cuda:0")
r = lambda *s: torch.randn(*s, device="cuda", dtype=torch.bfloat16)
z = lambda n: torch.zeros(n, device="cuda")
REP = 4
F, H, W, C = 16, 40, 64, 320
M = F * H * W
x = r(M, C)
wp = r(C, 9 * C) * 0.02
for _ in range(REP):
    ops.conv3x3(x, wp, z(C), F, H, W)                              # gemm_ring_kernel<A_CONV3X3_FAST>
w_ff1 = r(8 * C, C) * 0.05
for _ in range(REP):
    ops.gemm(x, w_ff1, z(8 * C), act="geglu")                      # M=40960 N=2560 K=320 geglu
x1 = r(16 * 20 * 32, 640)
w1 = r(5120, 640) * 0.02
for _ in range(REP):
    ops.gemm(x1, w1, z(5120), act="geglu")                         # M=10240 N=5120 K=640 geglu
w_qkv = r(3 * C, C) * 0.05
for _ in range(REP):
    ops.gemm(x, w_qkv)                                             # M=40960 N=960 K=320
w_o = r(C, C) * 0.05
res = torch.randn(M, C, device="cuda")
for _ in range(REP):
    ops.gemm(x, w_o, z(C), residual=res, stream=True)              # M=40960 N=320 K=320, f32 residual + output
xf = torch.randn(M, C, device="cuda") * 1.5
g1, b1 = torch.ones(C, device="cuda"), z(C)
for _ in range(REP):
    ops.ln_gemm(xf, g1, b1, w_qkv)                                 # pm_ln_gemm: LayerNorm + q|k|v, M=40960 N=960 K=320
for _ in range(REP):
    ops.ln_gemm(xf, g1, b1, w_ff1, z(8 * C), act="geglu")          # pm_ln_gemm: LayerNorm + GEGLU ff1, N=2560
# the 576x1024 level-1 / level-0 shapes the 256x128 ring kernel takes (gemm_ringw_kernel<A_DENSE> / <A_CONV3X3_FAST>)
x2 = r(16 * 36 * 64, 640)
for _ in range(REP):
    ops.gemm(x2, w1, z(5120), act="geglu")                         # M=36864 N=5120 K=640 geglu
xl0 = r(16 * 72 * 128, C)
for _ in range(REP):
    ops.conv3x3(xl0, wp, z(C), F, 72, 128)                         # M=147456 N=320 K=2880
del x2, xl0
xf2 = torch.randn(16 * 72 * 128, C, device="cuda") * 1.5
for _ in range(REP):
    ops.ln_gemm(xf2, g1, b1, w_ff1, z(8 * C), act="geglu")         # the same at 576x1024: M=147456
del xf2
for N in (2560, 9216):
    qkv = r(16, N, 960)
    for _ in range(REP):
        ops.attention(qkv[..., :320], qkv[..., 320:640], qkv[..., 640:], 5)
torch.cuda.synchronize()
