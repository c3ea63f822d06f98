"""Run a few representative GEMM-family launches (for rocprofv3 --pmc runs)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pandora_amd.ops_hip import HipOps
ops = HipOps(torch.bfloat16, "cuda:0")
r = lambda *s: torch.randn(*s, device="cuda", dtype=torch.bfloat16)
F, H, W, C = 16, 40, 64, 320
x = r(F * H * W, C); wp = r(C, 9 * C) * 0.02; b = torch.zeros(C, device="cuda")
for _ in range(5):
    ops.conv3x3(x, wp, b, F, H, W)
x1 = r(16 * 20 * 32, 640); w1 = r(5120, 640) * 0.02; b1 = torch.zeros(5120, device="cuda")
for _ in range(5):
    ops.gemm(x1, w1, b1, act="geglu")
qkv = r(16, 2560, 960)
for _ in range(5):
    ops.attention(qkv[..., :320], qkv[..., 320:640], qkv[..., 640:], 5)
torch.cuda.synchronize()
