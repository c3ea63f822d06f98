import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_pandora_amd.ops_hip import HipOps
for dt in (torch.float16, torch.bfloat16):
    ops = HipOps(dt, "cuda:0")
    for (M, N, K) in ((77, 4992, 1024), (77, 24960, 1024), (300, 320, 320)):
        a = torch.randn(M, K, device="cuda", dtype=dt)
        w = torch.randn(N, K, device="cuda", dtype=dt)
        try:
            y = ops.gemm(a, w)
            torch.cuda.synchronize()
            ref = a.float() @ w.float().t()
            print(dt, M, N, K, "ok", float((y.float() - ref).norm() / ref.norm()))
        except Exception as e:
            print(dt, M, N, K, "FAILED", repr(e)[:200])
