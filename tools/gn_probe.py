"""gn_apply us per launch at the pyramid levels, integer totals with nsum = 1 / 16 and f32 totals (A/B of two libraries:
PANDORA_LIB=<other .so> python tools/gn_probe.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pandora_amd.ops_hip import HipOps  # noqa: E402


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


ops = HipOps(torch.bfloat16, "cuda:0")
F = 16
for res, (h, w) in (("320x512", (40, 64)), ("576x1024", (72, 128))):
    for C, div in [(320, 1), (640, 2), (1280, 4), (1280, 8)]:
        P = (h // div) * (w // div)
        x = torch.randn(F * P, C, device="cuda")
        g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
        tot = ops.groupnorm_stats(x, F, 32)                      # int64 limbs [F, 32, 4, 8] (or f32 with PANDORA_STATS_I64=0)
        tf = ops.totals_f32(tot).contiguous()
        cnt = float(F * P * (C // 32))
        t1 = timed(lambda: ops.groupnorm_apply(x, tot, g, b, 1e-5, F, True))
        t16 = timed(lambda: ops.groupnorm_apply(x, tot, g, b, 1e-5, 1, True, cnt))
        tf1 = timed(lambda: ops.groupnorm_apply(x, tf, g, b, 1e-5, F, True))
        print(f"[{res}] {F * P:7d} x {C:4d}: i64 per frame {t1:6.1f} us   i64 clip (nsum 16) {t16:6.1f} us   f32 per frame {tf1:6.1f} us")
