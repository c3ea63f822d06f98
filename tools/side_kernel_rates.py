"""HBM-bound side kernels (VERDICT r04 #7b): rate of ALGORITHMIC bytes per second of gn_apply, gn_stats, layernorm and the
temporal attention at the four pyramid levels of both resolutions and at 4x / 16x the level-0 size - separates what the kernel
can stream (the large sizes) from what a launch of the model's size can reach at all (ramp + tail of a 10-40 us launch).
usage (GPU box): python tools/side_kernel_rates.py > gpurun_out/side_kernel_rates.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pandora_amd.ops_hip import HipOps  # noqa: E402

PEAK = 8.0e12


def timed(fn, reps=40):
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
        best = min(best, e0.elapsed_time(e1) / reps * 1e-3)
    return best


def main():
    dt = torch.bfloat16
    ops = HipOps(dt, "cuda:0")
    F = 16
    print(f"{'kernel':22s} {'rows x C':>16s} {'MB':>8s} {'us':>8s} {'TB/s':>6s} {'of 8 TB/s':>9s}")
    for res, (h, w) in (("320x512", (40, 64)), ("576x1024", (72, 128)), ("4x", (144, 256)), ("16x", (288, 512))):
        levels = [(320, 1), (640, 2), (1280, 4), (1280, 8)] if res in ("320x512", "576x1024") else [(320, 1)]
        for C, div in levels:
            P = (h // div) * (w // div)
            M = F * P
            x = torch.randn(M, C, device="cuda")
            g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
            tot = ops.groupnorm_stats(x, F, 32)
            rows = []
            t = timed(lambda: ops.groupnorm_apply(x, tot, g, b, 1e-5, F, True))
            rows.append(("gn_apply f32->16+silu", M * C * 6, t))
            t = timed(lambda: ops.groupnorm_stats(x, F, 32))
            rows.append(("gn_stats f32", M * C * 4, t))
            t = timed(lambda: ops.layernorm(x, g, b))
            rows.append(("layernorm f32->16", M * C * 6, t))
            heads = C // 64
            q = torch.randn(F, P, 3 * C, device="cuda", dtype=dt)
            t = timed(lambda: ops.attention_temporal(q[..., :C], q[..., C:2 * C], q[..., 2 * C:], heads))
            rows.append(("attention_temporal", M * C * 2 * 4, t))
            for name, byts, t in rows:
                print(f"{name:22s} {f'{M} x {C}':>16s} {byts / 1e6:8.1f} {t * 1e6:8.1f} {byts / t / 1e12:6.2f} {byts / t / PEAK:9.2f}   [{res}]")
            del x, q
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
