"""Host-side launch cost of one eager forward: full-width U-Net on an 8x8 latent (GPU work negligible).
usage: python tools/cpu_issue.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pandora_amd import factory, synth  # noqa: E402
from open_pandora_amd.ops_hip import HipOps  # noqa: E402

ops = HipOps(torch.bfloat16, "cuda:0")
pm = factory.build_diffusion("320x512", ops)
for (h, w) in ((8, 8), (40, 64)):
    ins = synth.synth_inputs(h, w, 16, seed=123)
    cond = {"c_crossattn": [ins["c_crossattn"].cuda()], "c_concat": [ins["c_concat"].cuda()]}
    x = ins["x_T"].cuda()
    ts = torch.full((1,), 500, device="cuda", dtype=torch.long)
    fs = torch.tensor([15], device="cuda")
    for _ in range(3):
        pm.apply_model(x, ts, cond, fs=fs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        pm.apply_model(x, ts, cond, fs=fs)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"latent {h}x{w}: host issue {1e3 * (t1 - t0) / 5:.2f} ms/forward, wall {1e3 * (t2 - t0) / 5:.2f} ms/forward")
