"""cProfile of the host side of one eager forward (full-width U-Net, 8x8 latent: GPU work negligible).
usage: python tools/host_profile.py"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pandora_amd import factory, synth  # noqa: E402
from open_pandora_amd.ops_hip import HipOps  # noqa: E402

ops = HipOps(torch.bfloat16, "cuda:0")
pm = factory.build_diffusion("320x512", ops)
ins = synth.synth_inputs(8, 8, 16, seed=123)
cond = {"c_crossattn": [ins["c_crossattn"].cuda()], "c_concat": [ins["c_concat"].cuda()]}
x = ins["x_T"].cuda()
ts = torch.full((1,), 500, device="cuda", dtype=torch.long)
fs = torch.tensor([15], device="cuda")
for _ in range(3):
    pm.apply_model(x, ts, cond, fs=fs)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    pm.apply_model(x, ts, cond, fs=fs)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
