"""Build the full-width model and run N graph-free forwards (for rocprofv3 --kernel-trace --stats: the
difference of two runs with different N is the per-forward kernel breakdown).
usage: python3 tools/fwd_only.py N [res]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pandora_amd import factory, synth  # noqa: E402
from open_pandora_amd.ops_hip import HipOps  # noqa: E402

n = int(sys.argv[1])
res = sys.argv[2] if len(sys.argv) > 2 else "320x512"
ops = HipOps(torch.bfloat16, "cuda:0")
pm = factory.build_diffusion(res, ops)
h, w = factory.RESOLUTIONS[res]["image_size"]
ins = synth.synth_inputs(h, w, 16, seed=123)
cond = {"c_crossattn": [ins["c_crossattn"].cuda()], "c_concat": [ins["c_concat"].cuda()]}
x = ins["x_T"].cuda()
ts = torch.full((1,), 500, device="cuda", dtype=torch.long)
fs = torch.tensor([15], device="cuda")
for _ in range(n):
    out = pm.apply_model(x, ts, cond, fs=fs)
torch.cuda.synchronize()
print("ok", float(out.float().abs().mean()))
