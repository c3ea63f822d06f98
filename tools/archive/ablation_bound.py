"""Bounds for fusing side kernels away: the CFG pair of U-Net forwards replayed as the two-stream HIP graph with the
launches of one C-ABI entry point at a time DOUBLED (every call issued twice: same inputs, same outputs, results stay
valid).  What the step loses to a second copy of a launch family is what it can gain by removing the first - measured in
place (launch gaps, overlap with the other stream) and at the same data-dependent clock.  (Turning the entry point into a
no-op instead leaves garbage / NaN operands downstream: the chip draws less power, clocks higher, and the "gain" is
inflated 1.5-3x: first version of this tool.)
usage: python tools/ablation_bound.py [--res 320x512] [--reps 8]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pandora_amd import factory, synth  # noqa: E402
from open_pandora_amd.ddim import _ForwardGraph  # noqa: E402
from open_pandora_amd.ops_hip import HipOps  # noqa: E402

CASES = [("baseline", []),
         ("2x gn_finalize_colstats", ["pm_groupnorm_finalize_colstats"]),
         ("2x gn_stats", ["pm_groupnorm_stats"]),
         ("2x gn_apply", ["pm_groupnorm_apply"]),
         ("2x layernorm", ["pm_layernorm"]),
         ("2x split16", ["pm_split16"]),
         ("2x temporal attention", ["pm_attention_temporal"])]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", default="320x512")
    ap.add_argument("--reps", type=int, default=8)
    a = ap.parse_args()
    ops = HipOps(torch.bfloat16, "cuda:0")
    pm = factory.build_diffusion(a.res, ops)
    h, w = factory.RESOLUTIONS[a.res]["image_size"]
    ins = synth.synth_inputs(h, w, 16, seed=123)
    cond = {"c_crossattn": [ins["c_crossattn"].cuda()], "c_concat": [ins["c_concat"].cuda()]}
    uc = {"c_crossattn": [ins["uc_crossattn"].cuda()], "c_concat": [ins["c_concat"].cuda()]}
    x, ts, fs = ins["x_T"].cuda(), torch.full((1,), 500, device="cuda", dtype=torch.long), torch.tensor([15], device="cuda")
    real = {}
    base = None
    for name, fns in CASES:
        for f in fns:
            if not hasattr(ops.lib, f):
                print(f"{name}: no entry point {f}")
                continue
            real.setdefault(f, getattr(ops.lib, f))
            setattr(ops.lib, f, (lambda fn: (lambda *args: (fn(*args), fn(*args))[1]))(real[f]))
        with torch.no_grad():
            g = _ForwardGraph(pm, x, ts, cond, uc, fs, {})
            for _ in range(2):
                g(x, ts)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                g(x, ts)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        base = ms if base is None else base
        print(f"{a.res} {name:32s}: {ms:8.3f} ms per CFG pair ({100.0 * (ms - base) / base:+.1f} %)")
        for f, fn in real.items():
            setattr(ops.lib, f, fn)
        del g
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
