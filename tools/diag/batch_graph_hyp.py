"""diag: which ingredient breaks replays >= 2 of the batched CFG graph"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import golden_recipe as gr  # noqa: E402
from open_pandora_amd import synth  # noqa: E402
from open_pandora_amd import ddim  # noqa: E402
from open_pandora_amd.ddpm import LatentVisualDiffusion  # noqa: E402
from open_pandora_amd.ops_hip import HipOps  # noqa: E402
from open_pandora_amd.unet import UNetModel  # noqa: E402
from test_oracle_golden import RH_KW, rel  # noqa: E402

ops = HipOps(torch.float16, "cuda:0")
m = UNetModel(**dict(RH_KW, model_channels=64)).eval()
m.load_state_dict(synth.synth_state_dict(m, seed=gr.WEIGHT_SEED))
pm = LatentVisualDiffusion(m.bind(ops))
ins, cond, uc = gr.sampler_inputs(8, 8)
dev = lambda c: {k: [v.cuda() for v in lst] for k, lst in c.items()}
cd, ud = dev(cond), dev(uc)
x, t, fs = ins["x_T"].cuda(), torch.tensor([500]).cuda(), torch.tensor([15]).cuda()
seq = [(x * 0.5, t), (x * 0.25, torch.tensor([300]).cuda()), (x, t)]
want = [(pm.apply_model(xi, ti, cd, fs=fs).clone(), pm.apply_model(xi, ti, ud, fs=fs).clone()) for xi, ti in seq]


def run(label):
    g = ddim._ForwardGraph(pm, x, t, cd, ud, fs, {})
    errs = []
    for (xi, ti), (ec, eu) in zip(seq, want):
        gc, gu = g(xi, ti)
        torch.cuda.synchronize()
        errs.append(f"{rel(gc.cpu(), ec.cpu()):.1e}/{rel(gu.cpu(), eu.cpu()):.1e}")
    print(f"{label}: batched={g.batched}  e_c/e_u error per call: {errs}", flush=True)
    g.close()


os.environ["PANDORA_CFG_BATCH"] = "0"
os.environ["PANDORA_CFG_STREAMS"] = "0"
run("H1 unbatched, ONE stream")
os.environ["PANDORA_CFG_STREAMS"] = "1"
os.environ["PANDORA_CFG_BATCH"] = "1"
run("H0 batched (as shipped)")
ops.conv_t3_clips = False
run("H2 batched, per-clip temporal conv launches")
ops.conv_t3_clips = True
orig = torch.cuda.graph


class G(orig):
    def __init__(self, *a, **k):
        k["capture_error_mode"] = "global"
        super().__init__(*a, **k)


torch.cuda.graph = G
run("H3 batched, global capture mode")
torch.cuda.graph = orig
# H4: the outputs cloned inside the graph
