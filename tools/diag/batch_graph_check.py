"""diag: the batched CFG pair through the graph vs eager (tests/test_unet_gpu.py::test_batched_clips_forward_and_sampler)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import golden_recipe as gr  # noqa: E402
from open_pandora_amd import synth  # noqa: E402
from open_pandora_amd.ddim import DDIMSampler, _ForwardGraph  # noqa: E402
from open_pandora_amd.ddpm import LatentVisualDiffusion  # noqa: E402
from open_pandora_amd.ops_hip import HipOps  # noqa: E402
from open_pandora_amd.unet import UNetModel  # noqa: E402
from test_oracle_golden import RH_KW, load, rel  # noqa: E402

ops = HipOps(torch.float16, "cuda:0")
m = UNetModel(**dict(RH_KW, model_channels=64)).eval()
m.load_state_dict(synth.synth_state_dict(m, seed=gr.WEIGHT_SEED))
pm = LatentVisualDiffusion(m.bind(ops))
ins, cond, uc = gr.sampler_inputs(8, 8)
dev = lambda c: {k: [v.cuda() for v in lst] for k, lst in c.items()}
cd, ud = dev(cond), dev(uc)
x, t, fs = ins["x_T"].cuda(), torch.tensor([500]).cuda(), torch.tensor([15]).cuda()
e_c = pm.apply_model(x, t, cd, fs=fs); e_u = pm.apply_model(x, t, ud, fs=fs)
os.environ["PANDORA_CFG_BATCH"] = "1"
g = _ForwardGraph(pm, x, t, cd, ud, fs, {})
print("batched flag", g.batched)
for i in range(3):
    gc, gu = g(x, t)
    torch.cuda.synchronize()
    print(f"replay {i}: graph e_c vs eager {rel(gc.cpu(), e_c.cpu()):.2e}  e_u {rel(gu.cpu(), e_u.cpu()):.2e}")
x2 = x * 0.5
e_c2 = pm.apply_model(x2, t, cd, fs=fs)
gc, gu = g(x2, t)
torch.cuda.synchronize()
print(f"other input: graph e_c vs eager {rel(gc.cpu(), e_c2.cpu()):.2e}")
for ug in (False, True):
    S, eta, cfg = 5, 0.0, 4.0
    gold = load("ddim_small.npz")[f"S{S}_eta{eta:g}_cfg{cfg:g}"]
    smp = DDIMSampler(pm, use_graph=ug)
    y, _ = smp.sample(S=S, batch_size=1, shape=(4, 16, 8, 8), conditioning=cd, verbose=False, unconditional_guidance_scale=cfg,
                      unconditional_conditioning=ud, eta=eta, fs=fs, timestep_spacing="uniform_trailing", x_T=x)
    print(f"sampler batched use_graph={ug}: rel err vs reference {rel(y.cpu(), gold):.2e}")

# ---- where does the staleness come from? ----
g = _ForwardGraph(pm, x, t, cd, ud, fs, {})
gc, gu = g(x2, t)
torch.cuda.synchronize()
print("g.x rows == x2:", torch.equal(g.x[0:1], x2), torch.equal(g.x[1:2], x2))
cc = {k: [torch.cat([a, b_], 0) for a, b_ in zip(cd[k], ud[k])] for k in cd}
out = pm.apply_model(g.x, g.t, cc, fs=fs)
print(f"eager batched forward on the graph's own static input vs graph output: {rel(gc.cpu(), out[0:1].cpu()):.2e}; vs eager single {rel(out[0:1].cpu(), e_c2.cpu()):.2e}")
# stage by stage inside a capture: which op stops following the static input?
from open_pandora_amd.ddim import _capture_streams  # noqa: E402
side = _capture_streams(x.device)[0]
xs = torch.cat([x, x], 0)
gr2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr2, stream=side, capture_error_mode="thread_local"):
    xc = torch.cat([xs] + cc["c_concat"], dim=1)
    xr = xc.permute(1, 0, 2, 3, 4).reshape(8, 32, 64)
    h = ops.pack_input(xr.contiguous(), None)
xs[0:1].copy_(x2); xs[1:2].copy_(x2)
gr2.replay(); torch.cuda.synchronize()
xc_e = torch.cat([xs] + cc["c_concat"], dim=1)
xr_e = xc_e.permute(1, 0, 2, 3, 4).reshape(8, 32, 64).contiguous()
h_e = ops.pack_input(xr_e, None)
torch.cuda.synchronize()
print("captured cat follows:", torch.equal(xc, xc_e), " permute/reshape follows:", torch.equal(xr, xr_e), " pack_input follows:", torch.equal(h, h_e))

# ---- sequence test: fresh graph, inputs a, b, c, a; each against the eager batched forward on the same input ----
for mode in ("1", "0"):
    os.environ["PANDORA_CFG_BATCH"] = mode
    g = _ForwardGraph(pm, x, t, cd, ud, fs, {})
    seq = [x2, x * 0.25, x, x2]
    tt = [t, torch.tensor([300]).cuda(), t, torch.tensor([700]).cuda()]
    for i, (xi, ti) in enumerate(zip(seq, tt)):
        gc, gu = g(xi, ti)
        torch.cuda.synchronize()
        gcc, guc = gc.clone(), gu.clone()
        ec, eu = pm.apply_model(xi, ti, cd, fs=fs), pm.apply_model(xi, ti, ud, fs=fs)
        torch.cuda.synchronize()
        print(f"batch={mode} call {i}: graph vs eager single  e_c {rel(gcc.cpu(), ec.cpu()):.2e}  e_u {rel(guc.cpu(), eu.cpu()):.2e}")
