"""diag: read-before-write hunt.  Run the (batched) forward eagerly on POISONED allocator memory: every cached free block is
filled with NaN first, so an op that reads memory it (or an earlier op) never wrote shows up as NaN in its output; the first
such op is printed."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import golden_recipe as gr  # noqa: E402
from open_pandora_amd import synth  # noqa: E402
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
cc = {k: [torch.cat([a, b_], 0) for a, b_ in zip(cd[k], ud[k])] for k in cd}
x2, t2 = torch.cat([x, x], 0), torch.cat([t, t], 0)


def poison():
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    big = torch.full((1 << 28,), float("nan"), device="cuda")  # 1 GiB of NaN, then back to the allocator's cache
    torch.cuda.synchronize()
    del big


log = []
names = ["gemm", "conv3x3", "conv_t3", "groupnorm", "groupnorm_stats", "groupnorm_apply", "layernorm", "ln_gemm", "attention",
         "attention_temporal", "pack_input", "unpack_output", "gemv", "split16", "split16_upsample2x", "_stats_end"]
for n in names:
    f = getattr(ops, n)

    def wrap(f=f, n=n):
        def g(*a, **k):
            y = f(*a, **k)
            outs = y if isinstance(y, tuple) else (y,)
            torch.cuda.synchronize()
            bad = any(torch.is_tensor(o) and o.is_floating_point() and not torch.isfinite(o.float()).all() for o in outs)
            shapes = [tuple(o.shape) for o in outs if torch.is_tensor(o)]
            log.append((n, shapes, bad))
            return y
        return g
    setattr(ops, n, wrap())

for label, (xx, tt, c) in (("single clip", (x, t, cd)), ("batched clips", (x2, t2, cc))):
    ref = pm.apply_model(xx, tt, c, fs=fs).clone()
    poison()
    log.clear()
    y = pm.apply_model(xx, tt, c, fs=fs)
    torch.cuda.synchronize()
    first = next(((i, e) for i, e in enumerate(log) if e[2]), None)
    print(f"{label}: result on poisoned memory vs clean {rel(y.float().cpu(), ref.float().cpu()) if torch.isfinite(y).all() else float('nan'):.2e}; "
          f"finite {bool(torch.isfinite(y).all())}; first op with a non-finite output: {first} (of {len(log)} ops)")
    if first:
        i = first[0]
        print("   ops around it:", log[max(0, i - 3):i + 2])
