"""Per-site contribution of the [hi | lo] norm outputs (HipOps parity sites): for every
(kind, level) site class of the U-Net, the full-width 10-step CFG-4 FRAMES error against the real reference's fixture with
that class taken OUT of the full parity configuration (leave-one-out) and with ONLY that class on, next to the time of one
graph-replayed CFG step.  From that table a greedy pass drops classes while the frames stay <= target.
usage (GPU box): python tools/parity_sites.py [--res 320x512] [--target 0.97e-3] [--out gpurun_out/parity_sites.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from open_pandora_amd import factory, synth  # noqa: E402
from open_pandora_amd.autoencoder import AutoencoderKL  # noqa: E402
from open_pandora_amd.ddim import DDIMSampler  # noqa: E402
from open_pandora_amd.ops_hip import HipOps  # noqa: E402

WEIGHT_SEED, INPUT_SEED = 20230211, 123  # (the fixtures' recipe: oracle/golden_recipe.py - nothing under oracle/ is imported here)


def rel_to_fixture(t, g, key):
    """relative L2 error against a fixture record: the whole tensor where the record holds it, else its prime-stride slice"""
    y = t.detach().float().cpu().reshape(-1)
    if f"{key}/full" in g:
        ref = torch.from_numpy(g[f"{key}/full"]).reshape(-1)
    else:
        ref = torch.from_numpy(g[f"{key}/slice"])
        y = y[::int(g[f"{key}/stride"])][:ref.numel()]
    return float((y.double() - ref.double()).norm() / ref.double().norm())


def sampler_inputs(h, w):
    ins = synth.synth_inputs(h, w, 16, seed=INPUT_SEED)
    cond = {"c_crossattn": [ins["c_crossattn"]], "c_concat": [ins["c_concat"]]}
    uc = {"c_crossattn": [ins["uc_crossattn"]], "c_concat": [ins["c_concat"]]}
    return ins, cond, uc


KINDS = ["gn3", "gnt", "gnp", "lns1", "lns2", "lns3", "lnt1", "lnt2", "lnt3", "split"]
ALL = [(k, lv) for k in KINDS for lv in range(4)]


def measure(res, sites, g, S=10, time_steps=6):
    h, w = factory.RESOLUTIONS[res]["image_size"]
    parity = True if sites == "all" else (False if not sites else sites)
    ops = HipOps(torch.float16, "cuda:0", parity=parity)
    pm = factory.build_diffusion(res, ops, seed=WEIGHT_SEED)
    ins, cond, uc = sampler_inputs(h, w)
    dev = lambda c: {k: [t.cuda() for t in v] for k, v in c.items()}
    smp = DDIMSampler(pm)
    kw = dict(batch_size=1, shape=(4, 16, h, w), conditioning=dev(cond), verbose=False, unconditional_guidance_scale=4.0,
              unconditional_conditioning=dev(uc), eta=0.0, fs=torch.tensor([15]).cuda(), timestep_spacing="uniform_trailing",
              x_T=ins["x_T"].cuda())
    z, _ = smp.sample(S=S, **kw)
    # step time: the same loop again (graph captured above), timed
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    smp.sample(S=time_steps, **kw)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / time_steps
    ae = AutoencoderKL()
    ae.load_state_dict(synth.synth_state_dict(ae, seed=WEIGHT_SEED))
    frames = ae.bind(HipOps(torch.float16, "cuda:0", parity=True)).decode_first_stage(z)
    e_z, e_f = rel_to_fixture(z, g, "latent"), rel_to_fixture(frames, g, "frames")
    smp.close()
    del pm, ae, smp
    torch.cuda.empty_cache()
    return e_z, e_f, ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", default="320x512")
    ap.add_argument("--target", type=float, default=0.97e-3)
    ap.add_argument("--out", default="gpurun_out/parity_sites.json")
    ap.add_argument("--quick", action="store_true", help="site classes by kind only (all four levels together)")
    ap.add_argument("--eval", action="store_true", help="only: default / full parity / ops_hip.SELECTIVE_PARITY_SITES")
    a = ap.parse_args()
    h, w = factory.RESOLUTIONS[a.res]["image_size"]
    g = np.load(os.path.join(ROOT, "tests", "golden", f"frames_full_{h}x{w}_s10_eta0.npz"))
    if a.eval:
        from open_pandora_amd import ops_hip
        out = {}
        for name, sites in (("none", []), ("all", "all"), ("selective", sorted(ops_hip.SELECTIVE_PARITY_SITES))):
            out[name] = measure(a.res, sites, g)
            print(f"{a.res} {name:9s}: latent {out[name][0]:.3e} frames {out[name][1]:.3e} step {out[name][2]:.1f} ms "
                  f"({out[name][2] / out['none'][2]:.3f} x default)", flush=True)
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        json.dump(out, open(a.out, "w"), indent=1)
        return
    rows = {}
    rows["none"] = measure(a.res, [], g)
    rows["all"] = measure(a.res, "all", g)
    print(f"none: latent {rows['none'][0]:.3e} frames {rows['none'][1]:.3e} step {rows['none'][2]:.1f} ms", flush=True)
    print(f"all : latent {rows['all'][0]:.3e} frames {rows['all'][1]:.3e} step {rows['all'][2]:.1f} ms", flush=True)
    classes = [[(k, lv) for lv in range(4)] for k in KINDS] if a.quick else [[s] for s in ALL]
    table = []
    for cls in classes:
        rest = [s for s in ALL if s not in cls]
        ez, ef, ms = measure(a.res, rest, g)
        name = cls[0][0] if a.quick else f"{cls[0][0]}@{cls[0][1]}"
        d2 = ef ** 2 - rows["all"][1] ** 2           # error^2 this class removes when it is ON
        dt = rows["all"][2] - ms                     # step time it costs
        table.append({"site": name, "frames_without": ef, "err2_bought": d2, "ms_cost": dt})
        print(f"without {name:8s}: frames {ef:.3e} (err^2 +{d2:.2e})  step {ms:.1f} ms (saves {dt:+.2f})", flush=True)
    # greedy: drop the classes with the least error bought per ms while the (additive) estimate stays under the target
    budget = a.target ** 2 - rows["all"][1] ** 2
    order = sorted(table, key=lambda r: (max(r["err2_bought"], 0.0) + 1e-12) / max(r["ms_cost"], 1e-3))
    dropped, used = [], 0.0
    for r in order:
        if r["ms_cost"] > 0.05 and used + max(r["err2_bought"], 0.0) <= budget:
            dropped.append(r["site"])
            used += max(r["err2_bought"], 0.0)
    keep = [s for s in ALL if (s[0] if a.quick else f"{s[0]}@{s[1]}") not in dropped]
    ez, ef, ms = measure(a.res, keep, g)
    print(f"selective (dropped {dropped}): latent {ez:.3e} frames {ef:.3e} step {ms:.1f} ms "
          f"= {ms / rows['none'][2]:.3f} x default, {ms / rows['all'][2]:.3f} x full parity", flush=True)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump({"res": a.res, "none": rows["none"], "all": rows["all"], "table": table, "dropped": dropped,
               "kept": [list(s) for s in keep], "selective": [ez, ef, ms]}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
