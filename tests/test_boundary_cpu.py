"""CPU (build container): the drop-in boundary, proven against the reference's OWN plumbing.

* the product UNetModel is built by the reference's `instantiate_from_config` (DynamiCrafter/utils/utils.py:27-42)
  inside the reference's `LatentVisualDiffusion` (ddpm3d.py:1036), weights are loaded through that parent shell, and
  the REFERENCE DDIMSampler (ddim.py:66) drives it: result == the goldens captured from the all-reference run;
* the converse: the product DDIMSampler around the all-reference shell;
* reloading through the parent after a forward re-packs the kernel-side weights;
* the training seam fails loudly; the documented ctypes stub matches capi.py and the header."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import golden_recipe as gr, ref_harness as rh
from oracle.ops_torch import TorchOps
from open_pandora_amd import capi, synth
from open_pandora_amd.ddim import DDIMSampler
from open_pandora_amd.unet import UNetModel
from test_oracle_golden import RH_KW, load, rel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needs_ref = pytest.mark.skipif(not rh.available(), reason="needs the reference checkout (build container only)")
S, ETA, CFG = 5, 0.0, 4.0


def _sample_kw(ins, cond, uc):
    return dict(S=S, batch_size=1, shape=(4, 16, 8, 8), conditioning=cond, verbose=False,
                unconditional_guidance_scale=CFG, unconditional_conditioning=uc, eta=ETA, fs=torch.tensor([15]),
                timestep_spacing="uniform_trailing", x_T=ins["x_T"])


@needs_ref
def test_product_unet_behind_the_reference_shell_and_sampler():
    g = load("ddim_small.npz")[f"S{S}_eta{ETA:g}_cfg{CFG:g}"]
    ref = rh.reference_diffusion(dict(model_channels=64), unet_target="open_pandora_amd.unet.UNetModel")
    unet = ref.model.diffusion_model
    assert type(unet) is UNetModel  # built by instantiate_from_config from the yaml-style `target:` string
    unet.bind(TorchOps())
    ins, cond, uc = gr.sampler_inputs(8, 8)
    # a forward BEFORE the weights arrive (packs whatever the init left), then the reference loader's move:
    # load_state_dict on the PARENT (scripts/evaluation/inference.py:27-52)
    with torch.no_grad():
        ref.apply_model(ins["x_T"], torch.tensor([500]), cond, fs=torch.tensor([15]))
    sd = {"model.diffusion_model." + k: v for k, v in synth.synth_state_dict(unet, seed=gr.WEIGHT_SEED).items()}
    missing, unexpected = ref.load_state_dict(sd, strict=False)
    assert not unexpected and not [k for k in missing if k.startswith("model.diffusion_model.")]
    y, _ = rh.reference_sampler(ref).sample(**_sample_kw(ins, cond, uc))
    assert rel(y, g) < 5e-5


@needs_ref
def test_product_sampler_around_the_reference_shell():
    g = load("ddim_small.npz")[f"S{S}_eta{ETA:g}_cfg{CFG:g}"]
    ref = rh.reference_diffusion(dict(model_channels=64))
    ru = ref.model.diffusion_model
    ru.load_state_dict(synth.synth_state_dict(ru, seed=gr.WEIGHT_SEED))
    ins, cond, uc = gr.sampler_inputs(8, 8)
    y, _ = DDIMSampler(ref, ops=TorchOps()).sample(**_sample_kw(ins, cond, uc))
    assert rel(y, g) < 5e-5


def test_parent_reload_inplace_edit_and_cast_repack_the_weights():
    from open_pandora_amd.ddpm import LatentVisualDiffusion
    m = UNetModel(**dict(RH_KW, model_channels=64)).eval().bind(TorchOps())
    pm = LatentVisualDiffusion(m)
    ins, _, _ = gr.sampler_inputs(8, 8)
    x = torch.cat([ins["x_T"], ins["c_concat"]], 1)
    fwd = lambda mod: mod(x, torch.tensor([500]), context=ins["c_crossattn"], fs=torch.tensor([15]))
    sd1, sd2 = synth.synth_state_dict(m, seed=1), synth.synth_state_dict(m, seed=2)
    m.load_state_dict(sd1)
    y1, epoch = fwd(m), m._pack_epoch
    full = pm.state_dict()
    full.update({"model.diffusion_model." + k: v for k, v in sd2.items()})
    pm.load_state_dict(full)  # through the parent: the child's load_state_dict override never runs
    assert m._pack_epoch > epoch
    fresh = UNetModel(**dict(RH_KW, model_channels=64)).eval().bind(TorchOps())
    fresh.load_state_dict(sd2)
    y2 = fwd(m)
    assert rel(y2, fwd(fresh)) == 0.0 and rel(y2, y1) > 0.1
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(1.01)  # what an optimizer step does
    assert rel(fwd(m), y2) > 1e-4
    m.double()
    assert m._packed is None


def test_training_seam_fails_loudly():
    m = UNetModel(**dict(RH_KW, model_channels=64)).bind(TorchOps())
    m.load_state_dict(synth.synth_state_dict(m, seed=3))
    ins, _, _ = gr.sampler_inputs(8, 8)
    x = torch.cat([ins["x_T"], ins["c_concat"]], 1)
    m.train()
    with pytest.raises(RuntimeError, match="inference module"):
        m(x, torch.tensor([500]), context=ins["c_crossattn"], fs=torch.tensor([15]))
    with torch.no_grad():  # (what WorldModel.generate does, model.py:783)
        assert torch.isfinite(m(x, torch.tensor([500]), context=ins["c_crossattn"], fs=torch.tensor([15]))).all()


@pytest.mark.parametrize("L,tag", gr.UNET_CTX_CASES)
def test_context_without_per_frame_image_tokens(L, tag):
    """openaimodel3d.py:565-566 (`else: context.repeat_interleave`) against the real reference's output."""
    g = load("unet_small_ctx.npz")[tag]
    m = UNetModel(**dict(RH_KW, model_channels=64)).eval().bind(TorchOps())
    m.load_state_dict(synth.synth_state_dict(m, seed=gr.WEIGHT_SEED))
    ins, _, _ = gr.sampler_inputs(8, 8)
    x = torch.cat([ins["x_T"], ins["c_concat"]], 1)
    y = m(x, torch.tensor([500]), context=ins["c_crossattn"][:, :L], fs=torch.tensor([15]))
    assert rel(y, g) < 2e-5


def _header_arg_counts():
    src = open(os.path.join(ROOT, "include", "pandora_mi355x.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(pm_\w+)\s*\(([^;{]*?)\)\s*;", src):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def test_documented_stub_matches_capi_and_header():
    import subprocess
    import sys
    counts = _header_arg_counts()
    assert set(counts) == set(capi.SIGNATURES), set(counts) ^ set(capi.SIGNATURES)
    for name, (_, args) in capi.SIGNATURES.items():
        assert len(args) == counts[name], (name, len(args), counts[name])
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = doc.split("<!-- BEGIN GENERATED STUB -->\n```python\n")[1].split("\n```\n<!-- END GENERATED STUB -->")[0]
    gen = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_binding_stub.py")], capture_output=True,
                         text=True, check=True).stdout.rstrip("\n")
    assert stub == gen, "INTEGRATION.md section 2 is stale: regenerate with python tools/gen_binding_stub.py"
    call = stub.split("rc = lib.pm_gemm(")[1].split("    if rc")[0]
    call = re.sub(r"#.*", "", call)
    depth, n = 0, 1
    for ch in call:
        depth += ch in "([" 
        depth -= ch in ")]"
        n += (ch == "," and depth == 0)
    assert n == counts["pm_gemm"] == 19
