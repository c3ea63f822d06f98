"""CPU (build container): the drop-in boundary, proven against the reference's OWN plumbing.

* the product UNetModel is built by the reference's `instantiate_from_config` (DynamiCrafter/utils/utils.py:27-42)
  inside the reference's `LatentVisualDiffusion` (ddpm3d.py:1036), weights are loaded through that parent shell, and
  the REFERENCE DDIMSampler (ddim.py:66) drives it: result == the goldens captured from the all-reference run;
* the converse: the product DDIMSampler around the all-reference shell;
* reloading through the parent after a forward re-packs the kernel-side weights;
* the training seam: gradients equal the reference module's, an optimizer step re-packs the kernel-side weights;
  the documented ctypes stub matches capi.py and the header."""
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
def test_shipped_256_yaml_unet_config_instantiates_the_product_unet():
    """configs/inference_256_v1.0.yaml (the config BASELINE.md's 256x256 A100 number is quoted on) sets
    `image_cross_attention_scale_learnable: true` (:48): its unet_config, verbatim, through the reference's own
    instantiate_from_config with the product class as `target:` - full width on the meta device (key set = the reference
    module's, the 16 `alpha` keys included), reduced width with weights against the real module's output."""
    import yaml
    rh._install_shims()
    from utils.utils import instantiate_from_config
    from lvdm.modules.networks.openaimodel3d import UNetModel as RefUNet
    with open(os.path.join(rh.REFERENCE_ROOT, "DynamiCrafter", "configs", "inference_256_v1.0.yaml")) as f:
        cfg = yaml.safe_load(f)["model"]["params"]["unet_config"]
    with torch.device("meta"):
        prod = instantiate_from_config(dict(cfg, target="open_pandora_amd.unet.UNetModel"))
        ref = RefUNet(**cfg["params"])
    assert type(prod) is UNetModel
    ps, rs = prod.state_dict(), ref.state_dict()
    assert set(ps) == set(rs) and all(ps[k].shape == rs[k].shape for k in rs)
    assert sum(k.endswith(".alpha") for k in ps) == 16
    small = instantiate_from_config(dict(target="open_pandora_amd.unet.UNetModel",
                                         params=dict(cfg["params"], model_channels=64))).eval()
    small.load_state_dict(synth.synth_state_dict(small, seed=gr.WEIGHT_SEED))
    tag, mc, h, w, t, fs = gr.UNET_SMALL_CASES[0]
    ins, _, _ = gr.sampler_inputs(h, w)
    y = small.bind(TorchOps())(torch.cat([ins["x_T"], ins["c_concat"]], 1), torch.tensor([t]), context=ins["c_crossattn"],
                               fs=torch.tensor([fs]))
    assert rel(y, load("unet_small_learnable.npz")["unet256/" + tag]) < 2e-5


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


def _no_dropout(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0  # (TemporalConvBlock hard-codes p = 0.1, openaimodel3d.py:266-273: not comparable draw by draw)
    return m


def _train_inputs(batch):
    ins, _, _ = gr.sampler_inputs(8, 8)
    x = torch.cat([ins["x_T"], ins["c_concat"]], 1)
    ctx = ins["c_crossattn"]
    if batch > 1:  # a second, different clip
        g = torch.Generator().manual_seed(77)
        x = torch.cat([x, torch.randn(x.shape, generator=g)], 0)
        ctx = torch.cat([ctx, torch.randn(ctx.shape, generator=g)], 0)
    t = torch.tensor([500, 120][:batch])
    fs = torch.tensor([15, 7][:batch])
    return x, t, ctx, fs


@needs_ref
@pytest.mark.parametrize("batch", [1, 2])
def test_training_seam_gradients_equal_the_reference_module(batch):
    """SURVEY §8(b) / model.py:926-942: in training mode with autograd on, the product UNetModel builds an autograd graph
    over its OWN parameters; loss and every parameter gradient equal the reference module's (f32, dropout off)."""
    ref = _no_dropout(rh.reference_unet(model_channels=64)).train()
    m = _no_dropout(UNetModel(**dict(RH_KW, model_channels=64))).train()  # note: NOT bound to any op table
    sd = synth.synth_state_dict(m, seed=3)
    ref.load_state_dict(sd)
    m.load_state_dict(sd)
    x, t, ctx, fs = _train_inputs(batch)
    target = torch.randn(batch, 4, 16, 8, 8, generator=torch.Generator().manual_seed(5))
    grads = []
    for mod in (ref, m):
        mod.zero_grad()
        y = mod(x, t, context=ctx, fs=fs)
        loss = torch.nn.functional.mse_loss(y, target)  # (the 'l2' loss of ddpm3d.py:258-273 on the v-target)
        loss.backward()
        grads.append((y.detach(), loss.item(), {k: p.grad.clone() for k, p in mod.named_parameters()}))
    (y0, l0, g0), (y1, l1, g1) = grads
    assert rel(y1, y0) < 1e-5 and abs(l1 - l0) < 1e-5 * abs(l0)
    assert g0.keys() == g1.keys()
    worst = max(rel(g1[k], g0[k]) for k in g0 if g0[k].norm() > 0)
    assert worst < 1e-5, worst
    assert all(g1[k].abs().max() > 0 for k in g1)  # every parameter is reached (incl. the zero-init modules, overwritten by synth)


def test_training_seam_trains_and_inference_stays_on_the_op_table():
    """One AdamW step through the seam (configure_optimizers, model.py:951-962) lowers the loss, re-packs the kernel-side
    weights, and eval() / no_grad calls still run on the bound op table (or raise when none is bound)."""
    m = _no_dropout(UNetModel(**dict(RH_KW, model_channels=64)))
    m.load_state_dict(synth.synth_state_dict(m, seed=3))
    x, t, ctx, fs = _train_inputs(1)
    m.eval()
    with pytest.raises(RuntimeError, match="bind"):  # inference without an op table: loud, no eager fallback
        m(x, t, context=ctx, fs=fs)
    m.bind(TorchOps())
    y_inf = m(x, t, context=ctx, fs=fs)
    assert not y_inf.requires_grad
    m.train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
    target = torch.zeros(1, 4, 16, 8, 8)
    y = m(x, t, context=ctx, fs=fs)
    assert y.requires_grad and rel(y.detach(), y_inf) < 2e-5  # same function as the inference graph
    loss0 = torch.nn.functional.mse_loss(y, target)
    loss0.backward()
    epoch = m._pack_epoch
    opt.step()
    with torch.no_grad():  # (what WorldModel.generate does, model.py:783): op table again, with the updated weights
        y2 = m(x, t, context=ctx, fs=fs)
    assert m._pack_epoch > epoch
    assert torch.nn.functional.mse_loss(y2, target) < loss0


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


@needs_ref
def test_training_step_surface_equals_the_reference_shell():
    """WorldModel.training_step -> `self.diffusion_model(x, c, fs=...)` (model.py:926-942 -> ddpm3d.py:700-705,741-797): the
    PRODUCT shell's forward / p_losses (timestep draw, dynamic rescale of x, q_sample, v-target, l2 mean, loss_dict keys) against
    the REAL LatentVisualDiffusion holding the same U-Net weights, same RNG seed -> same t and noise: loss and loss_dict equal to
    1e-6, gradients flow to the U-Net parameters; configure_optimizers builds the reference's AdamW parameter list."""
    from open_pandora_amd import model as M
    from open_pandora_amd.ddpm import LatentVisualDiffusion
    ref = _no_dropout(rh.reference_diffusion(dict(model_channels=64))).train()
    ru = ref.model.diffusion_model
    sd = synth.synth_state_dict(ru, seed=gr.WEIGHT_SEED)
    ru.load_state_dict(sd)
    m = _no_dropout(UNetModel(**dict(RH_KW, model_channels=64)))
    m.load_state_dict(sd)
    pm = LatentVisualDiffusion(m).train()
    ins, cond, _ = gr.sampler_inputs(8, 8)
    x0 = ins["x_T"] * 0.18215
    torch.manual_seed(1234)
    want, want_d = ref(x0, cond, fs=torch.tensor([15]))
    runner = type("R", (), {"diffusion_model": pm, "encode_first_stage": None})()
    wmod = M.WorldModel(runner, lambda *a: None, config=type("C", (), {"learning_rate": 1e-5})())
    wmod.get_batch_input = lambda random_uncond=False, **batch: (batch["x"], batch["c"], batch["fs"])
    torch.manual_seed(1234)
    got = wmod.training_step({"x": x0, "c": cond, "fs": torch.tensor([15.0])}, 0)
    assert got.requires_grad and abs(float(got.detach()) - float(want.detach())) <= 1e-6 * abs(float(want.detach()))
    assert set(wmod.last_loss_dict) == set(want_d) == {"train/loss_simple", "train/loss"}
    assert all(abs(float(wmod.last_loss_dict[k].detach()) - float(want_d[k].detach())) <= 1e-6 * abs(float(want_d[k].detach()))
               for k in want_d)
    got.backward()
    assert sum(p.grad is not None and float(p.grad.abs().sum()) > 0 for p in m.parameters()) > 1400
    opt = wmod.configure_optimizers()
    assert isinstance(opt, torch.optim.AdamW) and len(opt.param_groups[0]["params"]) == len(list(m.parameters())) == 1516
    assert opt.param_groups[0]["lr"] == 1e-5
