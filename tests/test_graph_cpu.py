"""CPU: the product's host graph (channels-last UNetModel, LatentVisualDiffusion, DDIMSampler) run
with the oracle's per-op table (oracle/ops_torch.TorchOps, tests only) against the golden fixtures
from the real reference: checks graph wiring, weight packing, schedule numerics and the state-dict
contract without a GPU.  The HIP kernels themselves are covered by the -m gpu tests."""
import os

import numpy as np
import pytest
import torch

from oracle import golden_recipe as gr
from oracle.ops_torch import TorchOps
from open_pandora_amd import synth
from open_pandora_amd.ddim import DDIMSampler
from open_pandora_amd.ddpm import LatentVisualDiffusion
from open_pandora_amd.unet import UNetModel
from test_oracle_golden import RH_KW, load, rel


def small_unet(mc):
    m = UNetModel(**dict(RH_KW, model_channels=mc)).eval()
    m.load_state_dict(synth.synth_state_dict(m, seed=gr.WEIGHT_SEED))
    return m.bind(TorchOps())


def test_state_dict_contract_full_width():
    """1516 tensors, 1.4389 B parameters, reference key spelling."""
    with torch.device("meta"):
        m = UNetModel(**dict(RH_KW, model_channels=320))
    sd = m.state_dict()
    assert len(sd) == 1516
    assert sum(v.numel() for v in sd.values()) == 1438854980
    for k, shape in {"input_blocks.0.0.weight": (320, 8, 3, 3),
                     "init_attn.0.proj_in.weight": (512, 320, 1),
                     "input_blocks.1.0.temopral_conv.conv1.2.weight": (320, 320, 3, 1, 1),
                     "input_blocks.4.1.transformer_blocks.0.attn2.to_k_ip.weight": (640, 1024),
                     "output_blocks.5.3.conv.weight": (1280, 1280, 3, 3),
                     "middle_block.2.transformer_blocks.0.ff.net.0.proj.weight": (10240, 1280),
                     "out.2.weight": (4, 320, 3, 3), "fps_embedding.2.bias": (1280,)}.items():
        assert tuple(sd[k].shape) == shape, k


@pytest.mark.parametrize("tag,mc,h,w,t,fs", gr.UNET_SMALL_CASES)
def test_graph_unet_small(tag, mc, h, w, t, fs):
    g = load("unet_small.npz")[tag]
    m = small_unet(mc)
    ins, _, _ = gr.sampler_inputs(h, w)
    x = torch.cat([ins["x_T"], ins["c_concat"]], 1)
    y = m(x, torch.tensor([t]), context=ins["c_crossattn"], fs=torch.tensor([fs]))
    assert rel(y, g) < 2e-5


def test_product_schedule_tables_bit_exact():
    g = load("schedule.npz")
    for tag, base in (("512", 0.7), ("1024", 0.3)):
        pm = LatentVisualDiffusion(small_unet(64), base_scale=base)
        for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
                  "sqrt_one_minus_alphas_cumprod", "scale_arr"):
            assert np.array_equal(getattr(pm, k).float().numpy(), g[f"{tag}/{k}"]), (tag, k)
        smp = DDIMSampler(pm)
        for S, eta in ((10, 0.0), (20, 1.0), (50, 1.0), (50, 0.0)):
            smp.make_schedule(S, "uniform_trailing", eta, verbose=False)
            p = f"{tag}/S{S}_eta{eta:g}"
            assert np.array_equal(smp.ddim_timesteps, g[f"{p}/timesteps"])
            assert np.array_equal(smp.ddim_alphas.numpy(), g[f"{p}/alphas"])
            assert np.array_equal(smp.ddim_alphas_prev, g[f"{p}/alphas_prev"])
            assert np.array_equal(smp.ddim_sigmas.numpy(), g[f"{p}/sigmas"], equal_nan=True)
            assert np.array_equal(smp.ddim_scale_arr.float().numpy(), g[f"{p}/scale"])


@pytest.mark.parametrize("S,eta,cfg", gr.DDIM_SMALL_CASES)
def test_graph_ddim_small(S, eta, cfg):
    g = load("ddim_small.npz")[f"S{S}_eta{eta:g}_cfg{cfg:g}"]
    pm = LatentVisualDiffusion(small_unet(64))
    ins, cond, uc = gr.sampler_inputs(8, 8)
    ns = gr.noises(ins["x_T"].shape, S)
    y, inter = DDIMSampler(pm).sample(S=S, batch_size=1, shape=(4, 16, 8, 8), conditioning=cond, verbose=False,
                                      unconditional_guidance_scale=cfg, unconditional_conditioning=uc, eta=eta,
                                      fs=torch.tensor([15]), timestep_spacing="uniform_trailing", x_T=ins["x_T"],
                                      noise_fn=lambda i, shape: ns[i])
    if np.isnan(g).any():
        assert torch.isnan(y).any()
    else:
        assert rel(y, g) < 5e-5
    assert len(inter["x_inter"]) >= 2


@pytest.mark.parametrize("S,eta,cfg,gres", gr.DDIM_RESCALE_CASES)
def test_graph_ddim_guidance_rescale(S, eta, cfg, gres):
    g = load("ddim_small_rescale.npz")[f"S{S}_eta{eta:g}_cfg{cfg:g}_gr{gres:g}"]
    pm = LatentVisualDiffusion(small_unet(64))
    ins, cond, uc = gr.sampler_inputs(8, 8)
    ns = gr.noises(ins["x_T"].shape, S)
    y, _ = DDIMSampler(pm).sample(S=S, batch_size=1, shape=(4, 16, 8, 8), conditioning=cond, verbose=False,
                                  unconditional_guidance_scale=cfg, unconditional_conditioning=uc, eta=eta,
                                  fs=torch.tensor([15]), timestep_spacing="uniform_trailing", x_T=ins["x_T"],
                                  noise_fn=lambda i, shape: ns[i], guidance_rescale=gres)
    assert rel(y, g) < 5e-5
