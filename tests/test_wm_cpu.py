"""CPU: the caller surfaces (open_pandora_amd/wm.py: image_guided_synthesis / generate / multi-round
helpers) drive the same sampler as a direct DDIMSampler call (op table: oracle TorchOps, tests only)."""
import torch

from oracle import golden_recipe as gr
from oracle.ops_torch import TorchOps
from open_pandora_amd import synth, wm
from open_pandora_amd.ddim import DDIMSampler
from open_pandora_amd.ddpm import LatentVisualDiffusion
from open_pandora_amd.unet import UNetModel
from test_oracle_golden import RH_KW


def test_generate_matches_direct_sampling_and_layout():
    torch.set_num_threads(4)
    m = UNetModel(**dict(RH_KW, model_channels=64)).eval()
    m.load_state_dict(synth.synth_state_dict(m, seed=3))
    pm = LatentVisualDiffusion(m.bind(TorchOps()))
    ins, cond, uc = gr.sampler_inputs(8, 8)
    text, img = ins["c_crossattn"][:, :77], ins["c_crossattn"][:, 77:]
    uct, uci = ins["uc_crossattn"][:, :77], ins["uc_crossattn"][:, 77:]
    # stand-ins for the out-of-scope encoders: AE = 8x average pool to 4 channels, image tower = lookup
    enc = lambda x: torch.nn.functional.avg_pool2d(torch.cat([x, x[:, :1]], 1), 8) * 0.18215
    runner = wm.DiffusionRunner(pm, lambda im: img if im.abs().sum() > 0 else uci, uct, enc)
    frames = torch.randn(3, 1, 64, 64, generator=torch.Generator().manual_seed(0))
    cond_img = torch.ones(1, 3, 64, 64)
    torch.manual_seed(5)
    out = runner.generate(text, frames, cond_img, n_samples=2, ddim_steps=2, ddim_eta=0.0)
    assert out.shape == (1, 2, 4, 16, 8, 8)
    z = wm.get_latent_z(enc, frames[None])
    assert z.shape == (1, 4, 16, 8, 8) and torch.equal(z[:, :, 0], z[:, :, 5])  # 1 frame tiled x16
    torch.manual_seed(5)
    want, _ = DDIMSampler(pm).sample(S=2, batch_size=1, shape=(4, 16, 8, 8), verbose=False, eta=0.0,
                                     conditioning={"c_crossattn": [torch.cat([text, img], 1)], "c_concat": [z]},
                                     unconditional_guidance_scale=4,
                                     unconditional_conditioning={"c_crossattn": [torch.cat([uct, uci], 1)], "c_concat": [z]},
                                     fs=torch.tensor([15]), timestep_spacing="uniform_trailing")
    assert torch.allclose(out[0, 0], want[0], atol=1e-5)
    # 4 conditioning frames are tiled x4; multi-round stitching keeps 12 + ... + 16 frames
    z4 = wm.get_latent_z(enc, torch.randn(1, 3, 4, 64, 64))
    assert z4.shape[2] == 16 and torch.equal(z4[:, :, 0], z4[:, :, 4])
    vids = [torch.randn(1, 1, 3, 16, 8, 8) for _ in range(5)]
    assert wm.DiffusionRunner.stitch_rounds(vids).shape[3] == 64
    assert wm.DiffusionRunner.next_round_condition(vids[0][0]).shape[2] == 4
