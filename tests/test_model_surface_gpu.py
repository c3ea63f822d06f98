"""GPU: the denoiser behind the reference's caller surfaces BY NAME (open_pandora_amd/model.py): `load_wm` -> `ChatWM` ->
`WorldModel.generate` -> `image_guided_synthesis` -> the product sampler and first stage on the HIP kernels, at 576x1024
(the resolution `dynamic_resize` fixes, model.py:507-513), reduced-width U-Net + reduced AutoencoderKL.

Two rounds through `ChatWM.generate_video_mutliround` (model.py:1094-1129) equal the same two rounds through
`wm.DiffusionRunner.generate_multiround` - the restatement that tests/test_frames_gpu.py pins against the CPU oracle - bit for
bit: the named surface adds the prompt / pixel plumbing and nothing numerical.  The single-round `generate_video` +
`generate_video_next_round2` session gives the same frames again.  (VERDICT r04 missing #4: "model.py keeps calling us" was
argued, not executed.)"""
import numpy as np
import pytest
import torch

from oracle import golden_recipe as gr
from open_pandora_amd import model as M, synth, wm
from open_pandora_amd.autoencoder import DDCONFIG, AutoencoderKL
from open_pandora_amd.ddpm import LatentVisualDiffusion
from open_pandora_amd.unet import UNetModel
from test_model_surface_cpu import _Tok, _image_processor
from test_oracle_golden import RH_KW

pytestmark = pytest.mark.gpu


def test_chatwm_drives_the_hip_denoiser_by_the_reference_names():
    from open_pandora_amd.ops_hip import HipOps
    ops = HipOps(torch.bfloat16, "cuda:0")
    m = UNetModel(**dict(RH_KW, model_channels=64)).eval()
    m.load_state_dict(synth.synth_state_dict(m, seed=gr.WEIGHT_SEED))
    pm = LatentVisualDiffusion(m.bind(ops), base_scale=0.3, image_size=(72, 128))
    ae = AutoencoderKL(ddconfig=dict(DDCONFIG, ch=32))
    ae.load_state_dict(synth.synth_state_dict(ae, seed=gr.WEIGHT_SEED))
    ae.bind(ops)
    ins, _, _ = gr.sampler_inputs(72, 128)
    text, img = ins["c_crossattn"][:, :77].cuda(), ins["c_crossattn"][:, 77:].cuda()
    uct, uci = ins["uc_crossattn"][:, :77].cuda(), ins["uc_crossattn"][:, 77:].cuda()
    zero_noise = lambda x, noise=None: ae.encode_first_stage(x, noise=torch.zeros(x.shape[0], 4, x.shape[2] // 8, x.shape[3] // 8,
                                                                                  device=x.device))
    embed = lambda im: img if float(im.float().abs().sum()) > 0 else uci
    runner = wm.DiffusionRunner(pm, embed, uct, zero_noise, ae.decode_first_stage)
    # the LLM side, injected: one (1, 77, 1024) conditioning per round, told apart by the length of the prompt
    n_ids = []

    def conditioner(input_ids, pixel_values, attention_mask, return_dict, oa, oh):
        n_ids.append(int(input_ids.shape[1]))
        return torch.cat([text * 0.5, text * (1.0 - 0.1 * (len(n_ids) - 1))])  # (generate keeps the LAST row)

    tok = _Tok()
    model, proc = M.load_wm("OpenSparseLLMs/Open-Pandora", runner=runner, get_diffusion_conditioning=conditioner, tokenizer=tok,
                            image_processor=_image_processor)
    assert isinstance(model, M.WorldModel) and model.diffusion_model is pm
    chat = M.ChatWM(model, proc)
    image = np.random.default_rng(3).integers(0, 255, (576, 1024, 3), dtype=np.uint8)
    x_T = ins["x_T"].cuda()
    chat.generate_kwargs["x_T"] = x_T  # (pins the draw: passes through **generate_kwargs to the sampler, as round_info does)
    r = chat.generate_video_mutliround(image, "move forward", 3, 15, 1, 4.0, 0.0, num_round=2, video_path="multi.mp4")
    stitched = chat.written["multi.mp4"]
    assert r[0] == "multi.mp4" and stitched.shape == (12 + 16, 576, 1024, 3) and len(n_ids) == 2 and n_ids[1] > n_ids[0]
    # the same two rounds through the runner's own driver
    px = chat.process_img(image)
    kw = dict(n_samples=1, ddim_steps=3, ddim_eta=0.0, unconditional_guidance_scale=4.0, fs=15, timestep_spacing="uniform_trailing",
              x_T=x_T)
    want = runner.generate_multiround([text, text * 0.9], px["diffusion_pixel_values"].cuda(), px["diffusion_cond_image"].cuda(), **kw)
    want8 = ((want[0, 0].float().clamp(-1, 1).cpu() + 1.) / 2. * 255.).permute(1, 2, 3, 0)
    assert torch.equal(stitched, want8)
    assert bool(torch.isfinite(stitched).all()) and float(stitched.std()) > 5.0
    # the interactive session: round 1, then "Action 2" - the same frames once more
    chat2 = M.ChatWM(model, proc)
    chat2.generate_kwargs["x_T"] = x_T
    n_ids.clear()
    chat2.generate_video(image, "move forward", 3, 15, 1, 4.0, 0.0)
    chat2.generate_video_next_round2("move forward", 3, 15, 1, 4.0, 0.0)
    assert torch.equal(chat2.written[chat2.video_path[0]], stitched)
    print(f"\n[parity] ChatWM.generate_video_mutliround (2 rounds, 576x1024, reduced width, bf16) == DiffusionRunner.generate_multiround "
          f"bit for bit; frame std {float(stitched.std()):.1f} / 255")
    runner.sampler.close()
