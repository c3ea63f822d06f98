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


def test_multiround_driver_conditions_each_round_on_the_previous_frames():
    """generate_video_mutliround (model.py:1094-1129) restated: R rounds, round r+1 conditioned on the last 4
    frames of round r through the 8-bit PIL round trip, the first image kept as `diffusion_cond_image`,
    stitched to 12 (R-1) + 16 frames."""
    torch.set_num_threads(4)
    m = UNetModel(**dict(RH_KW, model_channels=64)).eval()
    m.load_state_dict(synth.synth_state_dict(m, seed=3))
    pm = LatentVisualDiffusion(m.bind(TorchOps()))
    ins, _, _ = gr.sampler_inputs(8, 8)
    text, img = ins["c_crossattn"][:, :77], ins["c_crossattn"][:, 77:]
    uct, uci = ins["uc_crossattn"][:, :77], ins["uc_crossattn"][:, 77:]
    enc = lambda x: torch.nn.functional.avg_pool2d(torch.cat([x, x[:, :1]], 1), 8) * 0.18215
    dec = lambda z: torch.nn.functional.interpolate(z[0, :3].permute(1, 0, 2, 3), scale_factor=8).permute(1, 0, 2, 3)[None] * 3
    seen = []
    enc_log = lambda x: (seen.append(x.clone()), enc(x))[1]
    runner = wm.DiffusionRunner(pm, lambda im: img if im.abs().sum() > 0 else uci, uct, enc_log, dec)
    frames = torch.randn(3, 1, 64, 64, generator=torch.Generator().manual_seed(0)).clamp(-1, 1)
    torch.manual_seed(5)
    texts = [text, text * 0.9, text * 1.1]
    out = runner.generate_multiround(texts, frames, torch.ones(1, 3, 64, 64), n_samples=1, ddim_steps=2, ddim_eta=0.0)
    assert out.shape == (1, 1, 3, 12 * 2 + 16, 64, 64)
    # rounds 2 and 3 encoded 4 conditioning frames each = the previous round's last 4 frames, 8-bit quantised
    assert [t.shape[0] for t in seen] == [1, 4, 4]
    r1_last4 = out[0, 0][:, 8:12]  # (frames 12-15 of round 1 are cut by the stitching: recompute from round 2's input)
    assert seen[1].abs().max() <= 1.0 and torch.allclose(seen[1] * 127.5 + 127.5, (seen[1] * 127.5 + 127.5).round(), atol=1e-3)
    assert r1_last4.shape[1] == 4


def test_multiple_cond_cfg_selects_the_three_forward_sampler():
    """model.py:705,737-743 + ddim_multiplecond.py:214-236: `multiple_cond_cfg=True` through the caller surface builds the
    image-tokens-with-empty-text condition set and runs THREE forwards per step; the product sampler (on the oracle's op
    table here) reproduces the fixture of the reference's own sampling code (oracle/make_golden.py gen_ddim_multicond)."""
    from test_oracle_golden import load, rel
    torch.set_num_threads(4)
    S, eta, cfg, cfg_img, gres = gr.DDIM_MULTICOND_CASES[0]
    g = load("ddim_small_multicond.npz")[f"S{S}_eta{eta:g}_cfg{cfg:g}_img{cfg_img}_gr{gres:g}"]
    m = UNetModel(**dict(RH_KW, model_channels=64)).eval()
    m.load_state_dict(synth.synth_state_dict(m, seed=gr.WEIGHT_SEED))
    pm = LatentVisualDiffusion(m.bind(TorchOps()))
    calls = []
    inner = pm.apply_model
    pm.apply_model = lambda x, t, c, **kw: (calls.append(c["c_crossattn"][0]), inner(x, t, c, **kw))[1]
    ins, cond, uc = gr.sampler_inputs(8, 8)
    text, img = ins["c_crossattn"][:, :77], ins["c_crossattn"][:, 77:]
    uct, uci = ins["uc_crossattn"][:, :77], ins["uc_crossattn"][:, 77:]
    y = wm._synthesize(pm, text, img, uct, uci, ins["c_concat"], (1, 4, 16, 8, 8), n_samples=1, ddim_steps=S, ddim_eta=eta,
                       unconditional_guidance_scale=cfg, cfg_img=cfg_img, fs=15, multiple_cond_cfg=True,
                       timestep_spacing="uniform_trailing", guidance_rescale=gres, x_T=ins["x_T"])[:, 0]
    assert rel(y, g) < 5e-5
    assert len(calls) == 3 * S
    want_ui = gr.multicond_uc_img(ins, cond, uc)["c_crossattn"][0]
    assert torch.equal(calls[2], want_ui) and torch.equal(calls[0], cond["c_crossattn"][0]) and torch.equal(calls[1], uc["c_crossattn"][0])
    # without the flag: the two-forward sampler, as before
    calls.clear()
    wm._synthesize(pm, text, img, uct, uci, ins["c_concat"], (1, 4, 16, 8, 8), n_samples=1, ddim_steps=2, ddim_eta=0.0,
                   unconditional_guidance_scale=cfg, fs=15, timestep_spacing="uniform_trailing", x_T=ins["x_T"])
    assert len(calls) == 2 * 2


def test_multicond_twin_is_cached_configured_like_the_callers_sampler_and_closed_with_it():
    """with `multiple_cond_cfg=True` and a plain DDIMSampler passed in (what DiffusionRunner does), ONE
    DDIMSamplerMultiCond is built from that sampler's op table / graph switch / CFG-pair object and cached on it - not a fresh
    default-configured sampler (fresh warm-up + capture, leaked pool, multi-rank guard bypassed) per call."""
    from open_pandora_amd.ddim import DDIMSampler, DDIMSamplerMultiCond
    m = UNetModel(**dict(RH_KW, model_channels=64)).eval()
    pm = LatentVisualDiffusion(m.bind(TorchOps()))
    ops2, pair = TorchOps(), object()
    primary = DDIMSampler(pm, use_graph=False, cfg_parallel=pair, ops=ops2)
    twin, tmp = wm._multicond_sampler(pm, primary)
    assert isinstance(twin, DDIMSamplerMultiCond) and tmp is None
    assert twin.use_graph is False and twin.cfg_parallel is pair and twin._ops_override is ops2 and twin.model is pm
    assert wm._multicond_sampler(pm, primary)[0] is twin                      # cached: one per sampler
    assert wm._multicond_sampler(pm, twin) == (twin, None)                    # already the right class: used as is
    t2, tmp2 = wm._multicond_sampler(pm, None)                                # no sampler: a temporary, closed by the caller
    assert tmp2 is t2 and t2 is not twin
    closed = []
    twin.close = lambda *a, **k: closed.append("twin")
    primary.close()
    assert closed == ["twin"]
    # a CFG-pair deployment reaches the multi-condition sampler's own guard instead of sampling per-rank noise
    assert twin._multi_rank()
