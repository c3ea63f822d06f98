"""CPU: the oracle (oracle/*.py restatement) against the golden fixtures captured from the REAL
reference by oracle/make_golden.py.  This is what pins the oracle (the reference itself
has no tests or golden vectors for this path)."""
import os

import numpy as np
import pytest
import torch

from oracle import ddim_ref, golden_recipe as gr, unet_ref
from oracle.unet_ref import _SD
from open_pandora_amd import synth, unet as U

GOLD = os.path.join(os.path.dirname(__file__), "golden")
RH_KW = dict(in_channels=8, out_channels=4, attention_resolutions=[4, 2, 1], num_res_blocks=2,
             channel_mult=[1, 2, 4, 4], dropout=0.1, num_head_channels=64, transformer_depth=1,
             context_dim=1024, use_linear=True, use_checkpoint=False, temporal_conv=True,
             temporal_attention=True, temporal_selfatt_only=True, use_relative_position=False,
             use_causal_attention=False, temporal_length=16, addition_attention=True,
             image_cross_attention=True, default_fs=24, fs_condition=True)


def rel(a, b):
    a, b = torch.as_tensor(a).float(), torch.as_tensor(b).float()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def load(name):
    return np.load(os.path.join(GOLD, name))


def test_schedule_tables_bit_exact():
    g = load("schedule.npz")
    for tag, base in (("512", 0.7), ("1024", 0.3)):
        t = ddim_ref.schedule_tables(base_scale=base)
        for k in ("alphas_cumprod", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "scale_arr"):
            assert np.array_equal(t[k].float().numpy(), g[f"{tag}/{k}"]), (tag, k)
        for S, eta in ((10, 0.0), (20, 1.0), (50, 1.0), (50, 0.0)):
            d = ddim_ref.ddim_tables(t, S, eta, "uniform_trailing")
            p = f"{tag}/S{S}_eta{eta:g}"
            assert np.array_equal(d["timesteps"], g[f"{p}/timesteps"])
            assert np.array_equal(d["alphas"].numpy(), g[f"{p}/alphas"])
            assert np.array_equal(d["alphas_prev"], g[f"{p}/alphas_prev"])
            assert np.array_equal(np.asarray(d["sigmas"]), g[f"{p}/sigmas"], equal_nan=True)
            assert np.array_equal(d["scale"].float().numpy(), g[f"{p}/scale"])
            assert np.array_equal(d["scale_prev"].float().numpy(), g[f"{p}/scale_prev"])


def test_timestep_embedding_bit_exact():
    g = load("schedule.npz")
    emb = unet_ref.timestep_embedding(torch.tensor(g["timestep_embedding/t"]), 320)
    assert rel(emb, g["timestep_embedding/emb320"]) < 1e-6
    # the product's own copy of the formula (it feeds the HIP embedding GEMVs)
    emb_p = U.timestep_embedding(torch.tensor(g["timestep_embedding/t"]), 320)
    assert torch.equal(emb_p, emb)


def _sd(mod):
    return synth.synth_state_dict(mod, seed=gr.WEIGHT_SEED)


def test_modules_against_reference_classes():
    g = load("modules.npz")
    mi = gr.module_inputs()
    x4, x5, tok, ctx, emb = mi["x4"], mi["x5"], mi["tok"], mi["ctx"], mi["emb"]
    cases = {
        "cross_attention_self": (U.CrossAttention(128, None, 2, 64),
                                 lambda sd: unet_ref.cross_attention(sd, tok)),
        "cross_attention_text_image": (U.CrossAttention(128, 1024, 2, 64, image_cross_attention=True),
                                       lambda sd: unet_ref.cross_attention(sd, tok, ctx)),
        "spatial_transformer": (U.SpatialTransformer(64, 1, 64, 1024, True, True),
                                lambda sd: unet_ref.spatial_transformer(sd, x4, ctx)),
        "temporal_transformer_linear": (U.TemporalTransformer(64, 1, 64, True),
                                        lambda sd: unet_ref.temporal_transformer(sd, x5)),
        "temporal_transformer_conv1d": (U.TemporalTransformer(64, 2, 64, False),
                                        lambda sd: unet_ref.temporal_transformer(sd, x5)),
        "res_block_64_128": (U.ResBlock(64, 256, 0.1, 128, True), lambda sd: unet_ref.res_block(sd, x4, emb, 1)),
        "res_block_64_64": (U.ResBlock(64, 256, 0.1, 64, True), lambda sd: unet_ref.res_block(sd, x4, emb, 1)),
        "downsample": (U.Downsample(64, 64), lambda sd: unet_ref._run_sequential(sd, "", x4, None, None, 1)),
        "upsample": (U.Upsample(64, 64), lambda sd: unet_ref._run_sequential(sd, "", x4, None, None, 1)),
    }
    for name, (mod, fn) in cases.items():
        prefix = "0." if name in ("downsample", "upsample") else ""
        sd = {prefix + k: v for k, v in _sd(mod).items()}
        got = fn(_SD(sd))
        assert got.shape == g[name].shape, name
        assert rel(got, g[name]) < 2e-5, (name, rel(got, g[name]))


def test_learnable_image_attention_scale_against_reference():
    """`image_cross_attention_scale_learnable` (attention.py:77-78,138-142; inference_256_v1.0.yaml:48): the oracle AND the
    product graph on the oracle's op table against the REAL CrossAttention / the REAL UNetModel built from the 256 yaml's own
    unet_config (oracle/make_golden.py --learnable); ignoring alpha must fail."""
    from oracle.ops_torch import TorchOps
    g = load("unet_small_learnable.npz")
    mi = gr.module_inputs()
    mod = U.CrossAttention(128, 1024, 2, 64, image_cross_attention=True, image_cross_attention_scale_learnable=True)
    sd = _sd(mod)
    assert "alpha" in sd and abs(float(sd["alpha"])) > 0.05
    want = g["cross_attention_text_image_learnable"]
    assert rel(unet_ref.cross_attention(_SD(sd), mi["tok"], mi["ctx"]), want) < 2e-5
    assert rel(unet_ref.cross_attention(_SD({k: v for k, v in sd.items() if k != "alpha"}), mi["tok"], mi["ctx"]), want) > 1e-2
    tag, mc, h, w, t, fs = gr.UNET_SMALL_CASES[0]
    m = U.UNetModel(**dict(RH_KW, model_channels=mc, **gr.UNET_256_OVERRIDES)).eval()
    sd = synth.synth_state_dict(m, seed=gr.WEIGHT_SEED)
    assert sum(k.endswith("attn2.alpha") for k in sd) == 16  # one per SpatialTransformer, none in the temporal blocks
    ins, _, _ = gr.sampler_inputs(h, w)
    x = torch.cat([ins["x_T"], ins["c_concat"]], 1)
    y = unet_ref.unet_forward(sd, x, torch.tensor([t]), ins["c_crossattn"], torch.tensor([fs]), model_channels=mc)
    assert rel(y, g["unet256/" + tag]) < 2e-5
    m.load_state_dict(sd)
    y = m.bind(TorchOps())(x, torch.tensor([t]), context=ins["c_crossattn"], fs=torch.tensor([fs]))
    assert rel(y, g["unet256/" + tag]) < 2e-5
    with torch.no_grad():  # an in-place edit of alpha re-packs the host scalar (packing.PackedWeights fingerprint)
        for k, prm in m.named_parameters():
            if k.endswith("attn2.alpha"):
                prm.zero_()
    assert rel(m(x, torch.tensor([t]), context=ins["c_crossattn"], fs=torch.tensor([fs])), g["unet256/" + tag]) > 1e-3


@pytest.mark.parametrize("tag,mc,h,w,t,fs", gr.UNET_SMALL_CASES)
def test_unet_small_against_reference(tag, mc, h, w, t, fs):
    g = load("unet_small.npz")
    kw = dict(RH_KW, model_channels=mc)
    sd = _sd(U.UNetModel(**kw))
    ins, _, _ = gr.sampler_inputs(h, w)
    x = torch.cat([ins["x_T"], ins["c_concat"]], 1)
    y = unet_ref.unet_forward(sd, x, torch.tensor([t]), ins["c_crossattn"], torch.tensor([fs]), model_channels=mc)
    assert rel(y, g[tag]) < 2e-5


def test_unet_features_adapter_against_reference():
    """`features_adapter` (openaimodel3d.py:584-596): the real module's output with four plug-in feature maps against (i) the
    oracle's restatement and (ii) the PRODUCT U-Net on the oracle's op table (the in-place add into the skip-concatenation
    buffer, inference path) and (iii) its differentiable training walk."""
    from oracle.ops_torch import TorchOps
    g = load("unet_small_adapter.npz")["mc64_8x8_t500"]
    kw = dict(RH_KW, model_channels=64)
    m = U.UNetModel(**kw).eval()
    sd = _sd(m)
    ins, _, _ = gr.sampler_inputs(8, 8)
    x = torch.cat([ins["x_T"], ins["c_concat"]], 1)
    fa = gr.adapter_features(64, 8, 8)
    t, fs = torch.tensor([500]), torch.tensor([15])
    y = unet_ref.unet_forward(sd, x, t, ins["c_crossattn"], fs, model_channels=64, features_adapter=fa)
    assert rel(y, g) < 2e-5
    assert rel(unet_ref.unet_forward(sd, x, t, ins["c_crossattn"], fs, model_channels=64), g) > 1e-2  # (the features matter)
    m.load_state_dict(sd)
    y2 = m.bind(TorchOps())(x, t, context=ins["c_crossattn"], features_adapter=fa, fs=fs)
    assert rel(y2, g) < 2e-5
    m.train()
    for p_ in m.parameters():
        p_.requires_grad_(True)
    m.dropout = 0.0
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    y3 = m(x, t, context=ins["c_crossattn"], features_adapter=fa, fs=fs)
    assert y3.requires_grad and rel(y3.detach(), g) < 5e-5
    with pytest.raises(IndexError):  # (a short list: the reference indexes past its end too, openaimodel3d.py:591)
        m.eval()(x, t, context=ins["c_crossattn"], features_adapter=fa[:3], fs=fs)
    with pytest.raises(AssertionError, match="Wrong features_adapter"):  # (a long one: the reference's assert, :594-595)
        m.eval()(x, t, context=ins["c_crossattn"], features_adapter=fa + fa[:1], fs=fs)


@pytest.mark.parametrize("S,eta,cfg", gr.DDIM_SMALL_CASES)
def test_ddim_small_against_reference(S, eta, cfg):
    g = load("ddim_small.npz")[f"S{S}_eta{eta:g}_cfg{cfg:g}"]
    kw = dict(RH_KW, model_channels=64)
    sd = _sd(U.UNetModel(**kw))
    ins, cond, uc = gr.sampler_inputs(8, 8)
    apply = lambda x, t, c, fs: unet_ref.unet_forward(sd, torch.cat([x] + c["c_concat"], 1), t,
                                                      torch.cat(c["c_crossattn"], 1), fs, model_channels=64)
    y, _ = ddim_ref.ddim_sample(apply, ddim_ref.schedule_tables(), ins["x_T"], cond, uc, S, eta, cfg,
                                noises=gr.noises(ins["x_T"].shape, S), fs=torch.tensor([15]))
    if np.isnan(g).any():  # S=10, eta=1, uniform_trailing: the reference itself returns NaN
        assert torch.isnan(y).any()
    else:
        assert rel(y, g) < 5e-5


@pytest.mark.parametrize("S,eta,cfg,gres", gr.DDIM_RESCALE_CASES)
def test_ddim_guidance_rescale_against_reference(S, eta, cfg, gres):
    """rescale_noise_cfg (utils_diffusion.py:147-158) on the sampler path (ddim.py:240-241)."""
    g = load("ddim_small_rescale.npz")[f"S{S}_eta{eta:g}_cfg{cfg:g}_gr{gres:g}"]
    kw = dict(RH_KW, model_channels=64)
    sd = _sd(U.UNetModel(**kw))
    ins, cond, uc = gr.sampler_inputs(8, 8)
    apply = lambda x, t, c, fs: unet_ref.unet_forward(sd, torch.cat([x] + c["c_concat"], 1), t,
                                                      torch.cat(c["c_crossattn"], 1), fs, model_channels=64)
    y, _ = ddim_ref.ddim_sample(apply, ddim_ref.schedule_tables(), ins["x_T"], cond, uc, S, eta, cfg,
                                noises=gr.noises(ins["x_T"].shape, S), fs=torch.tensor([15]), guidance_rescale=gres)
    assert np.isfinite(g).all() and rel(y, g) < 5e-5


@pytest.mark.parametrize("S,eta,cfg,cfg_img,gres", gr.DDIM_MULTICOND_CASES)
def test_ddim_multicond_against_reference(S, eta, cfg, cfg_img, gres):
    """The multi-condition sampler (ddim_multiplecond.py:214-236, three forwards per step): the oracle's restatement against
    the fixture produced by the reference's own sample / ddim_sampling / p_sample_ddim (oracle/make_golden.py
    gen_ddim_multicond: on top of the main sampler's make_schedule - the shipped one dies on the bf16 buffers)."""
    g = load("ddim_small_multicond.npz")[f"S{S}_eta{eta:g}_cfg{cfg:g}_img{cfg_img}_gr{gres:g}"]
    kw = dict(RH_KW, model_channels=64)
    sd = _sd(U.UNetModel(**kw))
    ins, cond, uc = gr.sampler_inputs(8, 8)
    apply = lambda x, t, c, fs: unet_ref.unet_forward(sd, torch.cat([x] + c["c_concat"], 1), t,
                                                      torch.cat(c["c_crossattn"], 1), fs, model_channels=64)
    y, _ = ddim_ref.ddim_sample(apply, ddim_ref.schedule_tables(), ins["x_T"], cond, uc, S, eta, cfg,
                                noises=gr.noises(ins["x_T"].shape, S), fs=torch.tensor([15]), guidance_rescale=gres,
                                uncond_img=gr.multicond_uc_img(ins, cond, uc), cfg_img=cfg_img)
    assert np.isfinite(g).all() and rel(y, g) < 5e-5
    # against the two-way sampler: with cfg_img == cfg (the default, ddim_multiplecond.py:219-220) the image-only forward
    # cancels algebraically - the same result up to f32 rounding - otherwise the third forward matters
    y2, _ = ddim_ref.ddim_sample(apply, ddim_ref.schedule_tables(), ins["x_T"], cond, uc, S, eta, cfg,
                                 noises=gr.noises(ins["x_T"].shape, S), fs=torch.tensor([15]), guidance_rescale=gres)
    assert (rel(y2, g) < 5e-5) if cfg_img is None else (rel(y2, g) > 1e-2)


def test_oracle_72x128_fixture_matches_the_real_reference():
    """Two committed digests of the same full-width forward at 16x72x128: one from the oracle (chunked attention),
    one from the REAL reference (eager attention called per frame, oracle/make_golden.py --full-72x128)."""
    a, b = load("unet_full_72x128.npz"), load("unet_full_72x128_oracle.npz")
    key = "cond/full" if "cond/full" in a and "cond/full" in b else "cond/slice"  # format 2 commits the latent whole
    assert key == "cond/full" or int(a["cond/stride"]) == int(b["cond/stride"])
    assert rel(b[key], a[key]) < 2e-5
    for k in ("std", "mean", "absmax") + (("col_rms", "row_rms") if "cond/col_rms" in a and "cond/col_rms" in b else ()):
        assert np.allclose(a[f"cond/{k}"], b[f"cond/{k}"], rtol=2e-5, atol=2e-5 * float(a["cond/std"]))


def _option_kwargs(which):
    opt = {}
    if which in ("score_corrector", "both"):
        opt.update(score_corrector=gr.RecipeCorrector(), corrector_kwargs=dict(gr.CORRECTOR_KWARGS))
    if which in ("noise_dropout", "both"):
        opt.update(noise_dropout=gr.NOISE_DROPOUT_P)
    return opt


@pytest.mark.parametrize("S,eta,cfg,which", gr.DDIM_OPTION_CASES)
def test_ddim_sampler_options_against_reference(S, eta, cfg, which, monkeypatch):
    """`score_corrector` / `corrector_kwargs` (ddim.py:248-250) and `noise_dropout` (:283-284) of DDIMSampler.sample against the
    REAL sampler on the 256 yaml's eps path (oracle/make_golden.py --ddim-options; the dropout mask is golden_recipe's seeded
    stand-in for torch.nn.functional.dropout on both sides): the oracle's restatement, and the PRODUCT sampler on the oracle's
    op table.  A v-parameterised model with a corrector fails the reference's own assertion."""
    from oracle.ops_torch import TorchOps
    from open_pandora_amd.ddim import DDIMSampler
    from open_pandora_amd.ddpm import LatentVisualDiffusion
    torch.set_num_threads(4)
    want = load("ddim_small_options.npz")[f"S{S}_eta{eta:g}_cfg{cfg:g}_{which}"]
    assert np.isfinite(want).all()
    kw = dict(RH_KW, model_channels=64, **gr.UNET_256_OVERRIDES)
    m = U.UNetModel(**kw).eval()
    sd = synth.synth_state_dict(m, seed=gr.WEIGHT_SEED)
    tables = ddim_ref.schedule_tables(zero_snr=False, dynamic_rescale=False)
    ins, cond, uc = gr.sampler_inputs(8, 8)
    apply = lambda x, t, c, fs: unet_ref.unet_forward(sd, torch.cat([x] + c["c_concat"], 1), t, torch.cat(c["c_crossattn"], 1), fs,
                                                      model_channels=64)
    ns = gr.noises(ins["x_T"].shape, S)
    monkeypatch.setattr(torch.nn.functional, "dropout", gr.RecipeDropout())
    y, _ = ddim_ref.ddim_sample(apply, tables, ins["x_T"], cond, uc, S, eta, cfg, noises=ns, fs=torch.tensor([3]),
                                parameterization="eps", **_option_kwargs(which))
    assert rel(y, want) < 5e-5
    m.load_state_dict(sd)
    pm = LatentVisualDiffusion(m.bind(TorchOps()), parameterization="eps", rescale_betas_zero_snr=False, use_dynamic_rescale=False,
                               image_size=(32, 32))
    monkeypatch.setattr(torch.nn.functional, "dropout", gr.RecipeDropout())
    y2, _ = DDIMSampler(pm).sample(S=S, batch_size=1, shape=(4, 16, 8, 8), conditioning=cond, verbose=False,
                                   unconditional_guidance_scale=cfg, unconditional_conditioning=uc, eta=eta, fs=torch.tensor([3]),
                                   timestep_spacing="uniform_trailing", x_T=ins["x_T"], noise_fn=lambda i, shape: ns[i],
                                   **_option_kwargs(which))
    assert rel(y2, want) < 5e-5
    if which == "score_corrector":  # the reference asserts the eps parameterisation (ddim.py:249); so does the product
        pv = LatentVisualDiffusion(m, parameterization="v", rescale_betas_zero_snr=False, use_dynamic_rescale=False,
                                   image_size=(32, 32))
        with pytest.raises(AssertionError):
            DDIMSampler(pv).sample(S=2, batch_size=1, shape=(4, 16, 8, 8), conditioning=cond, verbose=False, eta=0.0,
                                   fs=torch.tensor([3]), timestep_spacing="uniform_trailing", x_T=ins["x_T"],
                                   score_corrector=gr.RecipeCorrector())


@pytest.mark.parametrize("S,eta,cfg", gr.DDIM_EPS_CASES)
def test_ddim_eps_parameterisation_against_reference(S, eta, cfg):
    """The 256 yaml's sampler path (eps-prediction, no zero-terminal-SNR rescale, no dynamic rescale; ddim.py:243-246,265-266)
    against the REAL LatentVisualDiffusion + DDIMSampler built from that yaml (oracle/make_golden.py --eps): the oracle, and the
    PRODUCT sampler on the oracle's op table - whose fused update takes the eps step through transformed scalars."""
    from oracle.ops_torch import TorchOps
    from open_pandora_amd.ddim import DDIMSampler
    from open_pandora_amd.ddpm import LatentVisualDiffusion
    torch.set_num_threads(4)
    g = load("ddim_small_eps.npz")
    want = g[f"S{S}_eta{eta:g}_cfg{cfg:g}"]
    kw = dict(RH_KW, model_channels=64, **gr.UNET_256_OVERRIDES)
    m = U.UNetModel(**kw).eval()
    sd = synth.synth_state_dict(m, seed=gr.WEIGHT_SEED)
    tables = ddim_ref.schedule_tables(zero_snr=False, dynamic_rescale=False)
    assert np.array_equal(tables["alphas_cumprod"].float().numpy(), g["alphas_cumprod"])  # the un-rescaled schedule, bit for bit
    ins, cond, uc = gr.sampler_inputs(8, 8)
    apply = lambda x, t, c, fs: unet_ref.unet_forward(sd, torch.cat([x] + c["c_concat"], 1), t, torch.cat(c["c_crossattn"], 1), fs,
                                                      model_channels=64)
    ns = gr.noises(ins["x_T"].shape, S)
    y, _ = ddim_ref.ddim_sample(apply, tables, ins["x_T"], cond, uc, S, eta, cfg, noises=ns, fs=torch.tensor([3]),
                                parameterization="eps")
    assert rel(y, want) < 5e-5
    m.load_state_dict(sd)
    pm = LatentVisualDiffusion(m.bind(TorchOps()), parameterization="eps", rescale_betas_zero_snr=False, use_dynamic_rescale=False,
                               image_size=(32, 32))
    y2, _ = DDIMSampler(pm).sample(S=S, batch_size=1, shape=(4, 16, 8, 8), conditioning=cond, verbose=False,
                                   unconditional_guidance_scale=cfg, unconditional_conditioning=uc, eta=eta, fs=torch.tensor([3]),
                                   timestep_spacing="uniform_trailing", x_T=ins["x_T"], noise_fn=lambda i, shape: ns[i])
    assert rel(y2, want) < 5e-5
