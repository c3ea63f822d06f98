"""Generate the committed golden fixtures under tests/golden/ by running the REAL reference
(/root/reference, imported through oracle/ref_harness.py) on seeded synthetic weights and inputs.

Run in the build container only:   python -m oracle.make_golden [--full] [--traj]
  default : schedule tables, module-level and reduced-width U-Net / DDIM fixtures (seconds)
  --full  : one full-width (1.44 B parameter) U-Net forward at 40x64, cond + uncond (minutes)
  --traj  : full-width 10-step eta=0 CFG trajectory at 40x64 = BASELINE config 1 (~20-30 min)
  --frames / --frames-full "10:0,50:1" : end-to-end FRAMES (sampler -> decode_first_stage), reduced / full width
Fixtures hold inputs' seeds and expected outputs only (data, no reference source).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_harness as rh  # noqa: E402
from open_pandora_amd import synth  # noqa: E402
from oracle import golden_recipe as gr  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
WEIGHT_SEED, INPUT_SEED, NOISE_SEED = gr.WEIGHT_SEED, gr.INPUT_SEED, gr.NOISE_SEED
SUFFIX = ""  # --heldout-seeds: appended to the fixture names of gen_frames_full


def digest(t, n=4096):
    """Fixture record of a full-size output (gr.make_digest, format 2: the whole tensor when it is a latent, else a slice at a
    prime stride coprime with every axis; global moments; per-column / per-row profiles)."""
    return gr.make_digest(t, n)


def save(name, **arrays):
    os.makedirs(GOLD, exist_ok=True)
    path = os.path.join(GOLD, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def gen_schedule():
    out = {}
    for base_scale, tag in ((0.7, "512"), (0.3, "1024")):
        m = rh.reference_diffusion(dict(model_channels=32, num_head_channels=32), base_scale=base_scale)
        for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
                  "sqrt_one_minus_alphas_cumprod", "scale_arr"):
            out[f"{tag}/{k}"] = getattr(m, k).float().numpy()
        smp = rh.reference_sampler(m)
        for S, eta in ((10, 0.0), (20, 1.0), (50, 1.0), (50, 0.0)):
            smp.make_schedule(S, "uniform_trailing", eta, verbose=False)
            p = f"{tag}/S{S}_eta{eta:g}"
            out[f"{p}/timesteps"] = np.asarray(smp.ddim_timesteps)
            out[f"{p}/alphas"] = smp.ddim_alphas.numpy()
            out[f"{p}/alphas_prev"] = np.asarray(smp.ddim_alphas_prev)
            out[f"{p}/sigmas"] = smp.ddim_sigmas.numpy()
            out[f"{p}/scale"] = smp.ddim_scale_arr.float().numpy()
            out[f"{p}/scale_prev"] = smp.ddim_scale_arr_prev.float().numpy()
    from lvdm.models.utils_diffusion import timestep_embedding
    out["timestep_embedding/t"] = np.asarray([0, 1, 99, 500, 999, 15, 24])
    out["timestep_embedding/emb320"] = timestep_embedding(torch.tensor(out["timestep_embedding/t"]), 320).numpy()
    save("schedule.npz", **out)


def gen_modules():
    """Reference module classes at C=64 on seeded weights: inputs are re-synthesised from the seed."""
    rh._install_shims()
    from lvdm.modules.attention import CrossAttention, SpatialTransformer, TemporalTransformer
    from lvdm.modules.networks.openaimodel3d import ResBlock, Downsample, Upsample
    out = {}

    def run(tag, mod, fn):
        sd = synth.synth_state_dict(mod, seed=WEIGHT_SEED)
        mod.load_state_dict(sd)
        with torch.no_grad():
            out[tag] = fn(mod.eval()).numpy()

    mi = gr.module_inputs()
    x4, x5, tok, ctx, emb = mi["x4"], mi["x5"], mi["tok"], mi["ctx"], mi["emb"]  # (b t) c h w | b c t h w | tokens
    run("cross_attention_self", CrossAttention(128, None, heads=2, dim_head=64), lambda m: m(tok))
    run("cross_attention_text_image",
        CrossAttention(128, 1024, heads=2, dim_head=64, image_cross_attention=True, video_length=16),
        lambda m: m(tok, context=ctx))
    run("spatial_transformer",
        SpatialTransformer(64, 1, 64, context_dim=1024, use_checkpoint=False, use_linear=True, video_length=16,
                           image_cross_attention=True), lambda m: m(x4, ctx))
    run("temporal_transformer_linear",
        TemporalTransformer(64, 1, 64, use_checkpoint=False, use_linear=True, only_self_att=True,
                            relative_position=False, temporal_length=16), lambda m: m(x5))
    run("temporal_transformer_conv1d",
        TemporalTransformer(64, 2, 64, use_checkpoint=False, use_linear=False, only_self_att=True,
                            relative_position=False, temporal_length=16), lambda m: m(x5))
    run("res_block_64_128", ResBlock(64, 256, 0.1, out_channels=128, dims=2, use_temporal_conv=True),
        lambda m: m(x4, emb, batch_size=1))
    run("res_block_64_64", ResBlock(64, 256, 0.1, out_channels=64, dims=2, use_temporal_conv=True),
        lambda m: m(x4, emb, batch_size=1))
    run("downsample", Downsample(64, True, dims=2, out_channels=64), lambda m: m(x4))
    run("upsample", Upsample(64, True, dims=2, out_channels=64), lambda m: m(x4))
    save("modules.npz", **out)


def _small_setup(mc=64, h=8, w=8, T=16):
    return gr.sampler_inputs(h, w, T)


def gen_unet_small():
    out = {}
    for tag, mc, h, w, t, fs in gr.UNET_SMALL_CASES:
        ref = rh.reference_unet(model_channels=mc)
        ref.load_state_dict(synth.synth_state_dict(ref, seed=WEIGHT_SEED))
        ins, _, _ = _small_setup(mc, h, w)
        x = torch.cat([ins["x_T"], ins["c_concat"]], 1)
        with torch.no_grad():
            y = ref(x, torch.tensor([t]), context=ins["c_crossattn"], fs=torch.tensor([fs]))
        out[tag] = y.numpy()
    save("unet_small.npz", **out)


def gen_learnable():
    """`image_cross_attention_scale_learnable` (attention.py:77-78,138-142; set by inference_256_v1.0.yaml:48): the real
    CrossAttention with the per-block `alpha`, and the real UNetModel built from the 256 yaml's own unet_config (read from the
    reference checkout, reduced to 64 base channels), seeded alphas well away from 0."""
    import yaml
    rh._install_shims()
    from lvdm.modules.attention import CrossAttention
    out = {}
    mi = gr.module_inputs()
    m = CrossAttention(128, 1024, heads=2, dim_head=64, image_cross_attention=True, video_length=16,
                       image_cross_attention_scale_learnable=True).eval()
    sd = synth.synth_state_dict(m, seed=WEIGHT_SEED)
    assert "alpha" in sd and abs(float(sd["alpha"])) > 0.05
    m.load_state_dict(sd)
    with torch.no_grad():
        out["cross_attention_text_image_learnable"] = m(mi["tok"], context=mi["ctx"]).numpy()
    with open(os.path.join(rh.REFERENCE_ROOT, "DynamiCrafter", "configs", "inference_256_v1.0.yaml")) as f:
        kw = yaml.safe_load(f)["model"]["params"]["unet_config"]["params"]
    assert all(kw[k] == v for k, v in gr.UNET_256_OVERRIDES.items())
    tag, mc, h, w, t, fs = gr.UNET_SMALL_CASES[0]
    ref = rh.reference_unet(**dict(kw, model_channels=mc, use_checkpoint=False))
    ref.load_state_dict(synth.synth_state_dict(ref, seed=WEIGHT_SEED))
    ins, _, _ = _small_setup(mc, h, w)
    x = torch.cat([ins["x_T"], ins["c_concat"]], 1)
    with torch.no_grad():
        out["unet256/" + tag] = ref(x, torch.tensor([t]), context=ins["c_crossattn"], fs=torch.tensor([fs])).numpy()
    save("unet_small_learnable.npz", **out)


def gen_ddim_eps():
    """The 256 model's sampler path: `parameterization == "eps"` (ddim.py:243-246,265-266: e_t = model output, pred_x0 =
    (x - sqrt(1 - a) e_t) / sqrt(a)), schedule without zero-terminal-SNR rescale, no dynamic rescale - the REAL
    LatentVisualDiffusion built with the 256 yaml's shell parameters and its unet_config at reduced width, the REAL DDIMSampler."""
    import yaml
    rh._install_shims()
    import lvdm.models.samplers.ddim as refddim
    with open(os.path.join(rh.REFERENCE_ROOT, "DynamiCrafter", "configs", "inference_256_v1.0.yaml")) as f:
        kw = yaml.safe_load(f)["model"]["params"]["unet_config"]["params"]
    m = rh.reference_diffusion(dict(kw, model_channels=64, use_checkpoint=False), shell=gr.SHELL_256)
    assert m.parameterization == "eps" and not m.use_dynamic_rescale
    m.model.diffusion_model.load_state_dict(synth.synth_state_dict(m.model.diffusion_model, seed=WEIGHT_SEED))
    ins, cond, uc = _small_setup()
    out = {"alphas_cumprod": m.alphas_cumprod.float().numpy()}
    for S, eta, cfg in gr.DDIM_EPS_CASES:
        noises = iter(gr.noises(ins["x_T"].shape, S))
        refddim.noise_like = lambda shape, device, repeat=False: next(noises)
        smp = rh.reference_sampler(m)
        y, _ = smp.sample(S=S, batch_size=1, shape=(4, 16, 8, 8), conditioning=cond, verbose=False,
                          unconditional_guidance_scale=cfg, unconditional_conditioning=uc, eta=eta, fs=torch.tensor([3]),
                          timestep_spacing="uniform_trailing", x_T=ins["x_T"])
        out[f"S{S}_eta{eta:g}_cfg{cfg:g}"] = y.numpy()
        print(f"eps S={S} eta={eta} cfg={cfg}: std {y.std():.4f}")
    save("ddim_small_eps.npz", **out)


def gen_ddim_options():
    """p_sample_ddim's `score_corrector` (ddim.py:248-250, eps parameterisation only) and `noise_dropout` (:283-284) through the
    REAL sampler on the 256 model's eps path (as gen_ddim_eps).  The corrector and the dropout mask are the seeded recipes of
    golden_recipe.py (torch.nn.functional.dropout is replaced by RecipeDropout for the run: the fixture does not depend on the
    order in which a sampler consumes the global RNG)."""
    import yaml
    rh._install_shims()
    import lvdm.models.samplers.ddim as refddim
    with open(os.path.join(rh.REFERENCE_ROOT, "DynamiCrafter", "configs", "inference_256_v1.0.yaml")) as f:
        kw = yaml.safe_load(f)["model"]["params"]["unet_config"]["params"]
    m = rh.reference_diffusion(dict(kw, model_channels=64, use_checkpoint=False), shell=gr.SHELL_256)
    assert m.parameterization == "eps"
    m.model.diffusion_model.load_state_dict(synth.synth_state_dict(m.model.diffusion_model, seed=WEIGHT_SEED))
    ins, cond, uc = _small_setup()
    out = {}
    real_dropout = torch.nn.functional.dropout
    try:
        for S, eta, cfg, which in gr.DDIM_OPTION_CASES:
            noises = iter(gr.noises(ins["x_T"].shape, S))
            refddim.noise_like = lambda shape, device, repeat=False: next(noises)
            torch.nn.functional.dropout = gr.RecipeDropout()
            opt = {}
            if which in ("score_corrector", "both"):
                opt.update(score_corrector=gr.RecipeCorrector(), corrector_kwargs=dict(gr.CORRECTOR_KWARGS))
            if which in ("noise_dropout", "both"):
                opt.update(noise_dropout=gr.NOISE_DROPOUT_P)
            smp = rh.reference_sampler(m)
            y, _ = smp.sample(S=S, batch_size=1, shape=(4, 16, 8, 8), conditioning=cond, verbose=False,
                              unconditional_guidance_scale=cfg, unconditional_conditioning=uc, eta=eta, fs=torch.tensor([3]),
                              timestep_spacing="uniform_trailing", x_T=ins["x_T"], **opt)
            if which != "score_corrector":
                assert torch.nn.functional.dropout.calls == S
            out[f"S{S}_eta{eta:g}_cfg{cfg:g}_{which}"] = y.numpy()
            print(f"options {which} S={S} eta={eta} cfg={cfg}: std {y.std():.4f}")
    finally:
        torch.nn.functional.dropout = real_dropout
    save("ddim_small_options.npz", **out)


def gen_clip_hf():
    """The image tower against a THIRD-PARTY implementation that is present in this image: HF `transformers.CLIPVisionModel`
    (the class that loads the laion/CLIP-ViT-H-14-laion2B-s32B-b79K conversion of the very checkpoint the reference pulls
    through open_clip, condition.py:306-308).  open_clip_torch itself (pinned 2.22.0, DynamiCrafter/requirements.txt:22) stays
    absent, so this pins the architecture's arithmetic - patch conv, class + position embeddings, pre-LN blocks, fused-qkv
    attention at head dim 80, GELU MLP, all tokens of the LAST block without post-LN / projection = HF's `last_hidden_state` =
    the reference's call site (condition.py:353-382) - not open_clip's code.  Seeded weights in open_clip's key layout
    (`model.visual.*`), mapped key by key onto the HF module; input = the preprocessed 224 x 224 pixels."""
    import transformers
    from transformers import CLIPVisionConfig, CLIPVisionModel
    from oracle import clip_vit_ref
    from open_pandora_amd.clip_vision import VIT_H_14, FrozenOpenCLIPImageEmbedderV2
    out = {"source": np.array(f"transformers {transformers.__version__} CLIPVisionModel.last_hidden_state (f32, CPU)")}
    for tag, cfg in (("small", gr.CLIP_SMALL), ("vit_h_14", dict(VIT_H_14))):
        with torch.device("meta"):
            prod = FrozenOpenCLIPImageEmbedderV2(vision_cfg=cfg)
        sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in prod.state_dict().items()}, seed=gr.CLIP_SEED)
        C, L, H = cfg["width"], cfg["layers"], cfg["heads"]
        hf = CLIPVisionModel(CLIPVisionConfig(hidden_size=C, intermediate_size=int(C * cfg["mlp_ratio"]), num_hidden_layers=L,
                                              num_attention_heads=H, image_size=cfg["image_size"], patch_size=cfg["patch_size"],
                                              hidden_act="gelu", layer_norm_eps=1e-5, attention_dropout=0.0)).eval()
        v = "model.visual."
        m = {"vision_model.embeddings.class_embedding": sd[v + "class_embedding"],
             "vision_model.embeddings.patch_embedding.weight": sd[v + "conv1.weight"],
             "vision_model.embeddings.position_embedding.weight": sd[v + "positional_embedding"],
             "vision_model.pre_layrnorm.weight": sd[v + "ln_pre.weight"], "vision_model.pre_layrnorm.bias": sd[v + "ln_pre.bias"],
             "vision_model.post_layernorm.weight": sd[v + "ln_post.weight"], "vision_model.post_layernorm.bias": sd[v + "ln_post.bias"]}
        for i in range(L):
            a, b = f"{v}transformer.resblocks.{i}.", f"vision_model.encoder.layers.{i}."
            w, bias = sd[a + "attn.in_proj_weight"], sd[a + "attn.in_proj_bias"]
            for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
                m[f"{b}self_attn.{n}.weight"], m[f"{b}self_attn.{n}.bias"] = w[j * C:(j + 1) * C], bias[j * C:(j + 1) * C]
            for src, dst in (("attn.out_proj", "self_attn.out_proj"), ("ln_1", "layer_norm1"), ("ln_2", "layer_norm2"),
                             ("mlp.c_fc", "mlp.fc1"), ("mlp.c_proj", "mlp.fc2")):
                m[f"{b}{dst}.weight"], m[f"{b}{dst}.bias"] = sd[a + src + ".weight"], sd[a + src + ".bias"]
        have = {k for k in hf.state_dict() if not k.endswith("position_ids")}
        if not any(k.startswith("vision_model.") for k in have):  # (transformers >= 5 dropped the wrapper level)
            m = {k[len("vision_model."):]: val for k, val in m.items()}
        assert set(m) == have, sorted(set(m) ^ have)[:6]
        missing, unexpected = hf.load_state_dict(m, strict=False)
        assert not unexpected and all(k.endswith("position_ids") for k in missing), (missing, unexpected)
        img = gr.clip_image(tag)
        px = clip_vit_ref.preprocess(img, cfg["image_size"])
        t0 = time.time()
        with torch.no_grad():
            y = hf(pixel_values=px).last_hidden_state
        print(f"HF CLIPVisionModel {tag}: {time.time() - t0:.1f}s out {tuple(y.shape)} std {y.std():.4f}")
        for k, val in digest(y).items():
            out[f"{tag}/{k}"] = val
    save("clip_vision_hf.npz", **out)


def gen_unet_ctx():
    """The `else` branch of UNetModel.forward's context handling (openaimodel3d.py:565-566): a context that is not
    77 + 16 t tokens long is repeated for every frame; CrossAttention still splits it at token 77."""
    out = {}
    ref = rh.reference_unet(model_channels=64)
    ref.load_state_dict(synth.synth_state_dict(ref, seed=WEIGHT_SEED))
    ins, _, _ = _small_setup(64, 8, 8)
    x = torch.cat([ins["x_T"], ins["c_concat"]], 1)
    for L, tag in gr.UNET_CTX_CASES:
        with torch.no_grad():
            out[tag] = ref(x, torch.tensor([500]), context=ins["c_crossattn"][:, :L], fs=torch.tensor([15])).numpy()
    save("unet_small_ctx.npz", **out)


def gen_unet_adapter():
    """UNetModel.forward with `features_adapter` (openaimodel3d.py:584-596: added to the stream - and thereby to the skip -
    behind input blocks 2, 5, 8, 11), the real module, reduced width."""
    ref = rh.reference_unet(model_channels=64)
    ref.load_state_dict(synth.synth_state_dict(ref, seed=WEIGHT_SEED))
    ins, _, _ = _small_setup(64, 8, 8)
    x = torch.cat([ins["x_T"], ins["c_concat"]], 1)
    with torch.no_grad():
        y = ref(x, torch.tensor([500]), context=ins["c_crossattn"], features_adapter=gr.adapter_features(64, 8, 8),
                fs=torch.tensor([15]))
    save("unet_small_adapter.npz", mc64_8x8_t500=y.numpy())


def gen_ddim_small():
    import lvdm.models.samplers.ddim as refddim
    out = {}
    m = rh.reference_diffusion(dict(model_channels=64))
    m.model.diffusion_model.load_state_dict(synth.synth_state_dict(m.model.diffusion_model, seed=WEIGHT_SEED))
    ins, cond, uc = _small_setup()
    for S, eta, cfg in gr.DDIM_SMALL_CASES:
        noises = iter(gr.noises(ins["x_T"].shape, S))
        refddim.noise_like = lambda shape, device, repeat=False: next(noises)
        smp = rh.reference_sampler(m)
        y, inter = smp.sample(S=S, batch_size=1, shape=(4, 16, 8, 8), conditioning=cond, verbose=False,
                              unconditional_guidance_scale=cfg, unconditional_conditioning=uc, eta=eta,
                              fs=torch.tensor([15]), timestep_spacing="uniform_trailing", x_T=ins["x_T"])
        out[f"S{S}_eta{eta:g}_cfg{cfg:g}"] = y.numpy()  # S10 eta1 is all-NaN by construction
    save("ddim_small.npz", **out)


def gen_ddim_rescale():
    """guidance_rescale > 0 (rescale_noise_cfg, utils_diffusion.py:147-158) through the real sampler."""
    rh._install_shims()
    import lvdm.models.samplers.ddim as refddim
    out = {}
    m = rh.reference_diffusion(dict(model_channels=64))
    m.model.diffusion_model.load_state_dict(synth.synth_state_dict(m.model.diffusion_model, seed=WEIGHT_SEED))
    ins, cond, uc = _small_setup()
    for S, eta, cfg, gres in gr.DDIM_RESCALE_CASES:
        noises = iter(gr.noises(ins["x_T"].shape, S))
        refddim.noise_like = lambda shape, device, repeat=False: next(noises)
        smp = rh.reference_sampler(m)
        y, _ = smp.sample(S=S, batch_size=1, shape=(4, 16, 8, 8), conditioning=cond, verbose=False,
                          unconditional_guidance_scale=cfg, unconditional_conditioning=uc, eta=eta,
                          fs=torch.tensor([15]), timestep_spacing="uniform_trailing", x_T=ins["x_T"],
                          guidance_rescale=gres)
        out[f"S{S}_eta{eta:g}_cfg{cfg:g}_gr{gres:g}"] = y.numpy()
    save("ddim_small_rescale.npz", **out)


def gen_ddim_multicond():
    """The multi-condition sampler (lvdm/models/samplers/ddim_multiplecond.py, selected by model.py:705).  As shipped it dies
    in make_schedule (np.sqrt on the bf16 alphas_cumprod buffer, :40: pinned by tests/test_oracle_vs_reference.py); its
    WORKING form is its own sample / ddim_sampling / p_sample_ddim (three forwards per step, :229-236) on top of the main
    sampler's make_schedule, which differs from :24-57 only by the cast this fork added there (`alphas_cumprod.to(float32)`,
    ddim.py:27).  Both halves are the reference's code, recombined here by inheritance - nothing is restated."""
    rh._install_shims()
    import lvdm.models.samplers.ddim as refddim
    import lvdm.models.samplers.ddim_multiplecond as refmc

    class WorkingMultiCond(refmc.DDIMSampler):
        make_schedule = refddim.DDIMSampler.make_schedule

        def register_buffer(self, name, attr):  # (bypasses the hard-coded `cuda`, as rh.reference_sampler does)
            setattr(self, name, attr)

    out = {}
    m = rh.reference_diffusion(dict(model_channels=64))
    m.model.diffusion_model.load_state_dict(synth.synth_state_dict(m.model.diffusion_model, seed=WEIGHT_SEED))
    ins, cond, uc = _small_setup()
    uc_img = gr.multicond_uc_img(ins, cond, uc)
    for S, eta, cfg, cfg_img, gres in gr.DDIM_MULTICOND_CASES:
        noises = iter(gr.noises(ins["x_T"].shape, S))
        refmc.noise_like = lambda shape, device, repeat=False: next(noises)
        y, _ = WorkingMultiCond(m).sample(S=S, batch_size=1, shape=(4, 16, 8, 8), conditioning=cond, verbose=False,
                                          unconditional_guidance_scale=cfg, unconditional_conditioning=uc, eta=eta,
                                          fs=torch.tensor([15]), timestep_spacing="uniform_trailing", x_T=ins["x_T"],
                                          guidance_rescale=gres, cfg_img=cfg_img,
                                          unconditional_conditioning_img_nonetext=uc_img)
        out[f"S{S}_eta{eta:g}_cfg{cfg:g}_img{cfg_img}_gr{gres:g}"] = y.numpy()
    save("ddim_small_multicond.npz", **out)


def gen_resampler():
    """The real Resampler (resampler.py:96-144) on seeded weights: reduced config in full, the shipped
    image_proj_stage_config (inference_512_v1.0.yaml:91-102) as a digest."""
    rh._install_shims()
    from lvdm.modules.encoders.resampler import Resampler
    out = {}
    for tag, kw, xs in gr.RESAMPLER_CASES:
        m = Resampler(**kw)
        m.load_state_dict(synth.synth_state_dict(m, seed=WEIGHT_SEED))
        x = gr.module_input(f"resampler/{tag}", *xs)
        with torch.no_grad():
            y = m(x)
        if tag == "small":
            out["small"] = y.numpy()
        else:
            for k, v in digest(y, n=8192).items():
                out[f"{tag}/{k}"] = v
    save("resampler.npz", **out)


def gen_full(traj):
    torch.set_num_threads(os.cpu_count() or 8)
    t0 = time.time()
    m = rh.reference_diffusion()
    unet = m.model.diffusion_model
    unet.load_state_dict(synth.synth_state_dict(unet, seed=WEIGHT_SEED))
    print(f"full model ready in {time.time() - t0:.0f}s")
    h, w = 40, 64
    ins, cond, uc = _small_setup(320, h, w)
    out = {}
    if not traj:
        for tag, c in (("cond", cond), ("uncond", uc)):
            t0 = time.time()
            with torch.no_grad():
                y = m.apply_model(ins["x_T"], torch.tensor([500]), c, fs=torch.tensor([15]))
            print(f"forward {tag}: {time.time() - t0:.0f}s std {y.std():.4f}")
            for k, v in digest(y).items():
                out[f"{tag}/{k}"] = v
        save("unet_full_40x64.npz", **out)
    else:
        t0 = time.time()
        smp = rh.reference_sampler(m)
        y, _ = smp.sample(S=10, batch_size=1, shape=(4, 16, h, w), conditioning=cond, verbose=True,
                          unconditional_guidance_scale=4.0, unconditional_conditioning=uc, eta=0.0,
                          fs=torch.tensor([15]), timestep_spacing="uniform_trailing", x_T=ins["x_T"])
        dt = time.time() - t0
        print(f"10-step trajectory: {dt:.0f}s")
        for k, v in digest(y, n=8192).items():
            out[f"sample/{k}"] = v
        out["wall_seconds"] = np.float64(dt)
        out["threads"] = np.int64(torch.get_num_threads())
        save("ddim_full_40x64_s10.npz", **out)


def gen_ae():
    """First-stage decoder: the real AutoencoderKL on seeded weights."""
    rh._install_shims()
    from lvdm.models.autoencoder import AutoencoderKL
    from open_pandora_amd.autoencoder import DDCONFIG
    out = {}
    for tag, ch, T, h, w in (("ch32_3x8x8", 32, 3, 8, 8), ("ch64_2x8x16", 64, 2, 8, 16)):
        ae = AutoencoderKL(ddconfig=dict(DDCONFIG, ch=ch), lossconfig=rh.AttrDict(target="torch.nn.Identity"),
                           embed_dim=4).eval()
        ae.load_state_dict(synth.synth_state_dict(ae, seed=WEIGHT_SEED))
        z = gr.ae_latent(T, h, w)
        with torch.no_grad():
            out[tag] = torch.stack([ae.decode(z[:, :, i] / 0.18215) for i in range(T)], 2).numpy()
            out["enc/" + tag] = ae.encode(gr.ae_pixels(T, 8 * h, 8 * w)).parameters.numpy()  # posterior moments
    save("ae_decode_small.npz", **out)
    torch.set_num_threads(os.cpu_count() or 8)
    ae = AutoencoderKL(ddconfig=dict(DDCONFIG), lossconfig=rh.AttrDict(target="torch.nn.Identity"), embed_dim=4).eval()
    ae.load_state_dict(synth.synth_state_dict(ae, seed=WEIGHT_SEED))
    z = gr.ae_latent(2, 40, 64)
    t0 = time.time()
    with torch.no_grad():
        y = torch.stack([ae.decode(z[:, :, i] / 0.18215) for i in range(2)], 2)
    print(f"full-width AE decode, 2 frames 40x64 -> 320x512: {time.time() - t0:.0f}s std {y.std():.4f}")
    save("ae_decode_full_40x64.npz", **{f"frames2/{k}": v for k, v in digest(y, n=8192).items()})


def gen_oracle_72x128():
    """576x1024 (16x72x128 latent): the reference's eager attention materialises a 27 GB f32 score
    tensor per level-0 block and cannot run in this container, so this fixture is
    produced by the ORACLE (oracle/unet_ref.py, attention chunked over heads - identical arithmetic per
    row), which is itself pinned to the reference by every other fixture.  Marked as such in the file."""
    from oracle import unet_ref
    from open_pandora_amd import factory
    torch.set_num_threads(os.cpu_count() or 8)
    with torch.device("meta"):
        from open_pandora_amd.unet import UNetModel
        shapes = {k: tuple(v.shape) for k, v in UNetModel(**dict(factory.UNET_PARAMS, default_fs=10)).state_dict().items()}
    sd = {k: synth.synth_tensor(k, sh, WEIGHT_SEED) for k, sh in shapes.items()}
    ins, cond, _ = _small_setup(320, 72, 128)
    x = torch.cat([ins["x_T"], ins["c_concat"]], 1)
    t0 = time.time()
    y = unet_ref.unet_forward(sd, x, torch.tensor([500]), ins["c_crossattn"], torch.tensor([15]))
    print(f"oracle forward 72x128: {time.time() - t0:.0f}s std {y.std():.4f}")
    out = {f"cond/{k}": v for k, v in digest(y, n=8192).items()}
    out["source"] = np.array("oracle (oracle/unet_ref.py); the reference cannot run this size on CPU here")
    save("unet_full_72x128_oracle.npz", **out)


def gen_full_72x128(traj_steps=0):
    """576x1024 (16x72x128 latent, 9216 spatial tokens at level 0) through the REAL reference: one conditional
    forward, or a short CFG-4 eta-0 DDIM trajectory.  The reference's attention is called frame by frame
    (rh.chunk_attention_over_frames: a memory shim, same arithmetic)."""
    torch.set_num_threads(os.cpu_count() or 8)
    t0 = time.time()
    m = rh.reference_diffusion(base_scale=0.3)
    unet = m.model.diffusion_model
    unet.load_state_dict(synth.synth_state_dict(unet, seed=WEIGHT_SEED))
    rh.chunk_attention_over_frames(unet)
    print(f"full model ready in {time.time() - t0:.0f}s", flush=True)
    h, w = 72, 128
    ins, cond, uc = _small_setup(320, h, w)
    out = {"source": np.array("the real reference (lvdm UNetModel / DDIMSampler), attention called per frame")}
    if not traj_steps:
        t0 = time.time()
        with torch.no_grad():
            y = m.apply_model(ins["x_T"], torch.tensor([500]), cond, fs=torch.tensor([15]))
        print(f"reference forward 72x128 cond: {time.time() - t0:.0f}s std {y.std():.4f}", flush=True)
        for k, v in digest(y, n=8192).items():
            out[f"cond/{k}"] = v
        save("unet_full_72x128.npz", **out)
    else:
        t0 = time.time()
        smp = rh.reference_sampler(m)
        y, _ = smp.sample(S=traj_steps, batch_size=1, shape=(4, 16, h, w), conditioning=cond, verbose=True,
                          unconditional_guidance_scale=4.0, unconditional_conditioning=uc, eta=0.0,
                          fs=torch.tensor([15]), timestep_spacing="uniform_trailing", x_T=ins["x_T"])
        dt = time.time() - t0
        print(f"{traj_steps}-step trajectory 72x128: {dt:.0f}s", flush=True)
        for k, v in digest(y, n=8192).items():
            out[f"sample/{k}"] = v
        out["wall_seconds"] = np.float64(dt)
        out["threads"] = np.int64(torch.get_num_threads())
        save(f"ddim_full_72x128_s{traj_steps}.npz", **out)


def _reference_first_stage(ch=128):
    """The real AutoencoderKL (lvdm/models/autoencoder.py) on seeded weights, as `first_stage_model`."""
    rh._install_shims()
    from lvdm.models.autoencoder import AutoencoderKL
    from open_pandora_amd.autoencoder import DDCONFIG
    ae = AutoencoderKL(ddconfig=dict(DDCONFIG, ch=ch), lossconfig=rh.AttrDict(target="torch.nn.Identity"),
                       embed_dim=4).eval()
    ae.load_state_dict(synth.synth_state_dict(ae, seed=WEIGHT_SEED))
    return ae


def _sample_and_decode(m, ins, cond, uc, h, w, S, eta, shared_noise):
    """The real DDIMSampler.sample (ddim.py:66) followed by the real LatentDiffusion.decode_first_stage
    (ddpm3d.py:630-655, per-frame AE): latents -> frames."""
    import lvdm.models.samplers.ddim as refddim
    if shared_noise:  # eta > 0: every draw of noise_like comes from the seeded recipe (gr.noises), as in gen_ddim_small
        noises = iter(gr.noises(ins["x_T"].shape, S))
        refddim.noise_like = lambda shape, device, repeat=False: next(noises)
    smp = rh.reference_sampler(m)
    z, _ = smp.sample(S=S, batch_size=1, shape=(4, 16, h, w), conditioning=cond, verbose=False,
                      unconditional_guidance_scale=4.0, unconditional_conditioning=uc, eta=eta,
                      fs=torch.tensor([15]), timestep_spacing="uniform_trailing", x_T=ins["x_T"])
    with torch.no_grad():
        frames = m.decode_first_stage(z)
    return z, frames


def gen_frames_small():
    """End-to-end FRAMES at reduced width (the north-star's tolerance is on frames): sampler -> decode_first_stage,
    all-reference, stored in full (1, 3, 16, 64, 64)."""
    out = {}
    m = rh.reference_diffusion(dict(model_channels=64))
    m.model.diffusion_model.load_state_dict(synth.synth_state_dict(m.model.diffusion_model, seed=WEIGHT_SEED))
    m.first_stage_model = _reference_first_stage(ch=32)
    ins, cond, uc = _small_setup()
    for S, eta in gr.FRAMES_SMALL_CASES:
        z, frames = _sample_and_decode(m, ins, cond, uc, 8, 8, S, eta, shared_noise=eta > 0)
        out[f"S{S}_eta{eta:g}/latent"] = z.numpy()
        out[f"S{S}_eta{eta:g}/frames"] = frames.numpy()
    save("frames_small.npz", **out)


def gen_frames_full(cases, h=40, w=64, forwards=False):
    """Full width (1.44 B U-Net, full AutoencoderKL) at 16x40x64 -> 320x512 frames through the REAL reference:
    (10, 0.0) = BASELINE config 1 (eta 0) and (50, 1.0) = the production schedule with the recipe's shared noise.
    Hours of CPU: run in the background (`nice`).
    (h, w) = (72, 128): BASELINE configs[2]'s 576x1024 frames (r04) - the 1024 yaml's base_scale 0.3, the reference's
    attention called frame by frame (rh.chunk_attention_over_frames: a memory shim, same arithmetic).
    r05 (fixture format 2): latents are committed WHOLE, frames as ~65 K samples at a prime stride + column / row profiles;
    the model is built once per resolution and also emits the single-forward fixtures (`forwards`) and the latent-only
    trajectory fixtures `ddim_full_{h}x{w}_s{S}.npz` (the same `sample()` call as the frames run with eta 0)."""
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", os.cpu_count() or 8)))
    t0 = time.time()
    big = (h, w) == (72, 128)
    m = rh.reference_diffusion(base_scale=0.3) if big else rh.reference_diffusion()
    unet = m.model.diffusion_model
    unet.load_state_dict(synth.synth_state_dict(unet, seed=WEIGHT_SEED))
    if big:
        rh.chunk_attention_over_frames(unet)
    m.first_stage_model = _reference_first_stage()
    print(f"full model + first stage ready in {time.time() - t0:.0f}s", flush=True)
    ins, cond, uc = _small_setup(320, h, w)
    src = "the real reference (lvdm UNetModel / DDIMSampler)" + (", attention called per frame" if big else "")
    if forwards:
        out = {"source": np.array(src)}
        for tag, c in (("cond", cond),) if big else (("cond", cond), ("uncond", uc)):
            t0 = time.time()
            with torch.no_grad():
                y = m.apply_model(ins["x_T"], torch.tensor([500]), c, fs=torch.tensor([15]))
            print(f"reference forward {h}x{w} {tag}: {time.time() - t0:.0f}s std {y.std():.4f}", flush=True)
            for k, v in digest(y).items():
                out[f"{tag}/{k}"] = v
        save(f"unet_full_{h}x{w}.npz", **out)
    for S, eta in cases:
        t0 = time.time()
        z, frames = _sample_and_decode(m, ins, cond, uc, h, w, S, eta, shared_noise=eta > 0)
        dt = time.time() - t0
        print(f"S={S} eta={eta}: {dt:.0f}s latent std {z.std():.4f} frames std {frames.std():.4f}", flush=True)
        out = {"source": np.array("the real reference: DDIMSampler.sample + LatentDiffusion.decode_first_stage")}
        for k, v in digest(z).items():
            out[f"latent/{k}"] = v
        for k, v in digest(frames, n=65536).items():
            out[f"frames/{k}"] = v
        out["frames/per_frame_mean"] = frames[0].double().mean(dim=(0, 2, 3)).numpy()
        out["frames/per_frame_std"] = frames[0].double().std(dim=(0, 2, 3)).numpy()
        out["wall_seconds"] = np.float64(dt)
        out["threads"] = np.int64(torch.get_num_threads())
        if SUFFIX:
            out["weight_seed"], out["input_seed"] = np.int64(WEIGHT_SEED), np.int64(gr.INPUT_SEED)
        save(f"frames_full_{h}x{w}_s{S}_eta{eta:g}{SUFFIX}.npz", **out)
        if eta == 0 and (h, w, S) in ((40, 64, 10), (72, 128, 2)) and not SUFFIX:
            out = {"source": np.array(src), "wall_seconds": np.float64(dt), "threads": np.int64(torch.get_num_threads())}
            for k, v in digest(z).items():
                out[f"sample/{k}"] = v
            save(f"ddim_full_{h}x{w}_s{S}.npz", **out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--full-72x128", action="store_true")
    ap.add_argument("--ctx", action="store_true")
    ap.add_argument("--eps", action="store_true", help="the 256 model's eps-parameterised sampler path (reduced width)")
    ap.add_argument("--ddim-options", action="store_true", help="score_corrector / noise_dropout on the eps sampler path")
    ap.add_argument("--clip-hf", action="store_true", help="image tower fixtures from transformers' CLIPVisionModel")
    ap.add_argument("--learnable", action="store_true", help="image_cross_attention_scale_learnable fixtures (256 yaml)")
    ap.add_argument("--traj-72x128", type=int, default=0)
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--traj", action="store_true")
    ap.add_argument("--oracle-72x128", action="store_true")
    ap.add_argument("--ae", action="store_true")
    ap.add_argument("--rescale", action="store_true")
    ap.add_argument("--multicond", action="store_true")
    ap.add_argument("--adapter", action="store_true")
    ap.add_argument("--resampler", action="store_true")
    ap.add_argument("--frames", action="store_true", help="reduced-width sampler -> decode_first_stage frames (seconds)")
    ap.add_argument("--frames-full", default="", help='full-width cases "S:eta,S:eta", e.g. "10:0,50:1" (hours of CPU)')
    ap.add_argument("--frames-full-72x128", default="", help='the same at 16x72x128 -> 576x1024 frames, e.g. "2:0"')
    ap.add_argument("--with-forwards", action="store_true", help="--frames-full*: also the single-forward fixtures unet_full_*")
    ap.add_argument("--heldout-seeds", default="", help='"W:I": other weight / input seeds for --frames-full* (r06: the held-out '
                    'fixture the selective-parity site list is NOT tuned on); files get the suffix _w<W>_i<I>')
    a = ap.parse_args()
    if a.heldout_seeds:
        assert a.frames_full or a.frames_full_72x128, "--heldout-seeds goes with --frames-full*"
        WEIGHT_SEED, gr.INPUT_SEED = (int(v) for v in a.heldout_seeds.split(":"))
        SUFFIX = f"_w{WEIGHT_SEED}_i{gr.INPUT_SEED}"
    if a.frames or a.frames_full or a.frames_full_72x128:
        assert rh.available()
        if a.frames:
            gen_frames_small()
        if a.frames_full_72x128:
            gen_frames_full([(int(c.split(":")[0]), float(c.split(":")[1])) for c in a.frames_full_72x128.split(",")], 72, 128,
                            forwards=a.with_forwards)
        if a.frames_full:
            gen_frames_full([(int(c.split(":")[0]), float(c.split(":")[1])) for c in a.frames_full.split(",")],
                            forwards=a.with_forwards)
        sys.exit(0)
    if a.ddim_options:
        assert rh.available()
        gen_ddim_options()
        sys.exit(0)
    if a.eps:
        assert rh.available()
        gen_ddim_eps()
        sys.exit(0)
    if a.clip_hf:
        gen_clip_hf()
        sys.exit(0)
    if a.learnable:
        assert rh.available()
        gen_learnable()
        sys.exit(0)
    if a.ctx:
        assert rh.available()
        gen_unet_ctx()
        sys.exit(0)
    if a.full_72x128 or a.traj_72x128:
        assert rh.available()
        gen_full_72x128(a.traj_72x128)
        sys.exit(0)
    if a.resampler:
        assert rh.available()
        gen_resampler()
        sys.exit(0)
    if a.rescale:
        assert rh.available()
        gen_ddim_rescale()
        sys.exit(0)
    if a.multicond:
        assert rh.available()
        gen_ddim_multicond()
        sys.exit(0)
    if a.adapter:
        assert rh.available()
        gen_unet_adapter()
        sys.exit(0)
    if a.ae:
        assert rh.available()
        gen_ae()
        sys.exit(0)
    if a.oracle_72x128:
        gen_oracle_72x128()
        sys.exit(0)
    assert rh.available(), "the reference checkout is required"
    if a.full or a.traj:
        gen_full(a.traj)
    else:
        gen_schedule()
        gen_modules()
        gen_unet_small()
        gen_unet_ctx()
        gen_unet_adapter()
        gen_ddim_small()
        gen_ddim_rescale()
        gen_ddim_multicond()
