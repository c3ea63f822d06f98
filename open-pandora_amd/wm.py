"""Caller surfaces of the hot path, with the names / arguments / return layout of the reference's
world-model driver (model.py), restated around the MI355X path:

  DiffusionRunner.image_guided_synthesis(...)   model.py:703-781   the reference's argument list (condition
                                assembly from `videos` / `diffusion_cond_image`, n_samples loop, sampler call);
                                `_synthesize` is the factored core on already-embedded conditions
  get_latent_z(...)             model.py:690-701   conditioning-frame tiling to 16 latent frames
  DiffusionRunner.generate(...) model.py:783-816   noise_shape from the conditioning frames
  multi-round stitching         model.py:1094-1129, 1199-1211

What is NOT here (out of scope for this tier): the ChatUniVi LLM + QFormer that produce
`diffusion_conditioning` and the OpenCLIP image tower.  The Resampler behind the tower is
`open_pandora_amd.resampler.Resampler` (§8f row 2; `ImageContext` below chains tower -> Resampler and
caches the input-independent unconditional tokens), the AutoencoderKL is `open_pandora_amd.autoencoder`.
They enter as callables / tensors: `embed_image(img) -> (1, 256, 1024)` tokens,
`uncond_text -> (1, 77, 1024)`, `encode_first_stage`, `decode_first_stage`; with `decode_first_stage=None`
the latents are returned (that is what bench.py and the parity tests consume).
"""
import torch
from einops import repeat

from .ddim import DDIMSampler, DDIMSamplerMultiCond


def get_latent_z(encode_first_stage, videos):
    """videos (b, c, t, h, w) pixels -> (b, 4, 16, h/8, w/8): 1 frame is tiled x16, 4 frames x4
    (model.py:690-701, the second - effective - definition)."""
    b, c, t, h, w = videos.shape
    z = encode_first_stage(videos.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w))
    z = z.reshape(b, t, *z.shape[1:]).permute(0, 2, 1, 3, 4)
    if t == 1:
        z = repeat(z, "b c t h w -> b c (repeat t) h w", repeat=4)
    return repeat(z, "b c t h w -> b c (repeat t) h w", repeat=4)


@torch.no_grad()
def _synthesize(diffusion_model, diffusion_conditioning, img_emb, uc_text_emb, uc_img_emb, z_cond,
                noise_shape, n_samples=1, ddim_steps=50, ddim_eta=1.0,
                unconditional_guidance_scale=1.0, cfg_img=None, fs=None, multiple_cond_cfg=False,
                timestep_spacing="uniform", guidance_rescale=0.0, decode_first_stage=None,
                sampler=None, **kwargs):
    """model.py:703-781 with the encoders factored out (private core of `DiffusionRunner.image_guided_synthesis`).
    Returns (batch, n_samples, c, t, h, w): decoded frames if `decode_first_stage` is given, else latents."""
    temporary = None
    if multiple_cond_cfg:
        # model.py:705: the multi-condition sampler.  The reference's own class cannot run on this fork's
        # bf16 schedule buffers (ddim_multiplecond.py:40 raises TypeError: pinned by tests/test_oracle_vs_reference.py);
        # DDIMSamplerMultiCond is its working form (see that class), so the flag now does what model.py:737-743 intends
        sampler, temporary = _multicond_sampler(diffusion_model, sampler)
    elif sampler is not None:
        twin = getattr(sampler, "_multicond_twin", None)
        if twin is not None and twin._graphs:
            twin.close()  # one graph pool (a forward's activations per branch) alive at a time
    sampler = sampler or DDIMSampler(diffusion_model)
    batch_size = noise_shape[0]
    dev = z_cond.device
    fs = torch.tensor([fs] * batch_size, dtype=torch.long, device=dev)
    cond = {"c_crossattn": [torch.cat([diffusion_conditioning, img_emb], dim=1)], "c_concat": [z_cond]}
    uc = None
    if unconditional_guidance_scale != 1.0:
        uc = {"c_crossattn": [torch.cat([uc_text_emb, uc_img_emb], dim=1)], "c_concat": [z_cond]}
    # model.py:736-743: one more unconditional set - image tokens kept, text empty - for the multi-condition sampler
    uc_2 = None
    if multiple_cond_cfg and cfg_img != 1.0 and unconditional_guidance_scale != 1.0:
        uc_2 = {"c_crossattn": [torch.cat([uc_text_emb, img_emb], dim=1)], "c_concat": [z_cond]}
    kwargs.update({"unconditional_conditioning_img_nonetext": uc_2})
    variants = []
    try:
        for _ in range(n_samples):  # independent replicas (model.py:749)
            samples, _ = sampler.sample(S=ddim_steps, conditioning=cond, batch_size=batch_size, shape=noise_shape[1:],
                                        verbose=kwargs.pop("verbose", False),
                                        unconditional_guidance_scale=unconditional_guidance_scale,
                                        unconditional_conditioning=uc, eta=ddim_eta, cfg_img=cfg_img, mask=None, x0=None,
                                        fs=fs, timestep_spacing=timestep_spacing, guidance_rescale=guidance_rescale,
                                        precision=diffusion_conditioning.dtype, **kwargs)
            variants.append(decode_first_stage(samples) if decode_first_stage is not None else samples)
    finally:
        if temporary is not None:
            temporary.close()  # (built for this call only: its three-forward graph and pool go with it)
    return torch.stack(variants).permute(1, 0, 2, 3, 4, 5)


def _multicond_sampler(diffusion_model, sampler):
    """The DDIMSamplerMultiCond to use where the caller holds a plain DDIMSampler: ONE twin per sampler, built
    with that sampler's own op table, graph switch and CFG-pair object - so a multi-rank deployment reaches the twin's
    NotImplementedError guard instead of sampling unsynchronised noise per rank - cached on it (one warm-up and one
    three-forward capture per configuration, not per call) and closed with it.  The primary's graphs are released first:
    two pools would hold two sets of a forward's activations.  -> (sampler, temporary or None)."""
    if isinstance(sampler, DDIMSamplerMultiCond):
        return sampler, None
    if sampler is None:
        tmp = DDIMSamplerMultiCond(diffusion_model)
        return tmp, tmp
    twin = getattr(sampler, "_multicond_twin", None)
    if twin is None or twin.model is not sampler.model:
        twin = DDIMSamplerMultiCond(sampler.model, schedule=sampler.schedule, use_graph=sampler.use_graph,
                                    cfg_parallel=sampler.cfg_parallel, ops=sampler._ops_override)
        sampler._multicond_twin = twin
    if sampler._graphs:
        sampler.close(twin=False)
    return twin, None


class ImageContext:
    """`embedder` (the OpenCLIP image tower, condition.py:300-382: img -> (b, 257, 1280) tokens) followed by
    the Resampler (`image_proj_model`, model.py:711-712).  The unconditional image tokens are the tokens of
    an all-zero image (model.py:728-729): input-independent, so computed once per image shape and cached."""

    def __init__(self, embedder, image_proj_model):
        self.embedder, self.image_proj_model = embedder, image_proj_model
        self._uncond = {}

    def __call__(self, img):
        return self.image_proj_model(self.embedder(img))

    def uncond(self, img):
        key = (tuple(img.shape), img.dtype, str(img.device))
        if key not in self._uncond:
            self._uncond[key] = self(torch.zeros_like(img))
        return self._uncond[key]


class DiffusionRunner:
    """The part of WorldModel.generate / ChatWM that drives the denoiser (model.py:783-816, 989-1129)."""

    GENERATE_KWARGS = {"unconditional_guidance_scale": 4, "ddim_steps": 50, "ddim_eta": 1.0, "fs": 15,
                       "timestep_spacing": "uniform_trailing", "n_samples": 4}  # model.py:989-996

    def __init__(self, diffusion_model, embed_image, uncond_text_emb, encode_first_stage, decode_first_stage=None):
        self.diffusion_model = diffusion_model
        self.embed_image = embed_image
        self.uncond_text_emb = uncond_text_emb
        self.encode_first_stage = encode_first_stage
        self.decode_first_stage = decode_first_stage
        self.sampler = DDIMSampler(diffusion_model)

    @torch.no_grad()
    def image_guided_synthesis(self, diffusion_conditioning, videos, diffusion_cond_image, noise_shape, n_samples=1,
                               ddim_steps=50, ddim_eta=1., unconditional_guidance_scale=1.0, cfg_img=None, fs=None,
                               multiple_cond_cfg=False, loop=False, gfi=False, timestep_spacing='uniform',
                               guidance_rescale=0.0, **kwargs):
        """WorldModel.image_guided_synthesis with the reference's argument list (model.py:703-704):
        diffusion_conditioning (b, 77, 1024) from the LLM side, `videos` (b, 3, 1|4, H, W) conditioning frames
        (-> c_concat through get_latent_z, :717-719), `diffusion_cond_image` (b, 3, H, W) (-> image tokens, :710-712;
        the zero image's tokens for the unconditional branch, :728-729), noise_shape [b, 4, T, h, w].  `loop` and
        `gfi` are accepted and unused, exactly as in the reference's body (they never reach the sampler there
        either: they are named parameters, not **kwargs).  -> (b, n_samples, c, T, H, W)."""
        del loop, gfi
        z = get_latent_z(self.encode_first_stage, videos)
        img_emb = self.embed_image(diffusion_cond_image)
        uc_text = uc_img_emb = None
        if unconditional_guidance_scale != 1.0:
            uc_text = self.uncond_text_emb  # uncond_type "empty_seq": the text encoder's tokens of "" (:723-725)
            uncond = getattr(self.embed_image, "uncond", None)  # ImageContext caches the zero-image tokens
            uc_img_emb = (uncond(diffusion_cond_image) if uncond
                          else self.embed_image(torch.zeros_like(diffusion_cond_image)))
        return _synthesize(self.diffusion_model, diffusion_conditioning, img_emb, uc_text, uc_img_emb, z, noise_shape,
                           n_samples=n_samples, ddim_steps=ddim_steps, ddim_eta=ddim_eta,
                           unconditional_guidance_scale=unconditional_guidance_scale, cfg_img=cfg_img, fs=fs,
                           multiple_cond_cfg=multiple_cond_cfg, timestep_spacing=timestep_spacing,
                           guidance_rescale=guidance_rescale, sampler=self.sampler,
                           decode_first_stage=self.decode_first_stage, **kwargs)

    @torch.no_grad()
    def generate(self, diffusion_conditioning, diffusion_pixel_values, diffusion_cond_image, **generate_kwargs):
        """diffusion_conditioning (1, 77, 1024) from the LLM side; diffusion_pixel_values (3, 1|4, H, W)
        conditioning frames; diffusion_cond_image (1, 3, H, W).  -> (1, n_samples, c, 16, h, w)  (model.py:783-816)."""
        kw = dict(self.GENERATE_KWARGS)
        kw.update(generate_kwargs)
        h, w = diffusion_pixel_values.shape[-2:]
        T = self.diffusion_model.temporal_length
        return self.image_guided_synthesis(diffusion_conditioning[-1:], diffusion_pixel_values[None, ...],
                                           diffusion_cond_image, [1, 4, T, h // 8, w // 8], **kw)

    @staticmethod
    def stitch_rounds(videos):
        """process_generated_video_multi (model.py:1199-1211): every round keeps frames 0-11, the last
        all 16 -> 5 rounds = 64 frames."""
        parts = [v[:, :, :, :12] for v in videos[:-1]] + [videos[-1]]
        return torch.cat(parts, dim=3)

    @staticmethod
    def next_round_condition(frames):
        """model.py:1120: the last 4 generated frames become the next round's conditioning frames."""
        return frames[:, :, -4:]

    @staticmethod
    def next_round_pixels(videos):
        """process_img_from_output (model.py:1179-1187): the first sample's last 4 frames, clamped to [-1, 1],
        through the reference's PIL round trip - `to_pil_image` of a float tensor is `mul(255).byte()` (8-bit,
        truncating) and the diffusion image processor maps it back to [-1, 1] - as `diffusion_pixel_values`
        (3, 4, H, W).  videos: (1, n_samples, 3, 16, H, W)."""
        x = videos[0, 0][:, -4:].detach().float().clamp(-1.0, 1.0)
        x8 = ((x + 1.0) / 2.0).mul(255.0).to(torch.uint8)
        # the way back to [-1, 1] is the reference's HOST arithmetic (ToTensor: / 255, Normalize: (t - 0.5) / 0.5, on the CPU):
        # a 256-entry table computed there - the device's f32 division differs from it in the last bit for some of the 256
        # values, which the bf16 conditioning frames then carry into the next round
        lut = ((torch.arange(256, dtype=torch.float32) / 255.0) - 0.5) / 0.5
        return lut.to(x8.device)[x8.long()]

    @torch.no_grad()
    def generate_multiround(self, conditionings, diffusion_pixel_values, diffusion_cond_image, **generate_kwargs):
        """ChatWM.generate_video_mutliround (model.py:1094-1129) around the denoiser: one `generate` per entry of
        `conditionings` (the LLM-side (1, 77, 1024) tensor of each round - the prompt grows by the 16 generated
        frames per round on that side, out of scope here); every further round is conditioned on the previous
        round's last 4 frames (tiled x4 by get_latent_z) while `diffusion_cond_image` stays the first image;
        rounds are stitched 12 + ... + 12 + 16 frames (process_generated_video_multi, :1199-1211).
        -> (1, n_samples, 3, 12 (R - 1) + 16, H, W).  Needs `decode_first_stage` (pixels feed the next round)."""
        if self.decode_first_stage is None:
            raise ValueError("generate_multiround needs decode_first_stage: round r+1 is conditioned on round r's frames")
        R = len(conditionings)
        videos, cat = None, []
        for r, cond in enumerate(conditionings):
            pix = diffusion_pixel_values if r == 0 else self.next_round_pixels(videos).to(diffusion_pixel_values)
            videos = self.generate(cond, pix, diffusion_cond_image, round_info=[r + 1, R], **generate_kwargs)
            cat.append(videos)
        return self.stitch_rounds(cat)
