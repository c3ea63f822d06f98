"""The reference's caller surfaces BY NAME (model.py of the reference checkout), around the MI355X denoising path:

  load_wm(repo_id, training_args=None, model=None) -> (model, processor)         model.py:469-504
  dynamic_resize(img)                                                             model.py:507-513
  WorldModel.generate(input_ids, pixel_values, diffusion_pixel_values, diffusion_cond_image, attention_mask, tokenizer,
                      **generate_kwargs) / .image_guided_synthesis(...) / .get_latent_z(...)           model.py:690-816
  ChatWM(model, processor, training_args=None, video_path=None)                                       model.py:982-1211
      .generate_video / .generate_video_next_round[2-5] / .generate_video_mutliround[_separate] (sic)
      .process_img / .process_img_from_output / .process_generated_video[_multi]

so that `gradio_app.py`'s wiring (`chatwm.generate_video`, ..., gradio_app.py:200-212) and scripts written against the
reference's `model.py` find the same names, argument lists, state (`cat_videos`, `text_list`, `current_round`,
`generate_kwargs` defaults model.py:989-996) and return layout.  What is NOT rebuilt here (out of scope
for this tier) is injected instead of constructed: the ChatUniVi LLM + QFormer behind `get_diffusion_conditioning`
(model.py:616-688), the HF tokenizer, the CLIP image processor, gradio itself and the mp4 writer.  Everything from
`diffusion_conditioning` on - conditioning-frame encode, image tokens, the CFG DDIM loop, first-stage decode, the
multi-round stitching - is `wm.DiffusionRunner` on the HIP kernels.
"""
import uuid

import numpy as np
import torch

from . import wm

torch_device = "cuda" if torch.cuda.is_available() else "cpu"  # (model.py:464)


def dynamic_resize(img):
    """model.py:507-513: shorter side to 576 (bilinear, antialiased, as torchvision's Resize on a PIL image), then a centre
    crop to 576 x 1024.  img: PIL image -> PIL image."""
    from PIL import Image
    w, h = img.size
    nw, nh = (576, int(576 * h / w)) if w <= h else (int(576 * w / h), 576)
    img = img.resize((nw, nh), Image.BILINEAR)
    if nw < 1024 or nh < 576:
        # torchvision's CenterCrop pads a smaller image with zeros FIRST: (crop - size) // 2 on the left / top, the rest on the
        # right / bottom (functional.center_crop) - not round((size - crop) / 2) of the negative offset (one pixel off
        # when the deficit is 3 mod 4)
        pl, pt = max(0, (1024 - nw) // 2), max(0, (576 - nh) // 2)
        canvas = Image.new(img.mode, (max(nw, 1024), max(nh, 576)))
        canvas.paste(img, (pl, pt))
        img, (nw, nh) = canvas, canvas.size
    left, top = int(round((nw - 1024) / 2.0)), int(round((nh - 576) / 2.0))
    return img.crop((left, top, left + 1024, top + 576))


def diffusion_image_processor(img):
    """`transforms.Compose([ToTensor(), Normalize(0.5, 0.5)])` of load_wm (model.py:490-492): PIL / HWC uint8 -> (3, H, W) in
    [-1, 1]."""
    a = np.asarray(img)
    if a.ndim == 2:
        a = a[:, :, None].repeat(3, 2)
    t = torch.from_numpy(np.array(a, copy=True)).permute(2, 0, 1).float() / 255.0
    return (t - 0.5) / 0.5


def _to_pil(frame):
    """torchvision's `to_pil_image(float tensor (3, H, W) in [0, 1], mode='RGB')`: `mul(255).byte()` - truncating."""
    from PIL import Image
    return Image.fromarray(frame.mul(255).to(torch.uint8).permute(1, 2, 0).cpu().numpy(), mode="RGB")


class WorldModel:
    """The part of the reference's WorldModel (model.py:506-974) that lies on the denoising path.  `runner`: the
    wm.DiffusionRunner bound to the HIP U-Net / first stage / image context; `get_diffusion_conditioning`: the LLM side
    (model.py:616-688) as a callable with the reference's positional signature
    `(input_ids, pixel_values, attention_mask, return_dict, output_attentions, output_hidden_states) -> (n, 77, 1024)`."""

    def __init__(self, runner, get_diffusion_conditioning, config=None, video_model=None):
        self.runner = runner
        self.diffusion_model = runner.diffusion_model
        self._conditioner = get_diffusion_conditioning
        self.config = config
        self.video_model = video_model

    def get_diffusion_conditioning(self, input_ids, pixel_values=None, attention_mask=None, return_dict=True,
                                   output_attentions=None, output_hidden_states=None):
        return self._conditioner(input_ids, pixel_values, attention_mask, return_dict, output_attentions, output_hidden_states)

    # -- training surface (model.py:926-974; the Lightning trainer around it is out of scope) ---------------------------------
    def training_step(self, batch, batch_idx):
        """model.py:926-942 (`do_alignment` False: the diffusion objective): `get_batch_input` - the LLM side + first-stage encode
        of the batch, injected as `self.get_batch_input` - then `loss, loss_dict = self.diffusion_model(x, c, fs=fs.long())`
        (LatentVisualDiffusion.forward -> p_losses; the U-Net takes its differentiable walk).  -> loss; the dictionary the
        reference hands to `log_dict` is kept as `self.last_loss_dict`."""
        del batch_idx
        # (factory.build_diffusion hands the U-Net out in eval(); its forward then takes the no_grad kernel path and
        # the loss has no grad_fn - say so here instead of failing inside autograd)
        unet = self.diffusion_model.model.diffusion_model
        if not unet.training:
            raise RuntimeError("WorldModel.training_step: the diffusion U-Net is in eval() - call .train() on it (WorldModel.train()) "
                               "first; in eval mode its forward runs the forward-only HIP kernels")
        x, c, fs = self.get_batch_input(**batch, random_uncond=False)
        loss, self.last_loss_dict = self.diffusion_model(x, c, fs=fs.long())
        if not loss.requires_grad:
            raise RuntimeError("WorldModel.training_step: the loss carries no gradient (autograd disabled, or no parameter requires grad)")
        return loss

    def train(self, mode=True):
        """nn.Module-style switch forwarded to the diffusion model (the shell itself is a plain object)"""
        self.diffusion_model.train(mode)
        return self

    def eval(self):
        return self.train(False)

    def configure_optimizers(self):
        """model.py:951-974 (`do_alignment` False): AdamW over `diffusion_model.model.parameters()` (the DiffusionWrapper = the
        U-Net's 1516 tensors) plus the LLM-side bridge parameters handed in as `self.extra_trainable`; lr from `config`."""
        if getattr(self, "config", None) is None or not hasattr(self.config, "learning_rate"):
            raise ValueError("WorldModel.configure_optimizers: `config.learning_rate` is required (model.py:951-974 reads it)")
        params = list(self.diffusion_model.model.parameters()) + list(getattr(self, "extra_trainable", []))
        return torch.optim.AdamW(params, lr=self.config.learning_rate)

    def get_latent_z(self, model, videos):
        """model.py:690-701 (`model` = the diffusion model, kept for the signature)."""
        del model
        return wm.get_latent_z(self.runner.encode_first_stage, videos)

    @torch.no_grad()
    def image_guided_synthesis(self, diffusion_conditioning, videos, diffusion_cond_image, noise_shape, n_samples=1,
                               ddim_steps=50, ddim_eta=1., unconditional_guidance_scale=1.0, cfg_img=None, fs=None,
                               multiple_cond_cfg=False, loop=False, gfi=False, timestep_spacing='uniform',
                               guidance_rescale=0.0, **kwargs):
        """model.py:703-781, argument for argument."""
        return self.runner.image_guided_synthesis(
            diffusion_conditioning, videos, diffusion_cond_image, noise_shape, n_samples=n_samples, ddim_steps=ddim_steps,
            ddim_eta=ddim_eta, unconditional_guidance_scale=unconditional_guidance_scale, cfg_img=cfg_img, fs=fs,
            multiple_cond_cfg=multiple_cond_cfg, loop=loop, gfi=gfi, timestep_spacing=timestep_spacing,
            guidance_rescale=guidance_rescale, **kwargs)

    @torch.no_grad()
    def generate(self, input_ids, pixel_values=None, diffusion_pixel_values=None, diffusion_cond_image=None,
                 attention_mask=None, tokenizer=None, **generate_kwargs):
        """model.py:783-816: batch size 1; the prompt must end in the image-prefix token; only the LAST conditioning row is
        generated; noise_shape from the conditioning frames.  -> (1, n_samples, 3, 16, H, W)."""
        assert input_ids.size(0) == 1, "Currently only support batch size 1"
        assert input_ids[0][-1] == tokenizer.image_prefix_token_id
        cond = self.get_diffusion_conditioning(input_ids, pixel_values, attention_mask, True, None, None)
        cond = cond[-1:]  # Only generate last video
        h, w = diffusion_pixel_values.shape[-2:]
        return self.image_guided_synthesis(diffusion_conditioning=cond, videos=diffusion_pixel_values[None, ...],
                                           diffusion_cond_image=diffusion_cond_image,
                                           noise_shape=[1, 4, self.diffusion_model.temporal_length, h // 8, w // 8],
                                           **generate_kwargs)


def load_wm(repo_id, training_args=None, model=None, *, runner=None, get_diffusion_conditioning=None, tokenizer=None,
            image_processor=None):
    """model.py:469-504: -> (model, processor) with processor = {'image_processor', 'diffusion_image_processor',
    'tokenizer'}.  The reference pulls config, LLM weights and tokenizer from the HF hub (`repo_id`); offline, and with the
    LLM out of scope, the pieces that are not the denoiser arrive as keyword arguments: `model` (a WorldModel) or `runner` +
    `get_diffusion_conditioning` to build one; `tokenizer` / `image_processor` as loaded by the caller.  The three special
    token ids the reference caches on the tokenizer (model.py:495-497) are set here too when the tokenizer can resolve them."""
    del training_args  # (do_alignment / learning_rate: training-side config, model.py:476-484)
    if model is None:
        if runner is None or get_diffusion_conditioning is None:
            raise ValueError(f"load_wm({repo_id!r}): offline build - pass model=WorldModel(...) or runner= and "
                             "get_diffusion_conditioning= (the LLM side is not part of this library)")
        model = WorldModel(runner, get_diffusion_conditioning)
    if tokenizer is not None and hasattr(tokenizer, "convert_tokens_to_ids"):
        tokenizer.image_start_token_id = tokenizer.convert_tokens_to_ids("<img_s>")
        tokenizer.image_token_id = tokenizer.convert_tokens_to_ids("<image>")
        tokenizer.image_prefix_token_id = tokenizer.convert_tokens_to_ids("[IMG_P]")
    processor = {"image_processor": image_processor, "diffusion_image_processor": diffusion_image_processor,
                 "tokenizer": tokenizer}
    return model, processor


def _ui_update(**kw):
    """Stand-in for `gr.update(...)` (gradio is not part of this library): the same dictionary gradio builds."""
    return dict(kw, __type__="update")


class ChatWM:
    """model.py:982-1211 without gradio and the mp4 encoder: the chat-session state machine around WorldModel.generate.
    `video_writer(path, uint8 frames (t, H, W, 3), fps)` receives what the reference hands to torchvision.io.write_video;
    default: keep the frames in `self.written[path]`."""

    def __init__(self, model, processor, training_args=None, video_path=None, video_writer=None):
        del video_path
        self.model = model
        self.image_processor = processor['image_processor']
        self.diffusion_image_processor = processor['diffusion_image_processor']
        self.tokenizer = processor['tokenizer']
        self.generate_kwargs = dict(wm.DiffusionRunner.GENERATE_KWARGS)  # model.py:989-996
        self.cat_videos = []
        self.text = ''
        self.pixel_values = None
        self.diffusion_cond_image = None
        self.current_round = 0
        self.video_path = [f'./video_output/video_output_gradio_round{i}_{uuid.uuid4()}.mp4' for i in range(10)]
        self.text_list = []
        self.config = training_args
        self.written = {}
        self._writer = video_writer or (lambda path, frames, fps: self.written.__setitem__(path, frames))

    # ---- request plumbing ------------------------------------------------------------------------------------------
    def _set_kwargs(self, ddim_steps, fs, n_samples, unconditional_guidance_scale, ddim_eta, progress, rounds):
        self.generate_kwargs.update(ddim_steps=ddim_steps, fs=fs, n_samples=n_samples,
                                    unconditional_guidance_scale=unconditional_guidance_scale, ddim_eta=ddim_eta,
                                    gr_progress_bar=progress, round_info=[1, rounds])

    def _batch(self, text, extra):
        batch = dict(self.tokenizer(text, return_tensors="pt", add_special_tokens=False))
        batch.update(extra)
        return {k: v.to(torch_device) for k, v in batch.items() if isinstance(v, torch.Tensor)}

    def _generate(self, batch):
        return self.model.generate(**batch, tokenizer=self.tokenizer, **self.generate_kwargs)

    # ---- the gradio callbacks (gradio_app.py:200-212) -----------------------------------------------------------------
    def generate_video(self, image, text_input, ddim_steps, fs, n_samples, unconditional_guidance_scale, ddim_eta,
                       progress=None):
        self._set_kwargs(ddim_steps, fs, n_samples, unconditional_guidance_scale, ddim_eta, progress, 1)
        self.current_round = 1
        if self.model is None:  # debug mode
            return self.video_path[0]
        self.text = self.tokenizer.bos_token + "<image> " + text_input + "[IMG_P]" * 64
        batch = self._batch(self.text, self.process_img(image))
        videos = self._generate(batch)
        self.cat_videos = [videos]
        self.text_list = [self.text]
        self.pixel_values = batch['pixel_values']
        self.diffusion_cond_image = batch['diffusion_cond_image']
        self.process_generated_video(videos, fps=8, video_path=self.video_path[1])
        return (self.video_path[1], self.video_path[1], _ui_update(interactive=True, value='🔄 Re-do Action 1'),
                _ui_update(interactive=True), _ui_update(interactive=False))

    def generate_video_next_round(self, text_input, ddim_steps, fs, n_samples, unconditional_guidance_scale, ddim_eta,
                                  progress=None):
        self._set_kwargs(ddim_steps, fs, n_samples, unconditional_guidance_scale, ddim_eta, progress, 1)
        if self.model is None:  # debug mode
            return self.video_path[0]
        self.cat_videos = self.cat_videos[:self.current_round - 1]
        self.text_list = self.text_list[:self.current_round - 1]
        self.text = ''.join(self.text_list) + "<image>" * 16 + text_input + "[IMG_P]" * 64
        extra = self.process_img_from_output(self.cat_videos[-1], self.pixel_values)
        extra['diffusion_cond_image'] = self.diffusion_cond_image
        batch = self._batch(self.text, extra)
        videos = self._generate(batch)
        self.cat_videos.append(videos)
        self.pixel_values = batch['pixel_values']
        self.process_generated_video(videos, fps=8, video_path=self.video_path[self.current_round])
        self.process_generated_video_multi(self.cat_videos, fps=8, video_path=self.video_path[0], num_round=len(self.cat_videos))
        return (self.video_path[0], self.video_path[self.current_round],
                _ui_update(interactive=True, value=f'🔄 Re-do Action {self.current_round}'), _ui_update(interactive=True))

    def _next_round(self, n, *a, **k):
        self.current_round = n
        return self.generate_video_next_round(*a, **k)

    def generate_video_next_round2(self, *a, **k):
        return self._next_round(2, *a, **k)

    def generate_video_next_round3(self, *a, **k):
        return self._next_round(3, *a, **k)

    def generate_video_next_round4(self, *a, **k):
        return self._next_round(4, *a, **k)

    def generate_video_next_round5(self, *a, **k):
        return self._next_round(5, *a, **k)

    def _rounds(self, image, text_input, num_round, each=None):
        """The loop shared by the two multi-round entry points (model.py:1094-1129 / 1131-1176): round r + 1 is conditioned
        on the first sample's last 4 frames of round r through the 8-bit PIL round trip (process_img_from_output), the prompt
        grows by the 16 generated frames per round, `diffusion_cond_image` stays the first image."""
        text = self.tokenizer.bos_token + "<image> " + text_input + "[IMG_P]" * 64
        batch = self._batch(text, self.process_img(image))
        videos = self._generate(batch)
        if each:
            each(0, videos)
        cat_videos = [videos]
        for j in range(1, num_round):
            self.generate_kwargs['round_info'][0] += 1
            text += "<image>" * 16 + text_input + "[IMG_P]" * 64
            batch.update(dict(self.tokenizer(text, return_tensors="pt", add_special_tokens=False)))
            batch.update(self.process_img_from_output(videos, batch['pixel_values']))
            batch = {k: v.to(torch_device) for k, v in batch.items() if isinstance(v, torch.Tensor)}
            videos = self._generate(batch)
            if each:
                each(j, videos)
            cat_videos.append(videos)
        return cat_videos

    def generate_video_mutliround(self, image, text_input, ddim_steps, fs, n_samples, unconditional_guidance_scale, ddim_eta,
                                  num_round=2, video_path=None, progress=None):
        video_path = video_path or f'./video_output/video_output_gradio_multiturn_{uuid.uuid4()}.mp4'
        self._set_kwargs(ddim_steps, fs, n_samples, unconditional_guidance_scale, ddim_eta, progress, num_round)
        if self.model is None:  # debug mode
            return video_path
        cat_videos = self._rounds(image, text_input, num_round)
        self.process_generated_video_multi(cat_videos, fps=8, video_path=video_path, num_round=num_round)
        return (video_path,) + tuple(_ui_update(interactive=False) for _ in range(4))

    def generate_video_mutliround_separate(self, image, text_input, ddim_steps, fs, n_samples, unconditional_guidance_scale,
                                           ddim_eta, num_round=2, progress=None):
        self._set_kwargs(ddim_steps, fs, n_samples, unconditional_guidance_scale, ddim_eta, progress, num_round)
        video_path_list = [f'./video_output/video_output_gradio_{i}.mp4' for i in range(num_round + 1)]
        if self.model is None:  # debug mode
            return video_path_list
        # (the reference writes round 1 to list[1] and round j + 1 to list[j]: round 2 overwrites round 1's file, :1160,1173)
        each = lambda j, v: self.process_generated_video(v, fps=8, video_path=video_path_list[1 if j == 0 else j])
        cat_videos = self._rounds(image, text_input, num_round, each)
        self.process_generated_video_multi(cat_videos, fps=8, video_path=video_path_list[0], num_round=num_round)
        return video_path_list

    # ---- pixels in, pixels out -------------------------------------------------------------------------------------------
    def process_img(self, image):
        """model.py:1179-1184: `image` HWC uint8 -> the LLM side's pixel_values (the injected CLIP processor), the 576 x 1024
        conditioning frame (3, 1, H, W) and the conditioning image (1, 3, H, W), all bf16."""
        from PIL import Image
        pixel_values = self.image_processor(images=image, return_tensors="pt").pixel_values.to(torch_device)
        resized = dynamic_resize(Image.fromarray(image))
        dpv = self.diffusion_image_processor(resized).unsqueeze(1)
        return {'pixel_values': pixel_values.bfloat16(), 'diffusion_pixel_values': dpv.bfloat16(),
                'diffusion_cond_image': dpv.unsqueeze(0)[:, :, 0].bfloat16()}

    def process_img_from_output(self, videos, pixel_values):
        """model.py:1186-1194: the first sample's 16 frames, clamped, as 8-bit PIL images -> appended to the LLM side's
        pixel_values; the last 4 of them, resized, as the next round's conditioning frames (3, 4, H, W)."""
        frames = videos.squeeze(0)[0].detach().permute((1, 0, 2, 3)).clamp(-1., 1.).to(torch.float32)
        pil = [_to_pil((f + 1.) / 2.) for f in frames]
        new_pv = self.image_processor(images=pil, return_tensors="pt").pixel_values.to(torch_device)
        pixel_values = torch.cat((pixel_values, new_pv.to(pixel_values)), dim=0)
        dpv = torch.stack([self.diffusion_image_processor(dynamic_resize(im).convert('RGB')) for im in pil[-4:]], dim=1)
        return {'pixel_values': pixel_values.bfloat16(), 'diffusion_pixel_values': dpv.bfloat16()}

    def process_generated_video(self, videos, fps=8, video_path='video_output.mp4'):
        """model.py:1198-1204: the n_samples clips tiled two per row into one frame sheet per time step, 8-bit."""
        video = videos.squeeze(0).detach().cpu().to(torch.float32).clamp(-1., 1.).permute(2, 0, 1, 3, 4)  # t n c h w
        t, n, c, h, w = video.shape
        cols = min(2, n)
        rows = (n + cols - 1) // cols
        grid = torch.zeros(t, c, rows * h, cols * w)
        for i in range(n):
            r, q = divmod(i, cols)
            grid[:, :, r * h:(r + 1) * h, q * w:(q + 1) * w] = video[:, i]
        self._writer(video_path, ((grid + 1.) / 2. * 255.).to(torch.uint8).permute(0, 2, 3, 1), fps)

    def process_generated_video_multi(self, cat_videos, fps=8, video_path='video_output.mp4', num_round=2):
        """model.py:1206-1219: frames 0-11 of every round but the last, all 16 of the last (the conditioning overlap is
        cut), first sample only."""
        keep = [list(range(0, 12))]
        for i in range(1, num_round):
            keep.append(list(range(i * 16, (i + 1) * 16 if i == num_round - 1 else (i + 1) * 16 - 4)))
        video = torch.cat(cat_videos, dim=3).squeeze(0)[0].detach().cpu().float().clamp(-1., 1.)
        video = ((video + 1.) / 2. * 255.).permute((1, 2, 3, 0))
        self._writer(video_path, torch.cat([video[idx] for idx in keep], dim=0), fps)
