"""Diffusion-model shell of the hot path: schedule tables, `apply_model`, and the v-parameterisation
helpers the sampler needs (the sampler seam of SURVEY §8(b)).

Mirrors the inference-relevant surface of lvdm.models.ddpm3d:
  DDPM.register_schedule            ddpm3d.py:119-182   (tables stored in bf16, :139)
  LatentDiffusion scale_arr         ddpm3d.py:505-510   (dynamic rescale, bf16)
  LatentDiffusion.apply_model       ddpm3d.py:724-739
  DiffusionWrapper.forward (hybrid) ddpm3d.py:1066-1081
  predict_{start,eps}_from_z_and_v  ddpm3d.py:235-247
Losses, logging, EMA, the DDPM ancestral sampler, first-stage and condition encoders are out of scope
(SURVEY §2.1 rows 2, 8, 9); condition tensors are handed in directly.
"""
import numpy as np
import torch
import torch.nn as nn

from .unet import UNetModel


def linear_beta_schedule(n_timestep, linear_start, linear_end):
    """utils_diffusion.py:31-36 ('linear'): squared linspace of the square roots, in float64."""
    return (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2).numpy()


def zero_terminal_snr(betas):
    """utils_diffusion.py:112-144 (arXiv 2305.08891 Alg. 1): shift/scale sqrt(alpha_bar) so that the
    last step has exactly zero SNR and the first keeps its value."""
    abar_sqrt = np.sqrt(np.cumprod(1.0 - betas, axis=0))
    first, last = abar_sqrt[0].copy(), abar_sqrt[-1].copy()
    abar_sqrt = (abar_sqrt - last) * (first / (first - last))
    abar = abar_sqrt ** 2
    alphas = np.concatenate([abar[0:1], abar[1:] / abar[:-1]])
    return 1.0 - alphas


class DiffusionWrapper(nn.Module):
    """ddpm3d.py:1060-1129, conditioning keys used by the shipped configs."""

    def __init__(self, diffusion_model, conditioning_key):
        super().__init__()
        self.diffusion_model = diffusion_model
        self.conditioning_key = conditioning_key
        if conditioning_key not in ("hybrid", "crossattn", "concat"):
            raise NotImplementedError(f"conditioning_key={conditioning_key!r}")

    def forward(self, x, t, c_concat=None, c_crossattn=None, **kwargs):
        if self.conditioning_key in ("hybrid", "concat"):
            x = torch.cat([x] + [c.to(x.dtype) for c in c_concat], dim=1)
        cc = torch.cat(c_crossattn, 1) if self.conditioning_key != "concat" else None
        return self.diffusion_model(x, t, context=cc, **kwargs)


class LatentVisualDiffusion(nn.Module):
    """Inference shell with the attribute set DDIMSampler reads (ddim.py:14,27-50,229-277)."""

    def __init__(self, unet_config, timesteps=1000, linear_start=0.00085, linear_end=0.012,
                 parameterization="v", rescale_betas_zero_snr=True, conditioning_key="hybrid",
                 use_dynamic_rescale=True, base_scale=0.7, turning_step=400, scale_factor=0.18215,
                 channels=4, image_size=(40, 64), **unused):
        super().__init__()
        if parameterization not in ("v", "eps"):  # ("x0": no shipped config; eps = the 256 yaml's class default)
            raise NotImplementedError(f"parameterization={parameterization!r}: the shipped configs use v- or eps-prediction")
        self.parameterization = parameterization
        self.rescale_betas_zero_snr = rescale_betas_zero_snr
        self.channels = channels
        self.image_size = list(image_size)
        self.scale_factor = scale_factor
        unet = unet_config if isinstance(unet_config, nn.Module) else UNetModel(**dict(unet_config))
        self.temporal_length = unet.temporal_length
        self.model = DiffusionWrapper(unet, conditioning_key)
        self.use_dynamic_rescale = use_dynamic_rescale
        self.register_schedule(timesteps, linear_start, linear_end)
        if use_dynamic_rescale:
            arr = np.concatenate((np.linspace(1.0, base_scale, turning_step), np.full(self.num_timesteps, base_scale)))
            self.register_buffer("scale_arr", torch.tensor(arr, dtype=torch.bfloat16))
        # the training objective's constants at their class defaults (ddpm3d.py:48-71,101-117): plain l2 on the prediction
        # target, no learned log-variance, no ELBO term
        self.loss_type, self.l_simple_weight, self.original_elbo_weight, self.learn_logvar = "l2", 1.0, 0.0, False
        self.logvar = torch.zeros(self.num_timesteps)

    @property
    def device(self):
        return self.betas.device

    def register_schedule(self, timesteps, linear_start, linear_end):
        betas = linear_beta_schedule(timesteps, linear_start, linear_end)
        if self.rescale_betas_zero_snr:
            betas = zero_terminal_snr(betas)
        alphas = 1.0 - betas
        ac = np.cumprod(alphas, axis=0)
        ac_prev = np.append(1.0, ac[:-1])
        self.num_timesteps = int(betas.shape[0])
        self.linear_start, self.linear_end = linear_start, linear_end
        bf = lambda a: torch.tensor(a, dtype=torch.bfloat16)  # the fork stores every table in bf16
        self.register_buffer("betas", bf(betas))
        self.register_buffer("alphas_cumprod", bf(ac))
        self.register_buffer("alphas_cumprod_prev", bf(ac_prev))
        self.register_buffer("sqrt_alphas_cumprod", bf(np.sqrt(ac)))
        self.register_buffer("sqrt_one_minus_alphas_cumprod", bf(np.sqrt(1.0 - ac)))
        with np.errstate(divide="ignore"):
            self.register_buffer("log_one_minus_alphas_cumprod", bf(np.log(1.0 - ac)))
        self.register_buffer("sqrt_recip_alphas_cumprod", torch.zeros(timesteps, dtype=torch.bfloat16))
        self.register_buffer("sqrt_recipm1_alphas_cumprod", torch.zeros(timesteps, dtype=torch.bfloat16))

    # -- sampler seam ----------------------------------------------------------------------------
    def apply_model(self, x_noisy, t, cond, **kwargs):
        if not isinstance(cond, dict):
            cond = {"c_crossattn": cond if isinstance(cond, list) else [cond]}
        out = self.model(x_noisy, t, **cond, **kwargs)
        return out[0] if isinstance(out, tuple) else out

    def _coef(self, table, t, x):
        return table[t].reshape(-1, *((1,) * (x.dim() - 1))).to(x.device)

    def predict_start_from_z_and_v(self, x_t, t, v):
        return self._coef(self.sqrt_alphas_cumprod, t, x_t) * x_t - self._coef(self.sqrt_one_minus_alphas_cumprod, t, x_t) * v

    def predict_eps_from_z_and_v(self, x_t, t, v):
        return self._coef(self.sqrt_alphas_cumprod, t, x_t) * v + self._coef(self.sqrt_one_minus_alphas_cumprod, t, x_t) * x_t

    # -- training seam (model.py:926-942 -> LatentDiffusion.forward / p_losses, ddpm3d.py:700-705,741-797) -----------------------
    def get_v(self, x, noise, t):
        return self._coef(self.sqrt_alphas_cumprod, t, x) * noise - self._coef(self.sqrt_one_minus_alphas_cumprod, t, x) * x

    def forward(self, x, c, **kwargs):
        """`loss, loss_dict = self.diffusion_model(x, c, fs=...)` of WorldModel.training_step: a uniformly drawn timestep per
        clip, the dynamic rescale of x (the 512 / 1024 models), then p_losses.  The U-Net below runs its differentiable walk
        (unet_train) whenever autograd is on and the module is in training mode."""
        t = torch.randint(0, self.num_timesteps, (x.shape[0],), device=self.device).long()
        if self.use_dynamic_rescale:
            x = x * self._coef(self.scale_arr, t, x)
        return self.p_losses(x, c, t, **kwargs)

    def p_losses(self, x_start, cond, t, noise=None, **kwargs):
        noise = torch.randn_like(x_start) if noise is None else noise
        x_noisy = self.q_sample(x_start=x_start, t=t, noise=noise)
        model_output = self.apply_model(x_noisy, t, cond, **kwargs)
        prefix = "train" if self.training else "val"
        target = noise if self.parameterization == "eps" else self.get_v(x_start, noise, t)
        loss_simple = torch.nn.functional.mse_loss(target, model_output, reduction="none").mean([1, 2, 3, 4])
        if torch.isnan(loss_simple).any():  # (ddpm3d.py:770-774: NaN clips are zeroed, not propagated)
            loss_simple = torch.where(torch.isnan(loss_simple), torch.zeros_like(loss_simple), loss_simple)
        loss_dict = {f"{prefix}/loss_simple": loss_simple.mean()}
        logvar_t = self.logvar.to(loss_simple.device)[t]
        loss = self.l_simple_weight * (loss_simple / torch.exp(logvar_t) + logvar_t).mean()
        loss_dict[f"{prefix}/loss"] = loss
        return loss, loss_dict

    def q_sample(self, x_start, t, noise=None):
        noise = torch.randn_like(x_start) if noise is None else noise
        return self._coef(self.sqrt_alphas_cumprod, t, x_start) * x_start + self._coef(self.sqrt_one_minus_alphas_cumprod, t, x_start) * noise
