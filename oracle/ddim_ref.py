"""ORACLE (test infrastructure, not product code): CPU restatement of the reference's diffusion
schedule and DDIM loop in plain f32 tensor arithmetic, functional style.

Pinned against the REAL reference (LatentVisualDiffusion + DDIMSampler imported from
/root/reference) by tests/test_oracle_vs_reference.py and the fixtures in tests/golden/.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

Reference call sites (relative to /root/reference/DynamiCrafter/lvdm):
  schedule_tables   models/ddpm3d.py:119-182,505-510; models/utils_diffusion.py:31-36,112-144
  ddim_tables       models/samplers/ddim.py:24-63; models/utils_diffusion.py:56-91
  ddim_sample       models/samplers/ddim.py:141-215 (loop), :218-290 (one step), ddpm3d.py:235-247;
                    `uncond_img` given: the three-way combine of models/samplers/ddim_multiplecond.py:229-236
"""
import numpy as np
import torch


def schedule_tables(timesteps=1000, linear_start=0.00085, linear_end=0.012, base_scale=0.7, turning_step=400,
                    zero_snr=True, dynamic_rescale=True):
    """zero_snr / dynamic_rescale False: the class defaults of ddpm3d.py:54-76 that the 256 yaml runs with (no
    rescale_zero_terminal_snr, scale array of ones = `use_dynamic_rescale: False`)."""
    betas = (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=torch.float64) ** 2).numpy()
    if zero_snr:  # zero-terminal-SNR rescale
        s = np.sqrt(np.cumprod(1.0 - betas, axis=0))
        s0, sT = s[0].copy(), s[-1].copy()
        s -= sT
        s *= s0 / (s0 - sT)
        bar = s ** 2
        alphas = np.concatenate([bar[0:1], bar[1:] / bar[:-1]])
        betas = 1 - alphas
    ac = np.cumprod(1.0 - betas, axis=0)
    bf = lambda a: torch.tensor(a, dtype=torch.bfloat16)
    scale_arr = np.concatenate((np.linspace(1.0, base_scale, turning_step), np.full(timesteps, base_scale)))
    if not dynamic_rescale:
        scale_arr = np.ones_like(scale_arr)
    return {"alphas_cumprod": bf(ac), "sqrt_alphas_cumprod": bf(np.sqrt(ac)),
            "sqrt_one_minus_alphas_cumprod": bf(np.sqrt(1.0 - ac)), "scale_arr": bf(scale_arr)}


def ddim_timesteps(method, S, T=1000):
    if method == "uniform":
        return np.asarray(list(range(0, T, T // S))) + 1
    if method == "uniform_trailing":
        return np.flip(np.round(np.arange(T, 0, -(T / S)))).astype(np.int64) - 1
    raise NotImplementedError(method)


def ddim_tables(tables, S, eta, spacing):
    ts = ddim_timesteps(spacing, S)
    alphacums = tables["alphas_cumprod"].to(torch.float32)
    alphas = alphacums[ts]                                                    # f32 tensor
    alphas_prev = np.asarray([alphacums[0]] + alphacums[ts[:-1]].tolist())   # float64 ndarray
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))  # f64 tensor
    scale = tables["scale_arr"][ts]
    scale_prev = torch.cat([scale[0:1], scale[:-1]])
    return {"timesteps": ts, "alphas": alphas, "alphas_prev": alphas_prev, "sigmas": sigmas,
            "scale": scale, "scale_prev": scale_prev}


@torch.no_grad()
def ddim_sample(apply_model, tables, x_T, cond, uncond, S, eta, cfg_scale, spacing="uniform_trailing",
                noises=None, fs=None, keep_pred_x0=False, guidance_rescale=0.0, uncond_img=None, cfg_img=None,
                parameterization="v", score_corrector=None, corrector_kwargs=None, noise_dropout=0.0):
    """apply_model(x, t, cond, fs) -> v prediction.  noises: list of S tensors (one per loop
    iteration, same shape as x_T) consumed when eta > 0.  Returns (x_0 sample, [pred_x0 per step]).
    score_corrector / corrector_kwargs / noise_dropout: ddim.py:248-250, 283-284."""
    d = ddim_tables(tables, S, eta, spacing)
    x = x_T.clone().float()
    b = x.shape[0]
    size = (b,) + (1,) * (x.dim() - 1)
    full = lambda v: torch.full(size, float(v), dtype=x.dtype)
    trace = []
    for i, step in enumerate(np.flip(d["timesteps"])):
        index = S - i - 1
        t = torch.full((b,), int(step), dtype=torch.long)
        e_c = apply_model(x, t, cond, fs)
        if uncond is None or cfg_scale == 1.0:
            v = e_c
        elif uncond_img is not None:  # multi-condition sampler (ddim_multiplecond.py:229-236): text on top of image guidance
            e_u = apply_model(x, t, uncond, fs)
            e_ui = apply_model(x, t, uncond_img, fs)
            ci = cfg_scale if cfg_img is None else cfg_img
            v = e_u + ci * (e_ui - e_u) + cfg_scale * (e_c - e_ui)
            if guidance_rescale > 0.0:
                dims = list(range(1, v.dim()))
                resc = v * (e_c.std(dim=dims, keepdim=True) / v.std(dim=dims, keepdim=True))
                v = guidance_rescale * resc + (1 - guidance_rescale) * v
        else:
            e_u = apply_model(x, t, uncond, fs)
            v = e_u + cfg_scale * (e_c - e_u)
            if guidance_rescale > 0.0:  # rescale_noise_cfg, utils_diffusion.py:147-158 (ddim.py:240-241)
                dims = list(range(1, v.dim()))
                resc = v * (e_c.std(dim=dims, keepdim=True) / v.std(dim=dims, keepdim=True))
                v = guidance_rescale * resc + (1 - guidance_rescale) * v
        if parameterization == "v":
            sa = tables["sqrt_alphas_cumprod"][t].reshape(size)         # bf16 scalars, promoted by x
            sm = tables["sqrt_one_minus_alphas_cumprod"][t].reshape(size)
            e_t = sa * v + sm * x
            pred_x0 = sa * x - sm * v
        else:  # eps (ddim.py:245-246,265-266): the model output IS e_t; x0 from the DDIM tables of this step
            e_t = v
            if score_corrector is not None:  # ddim.py:248-250 (asserted eps-only there)
                e_t = score_corrector.modify_score(None, e_t, x, t, cond, **(corrector_kwargs or {}))
            a_t = full(d["alphas"][index])
            pred_x0 = (x - full(torch.sqrt(1.0 - d["alphas"])[index]) * e_t) / a_t.sqrt()
        a_prev, sigma_t = full(d["alphas_prev"][index]), full(d["sigmas"][index])
        pred_x0 = pred_x0 * (full(d["scale_prev"][index]) / full(d["scale"][index]))
        dir_xt = (1.0 - a_prev - sigma_t ** 2).sqrt() * e_t
        noise = sigma_t * (noises[i].to(x.dtype) if noises is not None else torch.zeros_like(x))
        if noise_dropout > 0.0:  # ddim.py:283-284
            noise = torch.nn.functional.dropout(noise, p=noise_dropout)
        x = a_prev.sqrt() * pred_x0 + dir_xt + noise
        if keep_pred_x0:
            trace.append(pred_x0)
    return x, trace
