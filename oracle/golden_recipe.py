"""Seeds and synthetic-input recipes shared by oracle/make_golden.py (which runs the real reference)
and the tests that replay the same inputs through the oracle / the product.  Test infrastructure."""
import numpy as np

from open_pandora_amd import synth

WEIGHT_SEED, INPUT_SEED, NOISE_SEED = 20230211, 123, 7

UNET_SMALL_CASES = (("mc64_8x8_t500", 64, 8, 8, 500, 15), ("mc64_8x16_t999", 64, 8, 16, 999, 24),
                    ("mc128_8x8_t37", 128, 8, 8, 37, 3))
# context WITHOUT per-frame image tokens (openaimodel3d.py:565-566): total context length -> fixture tag
UNET_CTX_CASES = ((82, "ctx82_shared_image_tokens"), (77, "ctx77_text_only"))
# UNetModel params of configs/inference_256_v1.0.yaml:19-50 that differ from the 512 / 1024 yaml
UNET_256_OVERRIDES = dict(image_cross_attention_scale_learnable=True, default_fs=3)
# the 256 yaml's diffusion shell (configs/inference_256_v1.0.yaml:1-17): eps-prediction, no zero-terminal-SNR rescale, no
# dynamic rescale - the class defaults of ddpm3d.py:54-76 - cases (S, eta, cfg)
SHELL_256 = dict(parameterization=None, rescale_betas_zero_snr=None, use_dynamic_rescale=None, base_scale=None,
                 fps_condition_type=None, perframe_ae=None, image_size=[32, 32])
DDIM_EPS_CASES = ((5, 0.0, 4.0), (20, 1.0, 7.5))
# sampler options of p_sample_ddim (ddim.py:248-250 score_corrector, :283-284 noise_dropout) on the eps model: (S, eta, cfg, which)
DDIM_OPTION_CASES = ((10, 1.0, 4.0, "score_corrector"), (10, 1.0, 4.0, "noise_dropout"), (8, 1.0, 7.5, "both"))
CORRECTOR_KWARGS = dict(gain=0.9, mix=0.05)
NOISE_DROPOUT_P, DROPOUT_SEED = 0.25, 4242
DDIM_SMALL_CASES = ((5, 0.0, 4.0), (10, 0.0, 4.0), (20, 1.0, 4.0), (10, 0.0, 1.0), (10, 1.0, 4.0))
FRAMES_SMALL_CASES = ((5, 0.0), (50, 1.0))  # (S, eta): sampler -> decode_first_stage, cfg 4
DDIM_RESCALE_CASES = ((5, 0.0, 4.0, 0.7), (20, 1.0, 4.0, 0.3))  # (S, eta, cfg, guidance_rescale)
# the multi-condition sampler (ddim_multiplecond.py): (S, eta, cfg, cfg_img or None = "same as cfg", guidance_rescale)
DDIM_MULTICOND_CASES = ((5, 0.0, 4.0, 2.0, 0.0), (20, 1.0, 7.5, None, 0.7))

# (tag, constructor kwargs, x shape): a reduced Resampler and the shipped image_proj_stage_config
RESAMPLER_CASES = (("small", dict(dim=128, depth=2, dim_head=64, heads=2, num_queries=4, embedding_dim=192,
                                   output_dim=128, ff_mult=4, video_length=4), (2, 17, 192)),
                   ("full", dict(dim=1024, depth=4, dim_head=64, heads=12, num_queries=16, embedding_dim=1280,
                                 output_dim=1024, ff_mult=4, video_length=16), (1, 257, 1280)))


# image tower (clip_vision): seeded weights and images shared by oracle/make_golden.py --clip-hf and the tests
CLIP_SEED = 7
CLIP_SMALL = dict(image_size=56, patch_size=14, width=320, layers=3, heads=4, mlp_ratio=4.0, output_dim=64)  # head dim 80


def clip_image(tag):
    """(b, 3, H, W) in [-1, 1], NOT at the tower's input size: the preprocessing (resize + normalise) is part of the path."""
    b, h, w = (2, 70, 90) if tag == "small" else (1, 320, 512)
    return uniform_image(b, h, w, f"clip/{tag}")


def uniform_image(b, h, w, name):
    return synth.uniform_pm1(b * 3 * h * w, INPUT_SEED, name).reshape(b, 3, h, w)


def module_input(name, *shape):
    return (synth.uniform_pm1(int(np.prod(shape)), INPUT_SEED, name) * 3 ** 0.5).reshape(*shape)


def module_inputs():
    return {"x4": module_input("mod/x4", 16, 64, 4, 6), "x5": module_input("mod/x5", 1, 64, 16, 4, 6),
            "tok": module_input("mod/tok", 16, 24, 128), "ctx": module_input("mod/ctx", 16, 77 + 16, 1024),
            "emb": module_input("mod/emb", 16, 256)}


def multicond_uc_img(ins, cond, uc):
    """`unconditional_conditioning_img_nonetext` as model.py:737-743 builds it: the unconditional TEXT tokens followed by the
    conditional IMAGE tokens (text = first 77 tokens of c_crossattn)."""
    import torch
    return {"c_crossattn": [torch.cat([uc["c_crossattn"][0][:, :77], cond["c_crossattn"][0][:, 77:]], dim=1)],
            "c_concat": list(cond["c_concat"])}


def adapter_features(mc, h, w, T=16, channel_mult=(1, 2, 4, 4)):
    """`features_adapter` for UNetModel.forward (openaimodel3d.py:589-593): one (b t, C, H, W) tensor per input block id with
    (id + 1) % 3 == 0 - the last ResBlock of every level at num_res_blocks = 2 - seeded, a few tenths of the stream's scale."""
    feats = []
    for lvl, mult in enumerate(channel_mult):
        hh, ww = h, w
        for _ in range(lvl):
            hh, ww = (hh + 1) // 2, (ww + 1) // 2
        feats.append(0.3 * module_input(f"adapter/{lvl}", T, mc * mult, hh, ww))
    return feats


def sampler_inputs(h, w, T=16):
    ins = synth.synth_inputs(h, w, T, seed=INPUT_SEED)
    cond = {"c_crossattn": [ins["c_crossattn"]], "c_concat": [ins["c_concat"]]}
    uc = {"c_crossattn": [ins["uc_crossattn"]], "c_concat": [ins["c_concat"]]}
    return ins, cond, uc


def noises(shape, S):
    return [synth.synth_noise(shape, NOISE_SEED, i) for i in range(S)]


def digest_of(t, stride, n):
    flat = t.detach().float().reshape(-1)
    return flat[::int(stride)][:n]


def _is_prime(p):
    return p >= 2 and all(p % q for q in range(2, int(p ** 0.5) + 1))


def coprime_stride(shape, n):
    """Largest PRIME stride <= numel // n that divides no axis length: the flat samples k * stride then visit every
    residue of every axis (every column, row, frame, channel) - a stride sharing a factor with W only ever lands on
    W / gcd of the columns (72 on (..., 72, 128) saw columns 0, 8, ..., 120 only)."""
    numel = int(np.prod(shape))
    p = max(2, numel // int(n))
    while not (_is_prime(p) and all(d % p for d in shape if d > 1)):
        p -= 1
        if p < 2:
            return 1
    return p


FULL_LIMIT = 1 << 20  # tensors up to this many elements (every latent: <= 589 824) are committed whole


def make_digest(t, n=4096):
    """Fixture record of a tensor too big to commit: the WHOLE tensor when it has <= FULL_LIMIT elements, else a slice
    at a prime stride coprime with every axis; always the global moments and the per-column / per-row profiles (rms and mean
    over all other axes for every index of the last two axes), so that a defect confined to some columns or rows of a tile
    edge changes a number the tests assert."""
    import torch
    x = t.detach().to(torch.float64)
    flat = x.reshape(-1)
    rec = {"mean": np.float64(flat.mean()), "std": np.float64(flat.std()), "absmax": np.float64(flat.abs().max()),
           "numel": np.int64(flat.numel()), "shape": np.asarray(x.shape, np.int64)}
    if flat.numel() <= FULL_LIMIT:
        rec["full"] = t.detach().float().numpy().copy()
    stride = coprime_stride(tuple(x.shape), n)
    rec["slice"] = flat[::stride][:].float().numpy().copy()
    rec["stride"] = np.int64(stride)
    if x.dim() >= 2:
        cols = x.reshape(-1, x.shape[-1])
        rows = x.reshape(-1, x.shape[-2], x.shape[-1])
        rec["col_rms"], rec["col_mean"] = cols.pow(2).mean(0).sqrt().numpy(), cols.mean(0).numpy()
        rec["row_rms"], rec["row_mean"] = rows.pow(2).mean((0, 2)).sqrt().numpy(), rows.mean((0, 2)).numpy()
    return rec


def _rel(a, b):
    import torch
    a, b = torch.as_tensor(np.asarray(a)).double(), torch.as_tensor(np.asarray(b)).double()
    return float((a - b).norm() / b.norm())


def compare_digest(t, g, key, tol):
    """Relative L2 error of tensor `t` against fixture record `key` of archive `g` (whole tensor when the record holds it,
    else its strided slice), AFTER asserting everything else the record stores to `tol`: std and mean (in units of std),
    absmax (3 tol: one element), and - records of format 2 - the per-column and per-row rms / mean profiles.  Returns
    (err, std, fixture std).  Old records (stride from numel // n, no profiles) are still read."""
    import torch
    y = t.detach().float().cpu()
    yd = y.double()
    gstd, std = float(g[f"{key}/std"]), float(yd.std())
    assert abs(std - gstd) <= tol * gstd, f"{key}: std {std} vs {gstd}"
    assert abs(float(yd.mean()) - float(g[f"{key}/mean"])) <= tol * gstd, f"{key}: mean"
    assert abs(float(yd.abs().max()) - float(g[f"{key}/absmax"])) <= 3 * tol * float(g[f"{key}/absmax"]), f"{key}: absmax"
    assert int(g[f"{key}/numel"]) == y.numel()
    if f"{key}/col_rms" in g:
        assert tuple(int(d) for d in g[f"{key}/shape"]) == tuple(y.shape)
        cols, rows = yd.reshape(-1, y.shape[-1]), yd.reshape(-1, y.shape[-2], y.shape[-1])
        for name, got in (("col_rms", cols.pow(2).mean(0).sqrt()), ("row_rms", rows.pow(2).mean((0, 2)).sqrt())):
            e = _rel(got, g[f"{key}/{name}"])
            assert e <= tol, f"{key}/{name}: {e:.2e} > {tol:.1e}"
        for name, got in (("col_mean", cols.mean(0)), ("row_mean", rows.mean((0, 2)))):
            # a profile of means: L2 against the larger of its own norm and the noise floor std sqrt(n), and no single entry
            # off by more than 3 tol std (a 16-bit output's rounding is correlated down a column whose entries share a binade:
            # up to half an ulp of the column mean survives the average - measured 8e-3 std on the bf16 Resampler output)
            ref = torch.as_tensor(g[f"{key}/{name}"])
            e2 = float((got - ref).norm() / max(float(ref.norm()), gstd * ref.numel() ** 0.5))
            assert e2 <= tol, f"{key}/{name}: {e2:.2e} > {tol:.1e}"
            e = float((got - ref).abs().max())
            assert e <= 3 * tol * gstd, f"{key}/{name}: {e:.2e} > 3 x {tol:.1e} x std"
    if f"{key}/full" in g:
        err = _rel(y, g[f"{key}/full"])
    else:
        stride, sl = int(g[f"{key}/stride"]), g[f"{key}/slice"]
        err = _rel(digest_of(y, stride, len(sl)), sl)
    return err, std, gstd


def ae_latent(T, h, w):
    """scaled latent (1, 4, T, h, w) as the sampler returns it (std ~ scale_factor)."""
    n = 4 * T * h * w
    return (synth.uniform_pm1(n, INPUT_SEED, f"ae/z/{T}x{h}x{w}") * 3 ** 0.5 * 0.18215).reshape(1, 4, T, h, w)


def ae_pixels(n, H, W):
    """conditioning frames (n, 3, H, W) in [-1, 1]."""
    return synth.uniform_pm1(n * 3 * H * W, INPUT_SEED, f"ae/x/{n}x{H}x{W}").reshape(n, 3, H, W)



class RecipeCorrector:
    """A deterministic `score_corrector` (the reference only asks for modify_score(model, e_t, x, t, c, **kwargs), ddim.py:250)."""

    def modify_score(self, model, e_t, x, t, c, gain=1.0, mix=0.0):
        return gain * e_t + mix * x


class RecipeDropout:
    """Stands in for torch.nn.functional.dropout while a fixture is generated AND while it is checked: the k-th call keeps the
    elements of a mask drawn from a generator seeded with DROPOUT_SEED + k and scales them by 1 / (1 - p) - what dropout does,
    without tying the fixture to the order in which a sampler consumes the global RNG."""

    def __init__(self):
        self.calls = 0

    def __call__(self, input, p=0.5, training=True, inplace=False):
        import torch
        if not training or p == 0.0:  # (nn.Dropout modules of an eval-mode network land here too)
            return input
        g = torch.Generator().manual_seed(DROPOUT_SEED + self.calls)
        self.calls += 1
        keep = (torch.rand(input.shape, generator=g) >= p).to(device=input.device, dtype=input.dtype)
        return input * keep / (1.0 - p)
