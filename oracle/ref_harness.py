"""ORACLE tooling (build container only): import the REAL reference implementation from
/root/reference with harness-side shims; nothing under /root/reference is modified
or copied.  Used by oracle/make_golden.py and tests/test_oracle_vs_reference.py; absent on the GPU
box, where only the committed fixtures under tests/golden/ travel.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("PANDORA_REFERENCE", "/root/reference")
UNET_512 = dict(in_channels=8, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1],
                num_res_blocks=2, channel_mult=[1, 2, 4, 4], dropout=0.1, num_head_channels=64,
                transformer_depth=1, context_dim=1024, use_linear=True, use_checkpoint=False,
                temporal_conv=True, temporal_attention=True, temporal_selfatt_only=True,
                use_relative_position=False, use_causal_attention=False, temporal_length=16,
                addition_attention=True, image_cross_attention=True, default_fs=24, fs_condition=True)


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "DynamiCrafter", "lvdm"))


def _install_shims():
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tvu = types.ModuleType("torchvision.utils")
        tvu.make_grid = lambda *a, **k: None
        tv.utils = tvu
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.utils"] = tvu
    if "pytorch_lightning" not in sys.modules:
        import torch.nn as nn
        pl = types.ModuleType("pytorch_lightning")
        pl.LightningModule = nn.Module
        plu = types.ModuleType("pytorch_lightning.utilities")
        plu.rank_zero_only = lambda f: f
        pl.utilities = plu
        sys.modules["pytorch_lightning"] = pl
        sys.modules["pytorch_lightning.utilities"] = plu
    p = os.path.join(REFERENCE_ROOT, "DynamiCrafter")
    if p not in sys.path:
        sys.path.insert(0, p)


def reference_unet(**overrides):
    """Instantiate the reference UNetModel (openaimodel3d.py:284) with the shipped 512/1024 params."""
    _install_shims()
    from lvdm.modules.networks.openaimodel3d import UNetModel
    kw = dict(UNET_512)
    kw.update(overrides)
    return UNetModel(**kw).eval()


class AttrDict(dict):
    """Stand-in for OmegaConf nodes: item and attribute access, .get()."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return AttrDict(v) if isinstance(v, dict) else v


def reference_diffusion(unet_overrides=None, base_scale=0.7, unet_target="lvdm.modules.networks.openaimodel3d.UNetModel",
                        shell=None):
    """LatentVisualDiffusion (ddpm3d.py:1036) with Identity condition stages and first stage.  `shell`: overrides of the
    512 yaml's `model.params` (a value of None removes the key: the class default applies, as for the 256 yaml, which sets
    neither `parameterization` nor `rescale_betas_zero_snr` nor `use_dynamic_rescale`)."""
    _install_shims()
    import torch
    from lvdm.models.ddpm3d import LatentVisualDiffusion
    up = dict(UNET_512)
    up.update(unet_overrides or {})
    ident = AttrDict(target="torch.nn.Identity")
    cfg = dict(
        rescale_betas_zero_snr=True, parameterization="v", linear_start=0.00085, linear_end=0.012,
        num_timesteps_cond=1, timesteps=1000, first_stage_key="video", cond_stage_key="caption",
        cond_stage_trainable=False, conditioning_key="hybrid", image_size=[40, 64], channels=4,
        scale_by_std=False, scale_factor=0.18215, use_ema=False, uncond_type="empty_seq",
        use_dynamic_rescale=True, base_scale=base_scale, fps_condition_type="fps", perframe_ae=True,
        unet_config=AttrDict(target=unet_target, params=AttrDict(up)),
        first_stage_config=ident, cond_stage_config=ident, img_cond_stage_config=ident,
        image_proj_stage_config=ident)
    for k, val in (shell or {}).items():
        if val is None:
            cfg.pop(k, None)
        else:
            cfg[k] = val
    model = LatentVisualDiffusion(**cfg).eval()
    return model


def reference_sampler(model):
    """DDIMSampler (ddim.py:10) with the hard-coded `cuda` of register_buffer bypassed."""
    _install_shims()
    from lvdm.models.samplers.ddim import DDIMSampler

    class CPUSampler(DDIMSampler):
        def register_buffer(self, name, attr):
            setattr(self, name, attr)

    return CPUSampler(model)


def chunk_attention_over_frames(model, min_tokens=2304):
    """Harness-side memory shim for the 72x128 latent: the reference's eager CrossAttention.forward
    (attention.py:81-144) materialises a (frames*heads, N, N) f32 score tensor - 27 GB at N = 9216 - and then its
    softmax copy.  Attention is independent per batch element (= frame), so every CrossAttention module is
    wrapped to call ITS OWN unmodified forward once per frame and concatenate: same arithmetic per row, 1/16 of
    the peak memory.  Nothing under /root/reference is modified."""
    import torch
    from lvdm.modules.attention import CrossAttention
    for mod in model.modules():
        if isinstance(mod, CrossAttention) and not hasattr(mod, "_orig_forward"):
            orig = mod.forward
            mod._orig_forward = orig

            def fwd(x, context=None, mask=None, _o=orig):
                if x.shape[0] == 1 or x.shape[1] < min_tokens:
                    return _o(x, context=context, mask=mask)
                return torch.cat([_o(x[i:i + 1], context=None if context is None else context[i:i + 1], mask=mask)
                                  for i in range(x.shape[0])], 0)

            mod.forward = fwd
    return model
