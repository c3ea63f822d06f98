"""Model assembly for the two shipped resolutions (DynamiCrafter/configs/inference_{512,1024}_v1.0.yaml
`model.params`): U-Net hyper-parameters, diffusion shell, op-table binding, seeded synthetic weights."""
import torch
import yaml

from . import synth
from .ddpm import LatentVisualDiffusion
from .unet import UNetModel

UNET_PARAMS = dict(in_channels=8, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1],
                   num_res_blocks=2, channel_mult=[1, 2, 4, 4], dropout=0.1, num_head_channels=64,
                   transformer_depth=1, context_dim=1024, use_linear=True, use_checkpoint=True,
                   temporal_conv=True, temporal_attention=True, temporal_selfatt_only=True,
                   use_relative_position=False, use_causal_attention=False, temporal_length=16,
                   addition_attention=True, image_cross_attention=True, default_fs=24, fs_condition=True)
RESOLUTIONS = {"320x512": dict(image_size=(40, 64), base_scale=0.7, default_fs=24),
               "576x1024": dict(image_size=(72, 128), base_scale=0.3, default_fs=10)}


def params_from_yaml(path):
    """Read `model.params` of a DynamiCrafter inference yaml (the reference goes through OmegaConf +
    instantiate_from_config, DynamiCrafter/utils/utils.py:27-42; PyYAML is enough for the values)."""
    with open(path) as f:
        cfg = yaml.safe_load(f)["model"]["params"]
    return cfg


def build_diffusion(resolution="320x512", ops=None, seed=20230211, unet_overrides=None, yaml_path=None,
                    weights="synthetic"):
    """LatentVisualDiffusion with a U-Net built on the meta device and filled on `ops.device` with the
    seeded synthesiser (weights='synthetic') or left for load_state_dict (weights='empty')."""
    r = dict(RESOLUTIONS[resolution])
    up = dict(UNET_PARAMS, default_fs=r.pop("default_fs"))
    shell = dict(linear_start=0.00085, linear_end=0.012, timesteps=1000, parameterization="v",
                 rescale_betas_zero_snr=True, conditioning_key="hybrid", use_dynamic_rescale=True,
                 scale_factor=0.18215, channels=4)
    if yaml_path is not None:
        y = params_from_yaml(yaml_path)
        up = dict(y["unet_config"]["params"])
        # what a yaml does not set takes the reference CLASS defaults (ddpm3d.py:54-76: eps-prediction, no zero-terminal-SNR
        # rescale, no dynamic rescale) - the 256 yaml sets none of the three - not the 512 yaml's values
        shell.update(parameterization="eps", rescale_betas_zero_snr=False, use_dynamic_rescale=False)
        for k in list(shell) + ["base_scale", "image_size"]:
            if k in y:
                (r if k in ("base_scale", "image_size") else shell)[k] = y[k]
    up.update(unet_overrides or {})
    device = "cpu" if ops is None else ops.device
    with torch.device("meta"):
        unet = UNetModel(**up)
    if weights == "synthetic":
        shapes = {k: tuple(v.shape) for k, v in unet.state_dict().items()}
        sd = {k: synth.synth_tensor(k, s, seed, device) for k, s in shapes.items()}
        unet.load_state_dict(sd, assign=True)
    else:
        unet.to_empty(device=device)
    unet.eval()
    if ops is not None:
        unet.bind(ops)
    return LatentVisualDiffusion(unet, **shell, **r)
