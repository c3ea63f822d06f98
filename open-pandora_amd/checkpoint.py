"""Checkpoint wire formats of the reference, mapped onto the drop-in modules.

The reference moves weights around in four shapes:
  * Lightning / upstream DynamiCrafter `model.ckpt`: `{"state_dict": {...}}` with keys
    `model.diffusion_model.*` (U-Net), `first_stage_model.*` (AutoencoderKL), `cond_stage_model.*`,
    `embedder.*`, `image_proj_model.*`; the 256x256 release still says `framestride_embed` where the code
    has `fps_embedding` (scripts/evaluation/inference.py:27-52);
  * DeepSpeed stage-2 shards: `{"module": {...}}` whose keys carry a 16-character
    `_forward_module.` prefix (inference.py:46-50, tools/ckpt2bin.py:14, tools/pt2bin.py:13);
  * Open-Pandora `pytorch_model.bin` (HF `from_pretrained`, model.py:486-487,610-613): one flat dict
    for the whole world model where the diffusion part sits under `diffusion_model.` (model.py:599), i.e.
    the U-Net under `diffusion_model.model.diffusion_model.`;
  * a bare U-Net / AutoencoderKL state_dict (what `UNetModel.state_dict()` returns).
`split_checkpoint` normalises any of them; `load_unet` / `load_autoencoder` fill the MI355X modules
(strict: the key sets are identical by construction, tests/test_graph_cpu.py, tests/test_ae_cpu.py).
"""
from collections import OrderedDict

UNET_PREFIXES = ("diffusion_model.model.diffusion_model.", "model.diffusion_model.")
AE_PREFIXES = ("diffusion_model.first_stage_model.", "first_stage_model.")


def normalise_keys(obj):
    """Unwrap {'state_dict'|'module': ...}, strip `_forward_module.` / `module.`, apply the
    framestride_embed -> fps_embedding rename."""
    sd = obj
    for wrapper in ("state_dict", "module"):
        if isinstance(sd, dict) and wrapper in sd and isinstance(sd[wrapper], dict):
            sd = sd[wrapper]
    out = OrderedDict()
    for k, v in sd.items():
        k = k.replace("_forward_module.", "")
        if k.startswith("module."):
            k = k[len("module."):]
        out[k.replace("framestride_embed", "fps_embedding")] = v
    return out


def _take(sd, prefixes):
    for p in prefixes:
        sub = OrderedDict((k[len(p):], v) for k, v in sd.items() if k.startswith(p))
        if sub:
            return sub
    return None


def split_checkpoint(obj):
    """-> {'unet': state_dict or None, 'first_stage': state_dict or None, 'rest': other keys}."""
    sd = normalise_keys(obj)
    unet = _take(sd, UNET_PREFIXES)
    ae = _take(sd, AE_PREFIXES)
    if unet is None and any(k.startswith("input_blocks.") for k in sd):  # bare U-Net
        unet = OrderedDict((k, v) for k, v in sd.items() if not k.startswith(("encoder.", "decoder.", "quant_conv", "post_quant_conv")))
    if ae is None and any(k.startswith("decoder.") for k in sd) and not any(k.startswith("input_blocks.") for k in sd):
        ae = sd
    used = tuple(UNET_PREFIXES + AE_PREFIXES)
    rest = OrderedDict((k, v) for k, v in sd.items() if not k.startswith(used)) if (unet is not sd and ae is not sd) else OrderedDict()
    return {"unet": unet, "first_stage": ae, "rest": rest}


def load_unet(unet, obj, strict=True):
    parts = split_checkpoint(obj)
    if parts["unet"] is None:
        raise KeyError("no U-Net weights found (expected keys under model.diffusion_model. or diffusion_model.model.diffusion_model.)")
    return unet.load_state_dict(parts["unet"], strict=strict)


def load_autoencoder(ae, obj, strict=True):
    parts = split_checkpoint(obj)
    if parts["first_stage"] is None:
        raise KeyError("no first-stage weights found (expected keys under first_stage_model.)")
    sd = OrderedDict((k, v) for k, v in parts["first_stage"].items() if not k.startswith("loss."))
    return ae.load_state_dict(sd, strict=strict)
