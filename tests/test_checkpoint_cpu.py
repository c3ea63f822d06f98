"""CPU: the reference's checkpoint wire formats (Lightning state_dict, DeepSpeed module shard with the
`_forward_module.` prefix, Open-Pandora pytorch_model.bin, the 256-model `framestride_embed` rename)
load into the drop-in U-Net / AutoencoderKL (inference.py:27-52, tools/ckpt2bin.py:14, model.py:599)."""
import torch

from open_pandora_amd import checkpoint
from open_pandora_amd.autoencoder import DDCONFIG, AutoencoderKL
from open_pandora_amd.unet import UNetModel
from test_oracle_golden import RH_KW


def _models():
    unet = UNetModel(**dict(RH_KW, model_channels=64))
    ae = AutoencoderKL(ddconfig=dict(DDCONFIG, ch=32))
    return unet, ae


def test_wire_formats_round_trip():
    unet, ae = _models()
    g = torch.Generator().manual_seed(1)
    usd = {k: torch.randn(v.shape, generator=g) for k, v in unet.state_dict().items()}
    asd = {k: torch.randn(v.shape, generator=g) for k, v in ae.state_dict().items()}
    lightning = {"state_dict": {**{"model.diffusion_model." + k: v for k, v in usd.items()},
                                **{"first_stage_model." + k: v for k, v in asd.items()},
                                "cond_stage_model.dummy": torch.zeros(1), "scale_arr": torch.zeros(3)}}
    deepspeed = {"module": {"_forward_module." + k: v for k, v in lightning["state_dict"].items()}}
    pandora = {**{"diffusion_model.model.diffusion_model." + k: v for k, v in usd.items()},
               **{"diffusion_model.first_stage_model." + k: v for k, v in asd.items()},
               "video_model.lm_head.weight": torch.zeros(2, 2)}
    legacy = {"state_dict": {"model.diffusion_model." + k.replace("fps_embedding", "framestride_embed"): v
                             for k, v in usd.items()}}
    def poison(m):  # so that a load that skipped a tensor cannot pass the equality check below
        with torch.no_grad():
            for p in m.parameters():
                p.fill_(float("nan"))

    for blob in (lightning, deepspeed, pandora, legacy, usd):
        poison(unet)
        res = checkpoint.load_unet(unet, blob)
        assert not res.missing_keys and not res.unexpected_keys
        cur = unet.state_dict()
        assert all(torch.equal(cur[k], usd[k]) for k in usd)
    for blob in (lightning, deepspeed, pandora, asd):
        poison(ae)
        res = checkpoint.load_autoencoder(ae, blob)
        assert not res.missing_keys and not res.unexpected_keys
        cur = ae.state_dict()
        assert all(torch.equal(cur[k], asd[k]) for k in asd)
    parts = checkpoint.split_checkpoint(pandora)
    assert list(parts["rest"]) == ["video_model.lm_head.weight"]
