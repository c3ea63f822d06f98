"""CPU: first-stage decoder - the oracle (oracle/ae_ref.py) and the product graph (autoencoder.py on the
oracle's TorchOps) against fixtures captured from the real AutoencoderKL."""
import os

import numpy as np
import pytest
import torch

from oracle import ae_ref, golden_recipe as gr
from oracle.ops_torch import TorchOps
from open_pandora_amd import synth
from open_pandora_amd.autoencoder import DDCONFIG, AutoencoderKL
from test_oracle_golden import load, rel

CASES = (("ch32_3x8x8", 32, 3, 8, 8), ("ch64_2x8x16", 64, 2, 8, 16))


@pytest.mark.parametrize("tag,ch,T,h,w", CASES)
def test_ae_decode_small(tag, ch, T, h, w):
    g = load("ae_decode_small.npz")[tag]
    ae = AutoencoderKL(ddconfig=dict(DDCONFIG, ch=ch))
    sd = synth.synth_state_dict(ae, seed=gr.WEIGHT_SEED)
    z = gr.ae_latent(T, h, w)
    assert rel(ae_ref.ae_decode(sd, z), g) < 2e-5
    ae.load_state_dict(sd)
    y = ae.bind(TorchOps()).decode_first_stage(z)
    assert y.shape == (1, 3, T, 8 * h, 8 * w) and rel(y, g) < 2e-5
    # encoder: posterior moments of T frames, then the reparameterised, scaled sample
    gm = load("ae_decode_small.npz")["enc/" + tag]
    px = gr.ae_pixels(T, 8 * h, 8 * w)
    assert rel(ae_ref.ae_encode_moments(sd, px), gm) < 2e-5
    mom = ae.encode_moments(px)
    assert mom.shape == (T, 8, h, w) and rel(mom, gm) < 2e-5
    noise = gr.ae_latent(T, h, w)[0].permute(1, 0, 2, 3)
    assert rel(ae.encode_first_stage(px, noise), ae_ref.ae_sample_latent(torch.as_tensor(gm), noise)) < 2e-5


def test_ae_state_dict_contract_and_guards():
    with torch.device("meta"):
        ae = AutoencoderKL()
    sd = ae.state_dict()
    assert len(sd) == 248 and sum(v.numel() for v in sd.values()) == 83653863
    assert tuple(sd["decoder.mid.attn_1.q.weight"].shape) == (512, 512, 1, 1)
    assert tuple(sd["decoder.up.1.upsample.conv.weight"].shape) == (256, 256, 3, 3)
    assert tuple(sd["encoder.down.0.downsample.conv.weight"].shape) == (128, 128, 3, 3)
    with pytest.raises(RuntimeError):
        AutoencoderKL(ddconfig=dict(DDCONFIG, ch=32)).decode(torch.zeros(1, 4, 8, 8))
    with pytest.raises(RuntimeError):
        AutoencoderKL(ddconfig=dict(DDCONFIG, ch=32)).encode_moments(torch.zeros(1, 3, 64, 64))
