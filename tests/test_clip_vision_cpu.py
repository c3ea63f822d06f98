"""CPU: the OpenCLIP ViT image tower (open_pandora_amd.clip_vision) on the oracle's op table against
the oracle restatement (oracle/clip_vit_ref.py - parity UNPINNED: open_clip is a third-party package absent here),
and against torch's own nn.MultiheadAttention / Conv2d modules holding the same parameters (an independent
formulation of the published architecture)."""
import torch

from oracle import clip_vit_ref
from oracle.ops_torch import TorchOps
from open_pandora_amd import synth
from open_pandora_amd.clip_vision import VIT_H_14, FrozenOpenCLIPImageEmbedderV2

SMALL = dict(image_size=56, patch_size=14, width=320, layers=3, heads=4, mlp_ratio=4.0, output_dim=64)  # head dim 80


def _tower(cfg):
    m = FrozenOpenCLIPImageEmbedderV2(vision_cfg=cfg)
    m.load_state_dict(synth.synth_state_dict(m, seed=7))
    return m


def test_state_dict_keys_of_the_shipped_tower():
    with torch.device("meta"):
        m = FrozenOpenCLIPImageEmbedderV2()
    sd = m.state_dict()
    assert sd["model.visual.conv1.weight"].shape == (1280, 3, 14, 14)
    assert sd["model.visual.positional_embedding"].shape == (257, 1280)
    assert sd["model.visual.transformer.resblocks.31.attn.in_proj_weight"].shape == (3840, 1280)
    assert sd["model.visual.transformer.resblocks.0.mlp.c_fc.weight"].shape == (5120, 1280)
    assert sd["model.visual.proj"].shape == (1280, 1024) and len(sd) == 4 + 2 + 2 + 1 + 32 * 12 - 1  # 392 tensors
    assert VIT_H_14["width"] // VIT_H_14["heads"] == 80


def test_tower_graph_matches_oracle_and_torch_modules():
    m = _tower(SMALL).bind(TorchOps())
    img = torch.rand(2, 3, 70, 90, generator=torch.Generator().manual_seed(1)) * 2 - 1
    got = m(img)
    sd = {k: v.float() for k, v in m.state_dict().items()}
    want = clip_vit_ref.vision_tower_forward(sd, img, SMALL["heads"], SMALL["image_size"])
    assert got.shape == (2, 17, 320) and ((got - want).norm() / want.norm()).item() < 2e-5
    # independent formulation: torch's own MultiheadAttention forward on the same parameters
    vis = m.model.visual
    x = clip_vit_ref.preprocess(img, SMALL["image_size"])
    x = vis.conv1(x).flatten(2).permute(0, 2, 1)
    x = torch.cat([vis.class_embedding + torch.zeros(2, 1, 320), x], 1) + vis.positional_embedding
    x = vis.ln_pre(x).permute(1, 0, 2)
    for blk in vis.transformer.resblocks:
        y = blk.ln_1(x)
        x = x + blk.attn(y, y, y, need_weights=False)[0]
        x = x + blk.mlp(blk.ln_2(x))
    assert ((got - x.permute(1, 0, 2)).norm() / want.norm()).item() < 2e-5


def test_oracle_and_product_against_the_hf_clip_vision_model_fixture():
    """r05: the tower pinned to a third-party implementation that IS in the image - HF transformers' CLIPVisionModel, the
    class that loads the laion ViT-H-14 conversion of the checkpoint the reference pulls through open_clip
    (`last_hidden_state` = all tokens of the last block, no post-LN, no projection = condition.py:353-382) - on seeded weights
    in open_clip's key layout (oracle/make_golden.py --clip-hf).  The oracle restatement at both sizes, the product graph on
    the oracle's op table at the reduced size (the full-size product runs on the GPU)."""
    import os

    import numpy as np

    from oracle import golden_recipe as gr
    from open_pandora_amd.clip_vision import VIT_H_14
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "clip_vision_hf.npz"))
    assert "CLIPVisionModel" in str(g["source"])
    for tag, cfg in (("small", gr.CLIP_SMALL), ("vit_h_14", dict(VIT_H_14))):
        with torch.device("meta"):
            shapes = {k: tuple(v.shape) for k, v in FrozenOpenCLIPImageEmbedderV2(vision_cfg=cfg).state_dict().items()}
        sd = synth.synth_state_dict(shapes, seed=gr.CLIP_SEED)
        img = gr.clip_image(tag)
        y = clip_vit_ref.vision_tower_forward(sd, img, cfg["heads"], cfg["image_size"])
        err = gr.compare_digest(y, g, tag, 2e-5)[0]
        assert err < 2e-5, (tag, err)
        if tag == "small":
            m = FrozenOpenCLIPImageEmbedderV2(vision_cfg=cfg)
            m.load_state_dict(sd)
            assert gr.compare_digest(m.bind(TorchOps())(img), g, tag, 2e-5)[0] < 2e-5
