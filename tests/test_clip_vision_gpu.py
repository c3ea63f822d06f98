"""GPU: the OpenCLIP ViT image tower on the HIP kernels (pm_gemm, pm_layernorm, pm_attention_generic with head dim 80)
against the oracle restatement and against the fixture of HF transformers' CLIPVisionModel (oracle/clip_vit_ref.py: what is pinned and what is not), reduced and
full ViT-H/14 size, and chained with the Resampler through wm.ImageContext."""
import pytest
import torch

from oracle import clip_vit_ref
from open_pandora_amd import synth, wm
from open_pandora_amd.clip_vision import VIT_H_14, FrozenOpenCLIPImageEmbedderV2
from test_clip_vision_cpu import SMALL
from test_unet_gpu import FWD_TOL, FWD_TOL_REDUCED

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.float().cpu() - b).norm() / b.norm()).item()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_tower_small(hip_ops_factory, dtype):
    m = FrozenOpenCLIPImageEmbedderV2(vision_cfg=SMALL)
    m.load_state_dict(synth.synth_state_dict(m, seed=7))
    img = torch.rand(2, 3, 70, 90, generator=torch.Generator().manual_seed(1)) * 2 - 1
    want = clip_vit_ref.vision_tower_forward({k: v.float() for k, v in m.state_dict().items()}, img, SMALL["heads"],
                                             SMALL["image_size"])
    got = m.bind(hip_ops_factory(dtype))(img.cuda())
    err = _rel(got, want)
    print(f"\n[parity] clip tower small {dtype}: rel err {err:.2e}")
    assert err <= FWD_TOL_REDUCED[dtype]


@pytest.mark.parametrize("dtype", [torch.float16])
def test_tower_vit_h_14_and_image_context(hip_ops_factory, dtype):
    """The shipped tower (632 M parameters, 257 tokens x 32 layers) and tower -> Resampler -> 256 x 1024 context
    tokens (model.py:710-712), with the unconditional tokens cached (model.py:728-729)."""
    from open_pandora_amd.resampler import Resampler
    ops = hip_ops_factory(dtype)
    with torch.device("meta"):
        m = FrozenOpenCLIPImageEmbedderV2()
    m.load_state_dict({k: synth.synth_tensor(k, tuple(v.shape), 11, "cuda") for k, v in m.state_dict().items()}, assign=True)
    img = torch.rand(1, 3, 320, 512, generator=torch.Generator().manual_seed(2)) * 2 - 1
    sd = {k: v.float().cpu() for k, v in m.state_dict().items()}
    want = clip_vit_ref.vision_tower_forward(sd, img, VIT_H_14["heads"])
    got = m.bind(ops)(img.cuda())
    err = _rel(got, want)
    print(f"\n[parity] clip tower ViT-H/14 {dtype}: rel err {err:.2e}")
    assert got.shape == (1, 257, 1280) and err <= 2 * FWD_TOL[dtype]  # 64 residual adds of 16-bit branch results
    rs = Resampler(dim=1024, depth=4, dim_head=64, heads=12, num_queries=16, embedding_dim=1280, output_dim=1024,
                   ff_mult=4, video_length=16)
    rs.load_state_dict(synth.synth_state_dict(rs, seed=12))
    ctx = wm.ImageContext(m, rs.bind(ops))
    tokens = ctx(img.cuda())
    assert tokens.shape == (1, 256, 1024) and torch.isfinite(tokens).all()
    u1, u2 = ctx.uncond(img.cuda()), ctx.uncond(img.cuda())
    assert u1 is u2 and u1.shape == (1, 256, 1024)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_tower_against_the_hf_clip_vision_model_fixture(hip_ops_factory, dtype):
    """The HIP tower at both sizes against the fixture of HF transformers' CLIPVisionModel (tests/golden/clip_vision_hf.npz,
    oracle/make_golden.py --clip-hf): a third-party implementation of the architecture, not the builder's restatement."""
    import os

    import numpy as np

    from oracle import golden_recipe as gr
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "clip_vision_hf.npz"))
    for tag, cfg, tol in (("small", gr.CLIP_SMALL, FWD_TOL_REDUCED[dtype]), ("vit_h_14", dict(VIT_H_14), 2 * FWD_TOL[dtype])):
        with torch.device("meta"):
            m = FrozenOpenCLIPImageEmbedderV2(vision_cfg=cfg)
        m.load_state_dict({k: synth.synth_tensor(k, tuple(v.shape), gr.CLIP_SEED, "cuda") for k, v in m.state_dict().items()},
                          assign=True)
        got = m.bind(hip_ops_factory(dtype))(gr.clip_image(tag).cuda())
        err = gr.compare_digest(got, g, tag, tol)[0]
        print(f"\n[parity] clip tower {tag} {dtype} vs HF CLIPVisionModel: rel err {err:.2e}")
        assert err <= tol
