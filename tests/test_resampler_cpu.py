"""CPU: image-context Resampler - the oracle (oracle/resampler_ref.py) and the product graph
(resampler.py on the oracle's TorchOps) against fixtures captured from the real
lvdm.modules.encoders.resampler.Resampler (oracle/make_golden.py --resampler)."""
import pytest
import torch

from oracle import golden_recipe as gr, ref_harness as rh, resampler_ref
from oracle.ops_torch import TorchOps
from open_pandora_amd import synth
from open_pandora_amd.resampler import Resampler
from test_oracle_golden import load, rel


def _check(y, g, tag):
    if tag == "small":
        return rel(y, g["small"])
    return gr.compare_digest(y, g, tag, 2e-5)[0]


@pytest.mark.parametrize("tag,kw,xs", gr.RESAMPLER_CASES, ids=[c[0] for c in gr.RESAMPLER_CASES])
def test_resampler_against_reference(tag, kw, xs):
    g = load("resampler.npz")
    m = Resampler(**kw)
    sd = synth.synth_state_dict(m, seed=gr.WEIGHT_SEED)
    x = gr.module_input(f"resampler/{tag}", *xs)
    assert _check(resampler_ref.resampler_forward(sd, x, kw["heads"]), g, tag) < 2e-5
    m.load_state_dict(sd)
    y = m.bind(TorchOps())(x)
    nq = kw["num_queries"] * kw["video_length"]
    assert y.shape == (xs[0], nq, kw["output_dim"]) and _check(y, g, tag) < 2e-5


def test_resampler_contract():
    with pytest.raises(RuntimeError, match="no op table"):
        Resampler(dim=128, depth=1, heads=2, num_queries=4, embedding_dim=64, output_dim=128)(torch.zeros(1, 3, 64))
    with pytest.raises(NotImplementedError):
        Resampler(dim_head=80)


@pytest.mark.skipif(not rh.available(), reason="reference checkout not present")
def test_resampler_state_dict_matches_reference_keys():
    rh._install_shims()
    from lvdm.modules.encoders.resampler import Resampler as Ref
    kw = gr.RESAMPLER_CASES[0][1]
    ours, ref = Resampler(**kw).state_dict(), Ref(**kw).state_dict()
    assert list(ours) == list(ref) and all(ours[k].shape == ref[k].shape for k in ref)


def test_image_context_caches_unconditional_tokens():
    """model.py:711-712,728-729: tower -> Resampler; the zero-image tokens are computed once per shape."""
    from open_pandora_amd.wm import ImageContext
    kw = gr.RESAMPLER_CASES[0][1]
    m = Resampler(**kw)
    m.load_state_dict(synth.synth_state_dict(m, seed=gr.WEIGHT_SEED))
    m.bind(TorchOps())
    calls = []

    def tower(img):  # stand-in for the OpenCLIP tower: (b, 3, H, W) -> (b, 17, 192) tokens
        calls.append(float(img.abs().sum()))
        return gr.module_input("resampler/small", 2, 17, 192)[: img.shape[0]] * (1.0 + img.mean())

    ctx = ImageContext(tower, m)
    img = torch.ones(1, 3, 8, 8)
    a, u1, u2 = ctx(img), ctx.uncond(img), ctx.uncond(img)
    assert a.shape == (1, 16, 128) and u1 is u2 and calls == [192.0, 0.0]
    assert not torch.equal(a, u1)
