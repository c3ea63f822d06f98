"""CPU (oracle op table): the cond / uncond pair of a CFG step as ONE U-Net forward over 2 x 16 frames (clips batched along
the rows, UNetModel._forward with b = 2; PANDORA_CFG_BATCH=1 in the sampler) equals the two separate forwards the reference
runs back to back (ddim.py:233-234) - per-frame ops see 32 frames, the frame-coupling ops ((T,H,W) GroupNorm, temporal conv,
temporal attention) and the cross-attention keep the clips apart."""
import pytest
import torch

from oracle import golden_recipe as gr
from oracle.ops_torch import TorchOps
from open_pandora_amd import synth
from open_pandora_amd.ddim import DDIMSampler
from open_pandora_amd.ddpm import LatentVisualDiffusion
from open_pandora_amd.unet import UNetModel
from test_oracle_golden import RH_KW, load, rel


def _model():
    torch.set_num_threads(8)
    m = UNetModel(**dict(RH_KW, model_channels=64)).eval()
    m.load_state_dict(synth.synth_state_dict(m, seed=gr.WEIGHT_SEED))
    return m.bind(TorchOps())


@pytest.mark.parametrize("L", [None, 77, 82], ids=["per_frame_image_tokens", "text_only", "shared_image_tokens"])
def test_batched_forward_equals_two_forwards(L):
    m = _model()
    ins, _, _ = gr.sampler_inputs(8, 8)
    x = torch.cat([ins["x_T"], ins["c_concat"]], 1)
    t, fs = torch.tensor([500]), torch.tensor([15])
    ca, cb = ins["c_crossattn"][:, :L], ins["uc_crossattn"][:, :L]
    ya, yb = m(x, t, context=ca, fs=fs), m(x, t, context=cb, fs=fs)
    y2 = m(torch.cat([x, x], 0), torch.cat([t, t]), context=torch.cat([ca, cb], 0), fs=fs)
    assert y2.shape == (2, 4, 16, 8, 8)
    assert rel(y2[0:1], ya) < 1e-5 and rel(y2[1:2], yb) < 1e-5 and rel(ya, yb) > 0.1
    if L is None:
        assert rel(ya, load("unet_small.npz")["mc64_8x8_t500"]) < 2e-5  # (and it is still the reference's forward)
    with pytest.raises(NotImplementedError):
        m(torch.cat([x, x], 0), torch.tensor([500, 499]), context=torch.cat([ca, cb], 0), fs=fs)
    with pytest.raises(NotImplementedError):  # per-clip fs used to get clip 0's embedding silently
        m(torch.cat([x, x], 0), torch.cat([t, t]), context=torch.cat([ca, cb], 0), fs=torch.tensor([15, 24]))


def test_sampler_in_batch_mode_equals_the_reference_trajectory(monkeypatch):
    S, eta, cfg = 5, 0.0, 4.0
    g = load("ddim_small.npz")[f"S{S}_eta{eta:g}_cfg{cfg:g}"]
    pm = LatentVisualDiffusion(_model())
    ins, cond, uc = gr.sampler_inputs(8, 8)
    shapes = []
    inner = pm.apply_model
    pm.apply_model = lambda x, t, c, **kw: (shapes.append(tuple(x.shape)), inner(x, t, c, **kw))[1]
    monkeypatch.setenv("PANDORA_CFG_BATCH", "1")
    y, _ = DDIMSampler(pm).sample(S=S, batch_size=1, shape=(4, 16, 8, 8), conditioning=cond, verbose=False,
                                  unconditional_guidance_scale=cfg, unconditional_conditioning=uc, eta=eta,
                                  fs=torch.tensor([15]), timestep_spacing="uniform_trailing", x_T=ins["x_T"])
    assert shapes == [(2, 4, 16, 8, 8)] * S  # ONE forward per step
    assert rel(y, g) < 5e-5


def test_batch_mode_falls_back_to_two_forwards_for_per_clip_kwargs_and_inference_tensors(monkeypatch):
    """(i) PANDORA_CFG_BATCH=1 with a features_adapter sized for ONE clip keeps the two-forward form (a forward
    over 2 T frames would fail the adapter's shape assert); (ii) conditions created under torch.inference_mode() have no
    version counter - the graph-key recipe must not read it."""
    from open_pandora_amd import ddim
    m = _model()
    pm = LatentVisualDiffusion(m)
    ins, cond, uc = gr.sampler_inputs(8, 8)
    feats = gr.adapter_features(64, 8, 8)
    monkeypatch.setenv("PANDORA_CFG_BATCH", "1")
    assert ddim._cfg_batchable(m.ops, ins["x_T"], cond, uc, {})
    assert not ddim._cfg_batchable(m.ops, ins["x_T"], cond, uc, {"features_adapter": feats})
    shapes = []
    inner = pm.apply_model
    pm.apply_model = lambda x, t, c, **kw: (shapes.append(tuple(x.shape)), inner(x, t, c, **kw))[1]
    kw = dict(S=2, batch_size=1, shape=(4, 16, 8, 8), conditioning=cond, verbose=False, unconditional_guidance_scale=4.0,
              unconditional_conditioning=uc, eta=0.0, fs=torch.tensor([15]), timestep_spacing="uniform_trailing", x_T=ins["x_T"])
    ya, _ = DDIMSampler(pm).sample(features_adapter=feats, **kw)
    assert shapes == [(1, 4, 16, 8, 8)] * 4
    monkeypatch.setenv("PANDORA_CFG_BATCH", "0")
    yb, _ = DDIMSampler(pm).sample(features_adapter=feats, **kw)
    assert torch.equal(ya, yb)
    with torch.inference_mode():
        v = torch.ones(3)
    assert ddim._version_of(v) == 0 and ddim._version_of(torch.ones(3)) == 0


def test_norm_sites_are_named_by_kind_and_pyramid_level():
    """UNetModel._site (the handle of HipOps(parity="selective")): every GroupNorm / LayerNorm / stream
    conversion call of a forward carries (kind, level) in `ops.site` - the level counted by the Down / Upsample modules passed,
    not derived from the latent size - and the attribute is cleared when the forward returns (first stage / Resampler calls belong to no
    site)."""
    m = _model()
    ops, seen = m.ops, set()
    for name in ("groupnorm", "ln_gemm", "conv3x3"):
        inner = getattr(ops, name)
        setattr(ops, name, (lambda f, n: lambda *a, **k: (seen.add((n, ops.site)), f(*a, **k))[1])(inner, name))
    ins = synth.synth_inputs(24, 8, 16, seed=gr.INPUT_SEED)  # 24 x 8 -> 12 x 4 -> 6 x 2 -> 3 x 1 (not a power of two)
    x = torch.cat([ins["x_T"], ins["c_concat"]], 1)
    m(x, torch.tensor([500]), context=ins["c_crossattn"], fs=torch.tensor([15]))
    assert ops.site is None
    gn = {s for n, s in seen if n == "groupnorm"}
    assert gn == {(k, lv) for k in ("gn3", "gnt", "gnp") for lv in range(4)}
    ln = {s for n, s in seen if n == "ln_gemm"}
    assert ln == {(f"ln{b}{i}", lv) for b in "st" for i in (1, 2, 3) for lv in range(4)}
    split = {s for n, s in seen if n == "conv3x3" and s is not None and s[0] == "split"}
    assert split == {("split", 0), ("split", 1), ("split", 2), ("split", 3)}  # stem + Downsample 0-2 | Upsample from 3, 2, 1
