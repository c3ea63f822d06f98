"""Build-container only: the oracle against the LIVE reference implementation imported from
/root/reference (skipped where the checkout is absent, e.g. on the GPU box - the same comparison is
then carried by the committed fixtures, tests/test_oracle_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import ref_harness as rh

pytestmark = pytest.mark.skipif(not rh.available(), reason="reference checkout not present")


def test_oracle_unet_matches_live_reference():
    from oracle.unet_ref import unet_forward
    from open_pandora_amd import synth
    ref = rh.reference_unet(model_channels=64)
    sd = synth.synth_state_dict(ref, seed=5)
    ref.load_state_dict(sd)
    ins = synth.synth_inputs(8, 8, 16, seed=9)
    x = torch.cat([ins["x_T"], ins["c_concat"]], 1)
    ts, fs = torch.tensor([321]), torch.tensor([8])
    with torch.no_grad():
        want = ref(x, ts, context=ins["c_crossattn"], fs=fs)
    got = unet_forward(sd, x, ts, ins["c_crossattn"], fs, model_channels=64)
    assert ((got - want).norm() / want.norm()).item() < 2e-5


def test_fixture_regeneration_is_reproducible(tmp_path, monkeypatch):
    """The committed schedule fixture is exactly what the reference produces today."""
    import numpy as np
    import os
    from oracle import make_golden as mg
    monkeypatch.setattr(mg, "GOLD", str(tmp_path))
    mg.gen_schedule()
    new = np.load(tmp_path / "schedule.npz")
    old = np.load(os.path.join(os.path.dirname(__file__), "golden", "schedule.npz"))
    assert sorted(new.files) == sorted(old.files)
    for k in new.files:
        assert np.array_equal(new[k], old[k], equal_nan=True), k


def test_multicond_sampler_is_dead_in_the_reference_and_its_working_form_is_ours(tmp_path, monkeypatch):
    """`multiple_cond_cfg=True` selects ddim_multiplecond.DDIMSampler (model.py:705), whose make_schedule
    runs np.sqrt on the bf16 `alphas_cumprod` buffer and raises - as shipped, the class is dead.  Its working form (its own
    sampling code on the main sampler's make_schedule, regenerated here live) is what the product's DDIMSamplerMultiCond
    and the committed fixture restate."""
    rh._install_shims()
    import lvdm.models.samplers.ddim_multiplecond as refmc

    class CPUSampler(refmc.DDIMSampler):
        def register_buffer(self, name, attr):
            setattr(self, name, attr)

    m = rh.reference_diffusion(dict(model_channels=64))
    assert m.alphas_cumprod.dtype == torch.bfloat16
    with pytest.raises(TypeError, match="BFloat16"):
        CPUSampler(m).make_schedule(5, "uniform_trailing", 0.0, verbose=False)
    import os
    from oracle import golden_recipe as gr
    from oracle import make_golden as mg
    monkeypatch.setattr(mg, "GOLD", str(tmp_path))
    monkeypatch.setattr(gr, "DDIM_MULTICOND_CASES", gr.DDIM_MULTICOND_CASES[:1])
    mg.gen_ddim_multicond()
    new = np.load(tmp_path / "ddim_small_multicond.npz")
    old = np.load(os.path.join(os.path.dirname(__file__), "golden", "ddim_small_multicond.npz"))
    for k in new.files:  # (to f32 rounding: the reference's reductions depend on the thread count the session happens to run with)
        assert float(np.linalg.norm(new[k] - old[k]) / np.linalg.norm(old[k])) < 1e-5, k
