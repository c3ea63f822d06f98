"""GPU: where the whole-path error comes from (VERDICT r01 "close the tolerance gap or prove where it comes from").

Three machines run the SAME graph (open_pandora_amd.unet.UNetModel) on the same seeded weights and inputs:
  reference   the real lvdm UNetModel / DDIMSampler in f32 on CPU (committed goldens, oracle/make_golden.py);
  ideal16     exact f32 arithmetic, but every tensor the HIP path stores as 16 bit - the MFMA operands:
              normalised activations, q/k/v, attention probabilities and outputs, GEGLU products - rounded to
              that type (oracle TorchOps(model_16bit=True)): the error INHERENT to 16-bit matrix operands, no
              kernel involved;
  hip         the gfx950 kernels.
The reference -> ideal16 distance is the floor of the design (about 1.3e-3 for f16 on the reduced-width models,
1.0e-3 at full width; 57 % of its square is the rounding of the GroupNorm / LayerNorm outputs, i.e. of the convs'
and GEMMs' A operands, 26 % that of the 16-bit GEMM outputs q/k/v); the test asserts that the kernels sit on that
floor: its error against the reference is no more than 1.2 x the floor's (hip and ideal16 are two independent
realisations of the same rounding process - a last-bit difference in an f32 sum flips roundings downstream - so they sit
~sqrt(2) floors apart from each other).

The second test separates the per-forward error from its amplification by classifier-free guidance: along the
REFERENCE trajectory (oracle DDIM loop, f32) every step's x_t is fed to the HIP U-Net, so each forward is measured
on identical inputs; v = e_u + s (e_c - e_u) then carries (s |d e_c| + (s-1) |d e_u|) / |v| - an amplification
that is computed from the reference's own tensors, not fitted."""
import pytest
import torch

from oracle import ddim_ref, golden_recipe as gr, unet_ref
from oracle.ops_torch import TorchOps
from open_pandora_amd import synth
from open_pandora_amd.ddim import DDIMSampler
from open_pandora_amd.ddpm import LatentVisualDiffusion
from open_pandora_amd.unet import UNetModel
from test_oracle_golden import RH_KW, load, rel

pytestmark = pytest.mark.gpu

# measured floors (ideal16 vs reference, CPU arithmetic - reproducible anywhere): f16 1.3e-3, bf16 1.0e-2
FLOOR_MAX = {torch.float16: 1.5e-3, torch.bfloat16: 1.2e-2}
OVER_FLOOR = 1.2  # hip error / ideal16 error


def _model(mc, ops):
    m = UNetModel(**dict(RH_KW, model_channels=mc)).eval()
    m.load_state_dict(synth.synth_state_dict(m, seed=gr.WEIGHT_SEED))
    return m.bind(ops)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("tag,mc,h,w,t,fs", gr.UNET_SMALL_CASES)
def test_kernels_sit_on_the_16bit_operand_floor(hip_ops_factory, dtype, tag, mc, h, w, t, fs):
    g = load("unet_small.npz")[tag]
    ins, _, _ = gr.sampler_inputs(h, w)
    x = torch.cat([ins["x_T"], ins["c_concat"]], 1)
    args = lambda d: (x.to(d), torch.tensor([t]).to(d))
    kw = lambda d: dict(context=ins["c_crossattn"].to(d), fs=torch.tensor([fs]).to(d))
    ideal = _model(mc, TorchOps(dtype, model_16bit=True))(*args("cpu"), **kw("cpu")).float()
    hip = _model(mc, hip_ops_factory(dtype))(*args("cuda"), **kw("cuda")).float().cpu()
    e_ideal, e_hip, d = rel(ideal, g), rel(hip, g), rel(hip, ideal)
    print(f"\n[budget] {tag} {dtype}: reference->ideal16 {e_ideal:.2e}  reference->hip {e_hip:.2e}  ideal16->hip {d:.2e}")
    assert e_ideal <= FLOOR_MAX[dtype]
    assert e_hip <= OVER_FLOOR * e_ideal
    assert d <= 2.0 * max(e_ideal, e_hip)  # (two independent realisations of the rounding: ~sqrt 2 apart)


@pytest.mark.parametrize("dtype", [torch.float16])
def test_per_forward_error_vs_cfg_amplification(hip_ops_factory, dtype):
    S, eta, cfg, mc, h, w = 10, 0.0, 4.0, 64, 8, 8
    ops = hip_ops_factory(dtype)
    unet = _model(mc, ops)
    pm = LatentVisualDiffusion(unet)
    sd = {k: v.detach().float().cpu() for k, v in unet.state_dict().items()}
    ins, cond, uc = gr.sampler_inputs(h, w)
    fs = torch.tensor([15])
    log = []  # (x_t, t, which, e_ref) in call order: cond, uncond per step

    def apply(x, t, c, f):
        e = unet_ref.unet_forward(sd, torch.cat([x] + c["c_concat"], 1), t, torch.cat(c["c_crossattn"], 1), f,
                                  model_channels=mc)
        log.append((x.clone(), t.clone(), c, e))
        return e

    want, _ = ddim_ref.ddim_sample(apply, ddim_ref.schedule_tables(), ins["x_T"], cond, uc, S, eta, cfg, fs=fs)
    dev = lambda c: {k: [t.cuda() for t in v] for k, v in c.items()}
    worst_fwd, worst_ratio = 0.0, 0.0
    for i in range(0, len(log), 2):
        (x, t, c0, ec_ref), (_, _, c1, eu_ref) = log[i], log[i + 1]
        ec = pm.apply_model(x.cuda(), t.cuda(), dev(c0), fs=fs.cuda()).float().cpu()
        eu = pm.apply_model(x.cuda(), t.cuda(), dev(c1), fs=fs.cuda()).float().cpu()
        err_c, err_u = rel(ec, ec_ref), rel(eu, eu_ref)
        v_ref = eu_ref + cfg * (ec_ref - eu_ref)
        err_v = rel(eu + cfg * (ec - eu), v_ref)
        # what guidance may make of those two errors (triangle inequality on v = s e_c - (s-1) e_u)
        bound = (cfg * err_c * ec_ref.norm() + (cfg - 1) * err_u * eu_ref.norm()) / v_ref.norm()
        amp = float(bound / max(err_c, err_u))
        print(f"\n[budget] step {i // 2} t={int(t[0])}: forward err cond {err_c:.2e} uncond {err_u:.2e} | guided v {err_v:.2e} "
              f"(<= {float(bound):.2e}: amplification {amp:.1f}x from the reference's own |e_c|, |e_u|, |v|)")
        assert err_v <= float(bound) * (1 + 1e-3)
        worst_fwd = max(worst_fwd, err_c, err_u)
        worst_ratio = max(worst_ratio, err_v / max(err_c, err_u))
    assert worst_fwd <= 1.2 * FLOOR_MAX[dtype]  # every forward of the trajectory, not only t = 500
    # the whole trajectory (product sampler + fused update kernel) against the same reference run
    y, _ = DDIMSampler(pm).sample(S=S, batch_size=1, shape=(4, 16, h, w), conditioning=dev(cond), verbose=False,
                                  unconditional_guidance_scale=cfg, unconditional_conditioning=dev(uc), eta=eta,
                                  fs=fs.cuda(), timestep_spacing="uniform_trailing", x_T=ins["x_T"].cuda())
    err = rel(y.cpu(), want)
    print(f"\n[budget] {S}-step cfg {cfg} trajectory: {err:.2e}; worst forward {worst_fwd:.2e}, worst guided/forward ratio "
          f"{worst_ratio:.1f}x")
    assert err <= worst_fwd * worst_ratio * 1.5  # errors of successive steps partly cancel; never beyond the per-step product
