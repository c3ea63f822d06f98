"""GPU: the boundary rows on the HIP kernels - checkpoint wire formats into a HipOps-bound model, weight reload
through a parent shell (incl. the sampler's captured HIP graph), the shared-context branch of UNetModel.forward,
and the xformers-contract attention seam (attention.py:146-209)."""
import pytest
import torch

from oracle import golden_recipe as gr
from open_pandora_amd import checkpoint, synth
from open_pandora_amd.ddim import DDIMSampler
from open_pandora_amd.ddpm import LatentVisualDiffusion
from open_pandora_amd.ops_hip import memory_efficient_attention
from open_pandora_amd.unet import UNetModel
from test_oracle_golden import RH_KW, load, rel
from test_unet_gpu import FWD_TOL_REDUCED, TRAJ_TOL_REDUCED

pytestmark = pytest.mark.gpu
TAG, MC, H, W, T_STEP, FS = gr.UNET_SMALL_CASES[0]


def _inputs():
    ins, _, _ = gr.sampler_inputs(H, W)
    return torch.cat([ins["x_T"], ins["c_concat"]], 1).cuda(), ins["c_crossattn"].cuda()


def _wire_formats(usd):
    lightning = {"state_dict": {**{"model.diffusion_model." + k: v for k, v in usd.items()},
                                "cond_stage_model.dummy": torch.zeros(1), "scale_arr": torch.zeros(3)}}
    return {
        "lightning": lightning,                                                   # inference.py:27-45
        "deepspeed": {"module": {"_forward_module." + k: v for k, v in lightning["state_dict"].items()}},  # :46-50
        "pytorch_model.bin": {**{"diffusion_model.model.diffusion_model." + k: v for k, v in usd.items()},
                              "video_model.lm_head.weight": torch.zeros(2, 2)},   # model.py:599, tools/ckpt2bin.py
        "framestride_embed": {"state_dict": {"model.diffusion_model." + k.replace("fps_embedding", "framestride_embed"): v
                                             for k, v in usd.items()}},            # inference.py:36-43
    }


@pytest.mark.parametrize("dtype", [torch.float16])
@pytest.mark.parametrize("fmt", ["lightning", "deepspeed", "pytorch_model.bin", "framestride_embed"])
def test_checkpoint_wire_formats_into_a_hip_bound_unet(hip_ops_factory, dtype, fmt):
    """SURVEY 8f row 3 on the GPU: every wire format, loaded with checkpoint.load_unet into a model that is
    ALREADY bound to HipOps and has run (stale packed weights), reproduces the reference forward."""
    g = load("unet_small.npz")[TAG]
    m = UNetModel(**dict(RH_KW, model_channels=MC)).eval().bind(hip_ops_factory(dtype))
    m.load_state_dict(synth.synth_state_dict(m, seed=999))  # other weights first
    x, ctx = _inputs()
    t, fs = torch.tensor([T_STEP]).cuda(), torch.tensor([FS]).cuda()
    stale = m(x, t, context=ctx, fs=fs)
    blob = _wire_formats(synth.synth_state_dict(m, seed=gr.WEIGHT_SEED))[fmt]
    res = checkpoint.load_unet(m, blob)
    assert not res.missing_keys and not res.unexpected_keys
    y = m(x, t, context=ctx, fs=fs)
    assert rel(y.cpu(), g) <= FWD_TOL_REDUCED[dtype] and rel(y.cpu(), stale.cpu()) > 0.1


def test_parent_reload_invalidates_packed_weights_and_the_captured_graph(hip_ops_factory):
    dtype = torch.float16
    g = load("ddim_small.npz")["S5_eta0_cfg4"]
    m = UNetModel(**dict(RH_KW, model_channels=64)).eval().bind(hip_ops_factory(dtype))
    pm = LatentVisualDiffusion(m)
    m.load_state_dict(synth.synth_state_dict(m, seed=31))
    ins, cond, uc = gr.sampler_inputs(8, 8)
    dev = lambda c: {k: [t.cuda() for t in v] for k, v in c.items()}
    cond_d, uc_d, xT, fs = dev(cond), dev(uc), ins["x_T"].cuda(), torch.tensor([15]).cuda()
    smp = DDIMSampler(pm, use_graph=True)
    run = lambda: smp.sample(S=5, batch_size=1, shape=(4, 16, 8, 8), conditioning=cond_d, verbose=False,
                             unconditional_guidance_scale=4.0, unconditional_conditioning=uc_d, eta=0.0, fs=fs,
                             timestep_spacing="uniform_trailing", x_T=xT)[0]
    first = run()  # captures the forward graph on the seed-31 weights
    sd = pm.state_dict()
    sd.update({"model.diffusion_model." + k: v.to(sd["model.diffusion_model." + k].device)
               for k, v in synth.synth_state_dict(m, seed=gr.WEIGHT_SEED).items()})
    pm.load_state_dict(sd)  # the reference loader's move: through the PARENT (inference.py:27-52)
    y = run()               # same sampler object, same condition tensors: only the pack epoch differs
    assert rel(y.cpu(), g) <= TRAJ_TOL_REDUCED[dtype] and rel(y.cpu(), first.cpu()) > 0.1


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("L,tag", gr.UNET_CTX_CASES)
def test_context_without_per_frame_image_tokens_gpu(hip_ops_factory, dtype, L, tag):
    g = load("unet_small_ctx.npz")[tag]
    m = UNetModel(**dict(RH_KW, model_channels=64)).eval().bind(hip_ops_factory(dtype))
    m.load_state_dict(synth.synth_state_dict(m, seed=gr.WEIGHT_SEED))
    x, ctx = _inputs()
    y = m(x, torch.tensor([500]).cuda(), context=ctx[:, :L], fs=torch.tensor([15]).cuda())
    assert rel(y.cpu(), g) <= FWD_TOL_REDUCED[dtype]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_xformers_contract_attention_seam(hip_ops_factory, dtype):
    """memory_efficient_attention(q, k, v) on contiguous (b*heads, n, 64) tensors, as efficient_forward calls it
    (attention.py:166-175), against the eager formula of attention.py:103-125."""
    ops = hip_ops_factory(dtype)
    g = torch.Generator().manual_seed(5)
    b_h, nq, nk = 16 * 5, 333, 410  # self- and cross-shaped calls share the seam
    q = (torch.randn(b_h, nq, 64, generator=g) * 1.2).to(dtype)
    k = (torch.randn(b_h, nk, 64, generator=g) * 1.2).to(dtype)
    v = torch.randn(b_h, nk, 64, generator=g).to(dtype)
    sim = torch.einsum("bid,bjd->bij", q.float(), k.float()) * 64 ** -0.5
    want = torch.einsum("bij,bjd->bid", sim.softmax(-1), v.float())
    got = memory_efficient_attention(q.cuda(), k.cuda(), v.cuda(), attn_bias=None, op=None, ops=ops)
    assert got.shape == (b_h, nq, 64) and got.is_contiguous()
    assert rel(got.cpu(), want) <= (1e-3 if dtype == torch.float16 else 4e-3)
    with pytest.raises(NotImplementedError):
        memory_efficient_attention(q.cuda(), k.cuda(), v.cuda(), attn_bias=torch.zeros(1), ops=ops)
