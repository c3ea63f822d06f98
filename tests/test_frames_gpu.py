"""GPU: end-to-end FRAMES - the north-star's tolerance is on frames, not latents (VERDICT r02 missing #6, weak #2, next #4b/d,
#7).  Product sampler (DDIMSampler on HipOps) -> product first stage (AutoencoderKL.decode_first_stage on HipOps)
against fixtures of the REAL reference's chain (DDIMSampler.sample -> LatentDiffusion.decode_first_stage, ddim.py:66 /
ddpm3d.py:630-655; oracle/make_golden.py --frames / --frames-full):

* reduced width, stored in full: 5 steps eta 0 (parity dtype f16) and the PRODUCTION schedule - 50 steps, eta 1, shared
  noise - in f16 and bf16;
* full width (1.44 B U-Net + 83.7 M AutoencoderKL) at 16x40x64 -> 320x512: BASELINE config 1 (10 steps, eta 0) in f16 and
  the 50-step eta-1 production loop in bf16 (fixtures are digests of the reference's latent and frames: hours of CPU each);
* the multi-round driver (wm.DiffusionRunner.generate_multiround = ChatWM.generate_video_mutliround, model.py:1094-1129)
  on HipOps + the HIP AutoencoderKL, 2 rounds incl. the 8-bit round trip (model.py:1179-1187), fp8 attention off and on,
  against the same driver on the CPU oracle's op table and first stage.

Tolerances are stated per case next to the measured values the tests print ([parity] lines, repeated in the summary)."""
import os

import numpy as np
import pytest
import torch

from oracle import golden_recipe as gr
from open_pandora_amd import synth, wm
from open_pandora_amd.autoencoder import DDCONFIG, AutoencoderKL
from open_pandora_amd.ddim import DDIMSampler
from open_pandora_amd.ddpm import LatentVisualDiffusion
from open_pandora_amd.unet import UNetModel
from test_oracle_golden import GOLD, RH_KW, load, rel

pytestmark = pytest.mark.gpu

# reduced width (64 base channels, AE ch 32): trajectory tolerance of tests/test_unet_gpu.py (16-bit-operand floor x CFG
# amplification) carried through the decoder; the 50-step eta-1 loop accumulates 50 CFG steps
FRAMES_SMALL_TOL = {(5, 0.0, torch.float16): 6e-3, (50, 1.0, torch.float16): 8e-3, (50, 1.0, torch.bfloat16): 6e-2}  # measured 3.4e-3 / 3.1e-3 / 3.0e-2
# full width: f16 = the parity configuration (forward contract 1e-3, x the CFG amplification bound 3.5 of a trajectory,
# then the decoder's own 1e-3); bf16 = the dtype of every perf number, 50 steps of the production loop
FRAMES_FULL_TOL = {("s10_eta0", torch.float16): 3.5e-3, ("s50_eta1", torch.bfloat16): 3e-2}  # measured 1.34e-3 / 8.95e-3


def _small(ops):
    m = UNetModel(**dict(RH_KW, model_channels=64)).eval()
    m.load_state_dict(synth.synth_state_dict(m, seed=gr.WEIGHT_SEED))
    ae = AutoencoderKL(ddconfig=dict(DDCONFIG, ch=32))
    ae.load_state_dict(synth.synth_state_dict(ae, seed=gr.WEIGHT_SEED))
    return LatentVisualDiffusion(m.bind(ops)), ae.bind(ops)


def _sample(pm, h, w, S, eta):
    ins, cond, uc = gr.sampler_inputs(h, w)
    dev = lambda c: {k: [t.cuda() for t in v] for k, v in c.items()}
    ns = gr.noises(ins["x_T"].shape, S) if eta > 0 else None
    z, _ = DDIMSampler(pm).sample(S=S, batch_size=1, shape=(4, 16, h, w), conditioning=dev(cond), verbose=False,
                                  unconditional_guidance_scale=4.0, unconditional_conditioning=dev(uc), eta=eta,
                                  fs=torch.tensor([15]).cuda(), timestep_spacing="uniform_trailing", x_T=ins["x_T"].cuda(),
                                  noise_fn=(lambda i, shape: ns[i]) if ns else None)
    return z


@pytest.mark.parametrize("S,eta,dtype", [(5, 0.0, torch.float16), (50, 1.0, torch.float16), (50, 1.0, torch.bfloat16)])
def test_frames_reduced_width(hip_ops_factory, S, eta, dtype):
    g = load("frames_small.npz")
    pm, ae = _small(hip_ops_factory(dtype))
    z = _sample(pm, 8, 8, S, eta)
    frames = ae.decode_first_stage(z)
    e_z, e_f = rel(z.cpu(), g[f"S{S}_eta{eta:g}/latent"]), rel(frames.cpu(), g[f"S{S}_eta{eta:g}/frames"])
    print(f"\n[parity] frames reduced S={S} eta={eta:g} {dtype}: latent {e_z:.2e} -> frames {e_f:.2e}")
    assert frames.shape == (1, 3, 16, 64, 64) and e_f <= FRAMES_SMALL_TOL[(S, eta, dtype)]


def _digest(y, g, key, tol):
    """gr.compare_digest: relative error against the WHOLE reference latent / a prime-stride sample of the reference frames that
    visits every column, row and frame - after asserting std, mean, absmax and the per-column / per-row rms and mean profiles
    of the fixture to `tol` (the old strides aliased with the width)."""
    return gr.compare_digest(y, g, key, tol)


@pytest.mark.parametrize("tag,S,eta,dtype", [("s10_eta0", 10, 0.0, torch.float16), ("s50_eta1", 50, 1.0, torch.bfloat16)])
def test_frames_full_width_320x512(hip_ops_factory, tag, S, eta, dtype):
    path = os.path.join(GOLD, f"frames_full_40x64_s{S}_eta{eta:g}.npz")
    if not os.path.exists(path):
        pytest.skip(f"{os.path.basename(path)} not generated yet (oracle/make_golden.py --frames-full, hours of CPU)")
    from open_pandora_amd import factory
    g = np.load(path)
    ops = hip_ops_factory(dtype)
    pm = factory.build_diffusion("320x512", ops, seed=gr.WEIGHT_SEED)
    z = _sample(pm, 40, 64, S, eta)
    ae = AutoencoderKL()
    ae.load_state_dict(synth.synth_state_dict(ae, seed=gr.WEIGHT_SEED))
    frames = ae.bind(ops).decode_first_stage(z)
    tol = FRAMES_FULL_TOL[(tag, dtype)]
    e_z, _, _ = _digest(z, g, "latent", tol)
    e_f, std, gstd = _digest(frames, g, "frames", tol)
    print(f"\n[parity] frames full 320x512 S={S} eta={eta:g} {dtype}: latent {e_z:.2e} -> frames {e_f:.2e} "
          f"(std {std:.4f} vs {gstd:.4f})")
    assert frames.shape == (1, 3, 16, 320, 512) and e_f <= FRAMES_FULL_TOL[(tag, dtype)]
    del pm, ae
    torch.cuda.empty_cache()


# The north-star's number itself, on FRAMES, no amplification factor: the parity configuration HipOps(parity=True) (norm outputs
# carried as [hi | lo] 16-bit parts) in f16.  bench.py --parity --dtype f16 puts this configuration's step
# time on record next to the bf16 production number.
FRAMES_PARITY_TOL = 1e-3
# A 2-step trajectory is NOT the north-star's configuration: its two forwards sit at t = 999 (zero terminal SNR: the model output
# IS the sample) and t = 499 and enter the result with the full guidance amplification, nothing averages out (measured 1.5e-3 on
# the latent, 1.7e-3 on frames, against 8.1e-4 / 9.4e-4 for 10 steps at 320x512).  It is kept as the cheap 576x1024 fixture
# (937 s of reference CPU) with its own, stated bound; the 10-step 576x1024 fixture (~80 min of reference CPU) carries the 1e-3.
FRAMES_PARITY_TOL_2STEP = 2.0e-3


@pytest.mark.parametrize("res,h,w,S,tol,mode", [("320x512", 40, 64, 10, FRAMES_PARITY_TOL, True),
                                                ("576x1024", 72, 128, 10, FRAMES_PARITY_TOL, True),
                                                ("576x1024", 72, 128, 2, FRAMES_PARITY_TOL_2STEP, True),
                                                ("320x512", 40, 64, 10, FRAMES_PARITY_TOL, "selective"),
                                                ("576x1024", 72, 128, 10, FRAMES_PARITY_TOL, "selective")])
def test_frames_full_width_parity_mode(res, h, w, S, tol, mode):
    """Full-width sampler -> first-stage decode in the parity configuration against the REAL reference's frames
    (DDIMSampler.sample -> decode_first_stage, ddim.py:66 / ddpm3d.py:630-655): BASELINE config 1 (320x512, 10 CFG-4 steps,
    eta 0) and 576x1024 (configs[2]'s latent; 10 and 2 CFG-4 steps: one reference step is ~7 min of CPU at that size).
    mode "selective" (r05): [hi | lo] operands only at the sites that buy error (ops_hip.SELECTIVE_PARITY_SITES,
    from the leave-one-out table of tools/parity_sites.py) - the same 1e-3, no factor, at 1.22-1.23 x the default f16 step
    instead of 1.43-1.45 x (bench.py's `parity_mode` object carries the step time and this error in the driver's line)."""
    path = os.path.join(GOLD, f"frames_full_{h}x{w}_s{S}_eta0.npz")
    if not os.path.exists(path):
        pytest.skip(f"{os.path.basename(path)} not generated yet (oracle/make_golden.py --frames-full[-72x128])")
    from open_pandora_amd import factory
    from open_pandora_amd.ops_hip import HipOps
    g = np.load(path)
    ops = HipOps(torch.float16, "cuda:0", parity=mode)
    pm = factory.build_diffusion(res, ops, seed=gr.WEIGHT_SEED)
    z = _sample(pm, h, w, S, 0.0)
    del pm
    torch.cuda.empty_cache()
    ae = AutoencoderKL()
    ae.load_state_dict(synth.synth_state_dict(ae, seed=gr.WEIGHT_SEED))
    frames = ae.bind(ops).decode_first_stage(z)  # (outside the U-Net every norm output is [hi | lo] in either mode)
    e_z, _, _ = _digest(z, g, "latent", tol)
    e_f, std, gstd = _digest(frames, g, "frames", tol)
    print(f"\n[parity] PARITY MODE ({'all sites' if mode is True else mode}) frames full {res} S={S} eta=0 f16: latent {e_z:.2e} "
          f"-> frames {e_f:.2e} (std {std:.4f} vs {gstd:.4f})")
    assert frames.shape == (1, 3, 16, 8 * h, 8 * w) and e_f <= tol
    del ae
    torch.cuda.empty_cache()


# VERDICT r05 #4 / ADVICE r05 (medium): ops_hip.SELECTIVE_PARITY_SITES was chosen on the fixture it is asserted against (weights
# 20230211, inputs 123).  The HELD-OUT fixture - other weights (777001), other inputs (456), generated by the REAL reference with
# `oracle/make_golden.py --frames-full 10:0 --heldout-seeds 777001:456` (18 min of CPU) - says whether the contract number holds
# without tuning; the site list is NOT re-fitted on it.  MEASURED (r06, profiles/r06/parity_heldout.txt): it does not, for either
# configuration - the LATENT stays inside 1e-3 (8.1e-4 all sites / 8.6e-4 selective; 8.0e-4 / 8.5e-4 on the tuned pair), but the
# first-stage decoder amplifies a latent error by 1.17 on the tuned pair and by 1.28 on this one, so the FRAMES land at 1.04e-3 /
# 1.10e-3 (9.4e-4 / 9.9e-4 tuned).  Splitting the decoder's WEIGHTS hi / lo as well changed nothing (9.37e-4 vs 9.38e-4: its own
# error is not what is left).  "Frames within 1e-3" is therefore a statement about ONE seed pair with a 1-6 % margin, not a
# property of the f16 configurations; the test asserts what holds on both pairs - the latent inside 1e-3 - and bounds the frames
# at 1.1 x the measured value so that a regression fails it.  bench.py's `parity_mode` carries both pairs' margins.
FRAMES_HELDOUT_TOL = {True: 1.15e-3, "selective": 1.21e-3}  # measured 1.04e-3 / 1.10e-3
HELDOUT_W, HELDOUT_I = 777001, 456


@pytest.mark.parametrize("mode", [True, "selective"])
def test_frames_full_width_parity_mode_heldout_seeds(mode, monkeypatch):
    path = os.path.join(GOLD, f"frames_full_40x64_s10_eta0_w{HELDOUT_W}_i{HELDOUT_I}.npz")
    if not os.path.exists(path):
        pytest.skip("held-out fixture not generated (oracle/make_golden.py --frames-full 10:0 --heldout-seeds 777001:456)")
    from open_pandora_amd import factory
    from open_pandora_amd.ops_hip import HipOps
    g = np.load(path)
    assert int(g["weight_seed"]) == HELDOUT_W and int(g["input_seed"]) == HELDOUT_I
    monkeypatch.setattr(gr, "INPUT_SEED", HELDOUT_I)  # (gr.sampler_inputs reads it at call time)
    ops = HipOps(torch.float16, "cuda:0", parity=mode)
    pm = factory.build_diffusion("320x512", ops, seed=HELDOUT_W)
    z = _sample(pm, 40, 64, 10, 0.0)
    del pm
    torch.cuda.empty_cache()
    ae = AutoencoderKL()
    ae.load_state_dict(synth.synth_state_dict(ae, seed=HELDOUT_W))
    frames = ae.bind(ops).decode_first_stage(z)
    e_z, _, _ = _digest(z, g, "latent", FRAMES_PARITY_TOL)
    e_f, std, gstd = _digest(frames, g, "frames", FRAMES_HELDOUT_TOL[mode])
    print(f"\n[parity] HELD-OUT seeds (w {HELDOUT_W}, i {HELDOUT_I}) parity mode ({'all sites' if mode is True else mode}) frames 320x512 "
          f"S=10 eta=0 f16: latent {e_z:.2e} -> frames {e_f:.2e} (margin to 1e-3: {100 * (1 - e_f / FRAMES_PARITY_TOL):.1f} %; "
          f"std {std:.4f} vs {gstd:.4f})")
    assert frames.shape == (1, 3, 16, 320, 512) and e_z <= FRAMES_PARITY_TOL and e_f <= FRAMES_HELDOUT_TOL[mode]
    del ae
    torch.cuda.empty_cache()


# BASELINE configs[4] ("fp8 MFMA attention") at its own resolution and width, SELECTIVELY (e4m3 on every
# attention of a model is not a usable design): HipOps(fp8_attention=True) with the default fp8_min_tokens = 2048 puts the
# spatial self-attention of levels 0 and 1 (9216 / 2304 tokens at 576x1024: 94 % of the attention FLOPs) on the block-scaled
# MFMA and leaves the deep levels, the cross-attentions and the temporal attentions on 16-bit operands.  Frames against the
# REAL reference's 2-step fixture.  The bound is 1.3 x the measured value (printed): a regression of the selection rule or
# of the fp8 kernel fails it - it is not a loose "cannot fail" number.
# r06 (VERDICT r05 #6a): the same on the committed 10-STEP fixture - configs[4]'s per-round loop length class, where the CFG steps'
# errors average instead of entering the result at full guidance amplification as the two steps of the short fixture do.
FRAMES_FP8_576_TOL = {2: 1.1e-2, 10: 5.1e-3}  # = 1.3 x measured. 2 steps: 8.1e-3 (latent 7.4e-3; f16 without fp8 2.3e-3, parity mode 1.7e-3 on the same fixture); 10 steps: 3.9e-3 (latent 3.7e-3), profiles/r06/fp8_frames_10step.txt


@pytest.mark.parametrize("S", [2, 10])
def test_frames_full_width_576x1024_fp8_attention_selective(monkeypatch, S):
    path = os.path.join(GOLD, f"frames_full_72x128_s{S}_eta0.npz")
    if not os.path.exists(path):
        pytest.skip(f"frames_full_72x128_s{S}_eta0.npz not generated yet (oracle/make_golden.py --frames-full-72x128 {S}:0)")
    from open_pandora_amd import factory
    from open_pandora_amd.ops_hip import HipOps
    g = np.load(path)
    ops = HipOps(torch.float16, "cuda:0", fp8_attention=True)
    calls = {"fp8": 0, "f16": 0}
    fp8_inner, att_inner = ops.attention_fp8, ops.attention
    ops.attention_fp8 = lambda *a, **k: (calls.__setitem__("fp8", calls["fp8"] + 1), fp8_inner(*a, **k))[1]
    ops.attention = lambda *a, **k: (calls.__setitem__("f16", calls["f16"] + 1), att_inner(*a, **k))[1]
    pm = factory.build_diffusion("576x1024", ops, seed=gr.WEIGHT_SEED)
    monkeypatch.setenv("PANDORA_HIPGRAPH", "0")  # (eager: the call counters above see every forward; restored by pytest -
    z = _sample(pm, 72, 128, S, 0.0)              #  an externally set value survives this test)
    # per forward: 10 spatial self-attentions on fp8 (levels 0 and 1: 5 + 5), 6 (levels 2, 3 + middle) and the 16 cross-attentions on f16
    if os.environ.get("PANDORA_CFG_BATCH", "0") == "1":  # one forward over both clips per step, cross-attention per clip
        assert calls["fp8"] == 10 * S and calls["f16"] == (6 + 16 * 2) * S, calls
    else:
        assert calls["fp8"] == 10 * 2 * S and calls["f16"] == (6 + 16) * 2 * S, calls
    del pm
    torch.cuda.empty_cache()
    ae = AutoencoderKL()
    ae.load_state_dict(synth.synth_state_dict(ae, seed=gr.WEIGHT_SEED))
    frames = ae.bind(HipOps(torch.float16, "cuda:0")).decode_first_stage(z)
    e_z, _, _ = _digest(z, g, "latent", FRAMES_FP8_576_TOL[S])
    e_f, std, gstd = _digest(frames, g, "frames", FRAMES_FP8_576_TOL[S])
    print(f"\n[parity] frames full 576x1024 S={S} f16 + SELECTIVE fp8 attention (levels 0-1): latent {e_z:.2e} -> frames {e_f:.2e} "
          f"(std {std:.4f} vs {gstd:.4f})")
    assert frames.shape == (1, 3, 16, 576, 1024) and e_f <= FRAMES_FP8_576_TOL[S]
    del ae
    torch.cuda.empty_cache()


def _oracle_runner():
    """The same driver on the CPU oracle: TorchOps op table + oracle/ae_ref first stage (tests only)."""
    from oracle import ae_ref
    from oracle.ops_torch import TorchOps
    m = UNetModel(**dict(RH_KW, model_channels=64)).eval()
    m.load_state_dict(synth.synth_state_dict(m, seed=gr.WEIGHT_SEED))
    ae = AutoencoderKL(ddconfig=dict(DDCONFIG, ch=32))
    sd = synth.synth_state_dict(ae, seed=gr.WEIGHT_SEED)
    enc = lambda x: ae_ref.ae_sample_latent(ae_ref.ae_encode_moments(sd, x), torch.zeros(x.shape[0], 4, x.shape[2] // 8, x.shape[3] // 8))
    dec = lambda z: ae_ref.ae_decode(sd, z)
    return LatentVisualDiffusion(m.bind(TorchOps())), enc, dec


@pytest.mark.parametrize("fp8", [False, True])
def test_multiround_driver_on_the_gpu(fp8):
    """BASELINE configs[4] at reduced width: 2 autoregressive rounds; round 2 is conditioned on round 1's last 4 frames
    through the 8-bit PIL round trip; fp8 (e4m3) spatial attention off and on."""
    from open_pandora_amd.ops_hip import HipOps
    torch.set_num_threads(8)
    ops = HipOps(torch.float16, "cuda:0", fp8_attention=fp8, fp8_min_tokens=0)
    pm, ae = _small(ops)
    ins, _, _ = gr.sampler_inputs(8, 8)
    text, img = ins["c_crossattn"][:, :77], ins["c_crossattn"][:, 77:]
    uct, uci = ins["uc_crossattn"][:, :77], ins["uc_crossattn"][:, 77:]
    frame0 = gr.ae_pixels(1, 64, 64).permute(1, 0, 2, 3)  # (3, 1, H, W)
    kw = dict(n_samples=1, ddim_steps=3, ddim_eta=0.0, unconditional_guidance_scale=4.0, fs=15,
              timestep_spacing="uniform_trailing")
    zero_noise = lambda x, noise=None: ae.encode_first_stage(x, noise=torch.zeros(x.shape[0], 4, x.shape[2] // 8, x.shape[3] // 8))
    gpu = wm.DiffusionRunner(pm, lambda im: (img if float(im.abs().sum()) > 0 else uci).cuda(), uct.cuda(), zero_noise,
                             ae.decode_first_stage)
    got = gpu.generate_multiround([text.cuda(), (text * 0.9).cuda()], frame0.cuda(), frame0[None, :, 0].cuda(),
                                  x_T=ins["x_T"].cuda(), **kw)
    pm_o, enc_o, dec_o = _oracle_runner()
    cpu = wm.DiffusionRunner(pm_o, lambda im: img if float(im.abs().sum()) > 0 else uci, uct, enc_o, dec_o)
    want = cpu.generate_multiround([text, text * 0.9], frame0, frame0[None, :, 0], x_T=ins["x_T"], **kw)
    assert got.shape == want.shape == (1, 1, 3, 12 + 16, 64, 64)
    e1 = rel(got[0, 0][:, :12].cpu(), want[0, 0][:, :12])   # round 1 (frames 0-11 survive the stitching)
    e2 = rel(got[0, 0][:, 12:].cpu(), want[0, 0][:, 12:])   # round 2: through the 8-bit conditioning frames
    print(f"\n[parity] multiround reduced f16 fp8_attention={fp8}: round 1 frames {e1:.2e}, round 2 frames {e2:.2e}")
    # f16: trajectory tolerance carried through the decoder, x ~3 for round 2 (conditioned on 8-bit frames of round 1: one
    # uint8 flip is 8e-3 of a pixel).  fp8 attention: 5.5e-2 per attention CALL (three e4m3 roundings) on
    # EVERY spatial self-attention of the reduced model (fp8_min_tokens=0): measured 7.6e-2 / 1.6e-1 in the frames
    tol = 1.0e-1 if fp8 else 1.2e-2
    assert e1 <= tol and e2 <= 2 * tol


# bf16 = the dtype every perf number is quoted in (the reference hard-codes it, openaimodel3d.py:364) at BASELINE configs[2]'s
# resolution, DEFAULT mode, against the REAL reference's 10-step 576x1024 frames (this comparison lived only in
# bench.py's dtype).  Tolerance = 1.3 x measured (8 x coarser operands than f16: not the 1e-3 contract).
FRAMES_576_BF16_TOL = 1.3 * 1.06e-2  # measured: latent 9.0e-3 -> frames 1.06e-2


def test_frames_full_width_576x1024_bf16_default_mode(hip_ops_factory):
    path = os.path.join(GOLD, "frames_full_72x128_s10_eta0.npz")
    if not os.path.exists(path):
        pytest.skip("frames_full_72x128_s10_eta0.npz not generated yet (oracle/make_golden.py --frames-full-72x128 10:0)")
    from open_pandora_amd import factory
    g = np.load(path)
    ops = hip_ops_factory(torch.bfloat16)
    pm = factory.build_diffusion("576x1024", ops, seed=gr.WEIGHT_SEED)
    z = _sample(pm, 72, 128, 10, 0.0)
    ae = AutoencoderKL()
    ae.load_state_dict(synth.synth_state_dict(ae, seed=gr.WEIGHT_SEED))
    frames = ae.bind(ops).decode_first_stage(z)
    e_z, _, _ = _digest(z, g, "latent", FRAMES_576_BF16_TOL)
    e_f, std, gstd = _digest(frames, g, "frames", FRAMES_576_BF16_TOL)
    print(f"\n[parity] frames full 576x1024 S=10 bf16 DEFAULT mode: latent {e_z:.2e} -> frames {e_f:.2e} (std {std:.4f} vs {gstd:.4f})")
    assert frames.shape == (1, 3, 16, 576, 1024) and e_f <= FRAMES_576_BF16_TOL
    del pm, ae
    torch.cuda.empty_cache()


@pytest.mark.slow
@pytest.mark.parametrize("fp8", [False, True], ids=["bf16_attention", "fp8_attention"])
def test_multiround_driver_5_rounds_576x1024_full_width(fp8):
    """BASELINE configs[4]'s driver at FULL size under pytest - with the spatial self-attention of levels 0-1 on the fp8 (e4m3)
    MFMA kernel too, which is configs[4]'s named configuration (it ran only in bench.py --multiround): 5
    autoregressive rounds at 576x1024 on the 1.44 B U-Net + the full AutoencoderKL in bf16, 4 DDIM steps per round (the loop
    length is not what this exercises).  Shape (1, 1, 3, 12 x 4 + 16, 576, 1024), finite (the decoder of the seeded weights overshoots [-1, 1]: |x| < 10); round 1's frames are
    the single-round `generate` of the same inputs bit for bit; every later round was conditioned on the previous round's last
    4 frames through the 8-bit round trip (model.py:1179-1187) - the encoder saw 1, 4, 4, 4, 4 frames."""
    from open_pandora_amd import factory
    from open_pandora_amd.ops_hip import HipOps
    ops = HipOps(torch.bfloat16, "cuda:0", fp8_attention=fp8)
    fp8_calls = [0]
    if fp8:
        inner = ops.attention_fp8
        ops.attention_fp8 = lambda *a, **k: (fp8_calls.__setitem__(0, fp8_calls[0] + 1), inner(*a, **k))[1]
    pm = factory.build_diffusion("576x1024", ops, seed=gr.WEIGHT_SEED)
    ae = AutoencoderKL()
    ae.load_state_dict(synth.synth_state_dict(ae, seed=gr.WEIGHT_SEED))
    ae.bind(ops)
    ins, _, _ = gr.sampler_inputs(72, 128)
    text, img = ins["c_crossattn"][:, :77].cuda(), ins["c_crossattn"][:, 77:].cuda()
    uct, uci = ins["uc_crossattn"][:, :77].cuda(), ins["uc_crossattn"][:, 77:].cuda()
    frame0 = gr.ae_pixels(1, 576, 1024).permute(1, 0, 2, 3).cuda()  # (3, 1, H, W)
    seen = []

    def enc(x, noise=None):
        seen.append(x.shape[0])
        return ae.encode_first_stage(x, noise=torch.zeros(x.shape[0], 4, x.shape[2] // 8, x.shape[3] // 8, device=x.device))

    kw = dict(n_samples=1, ddim_steps=4, ddim_eta=0.0, unconditional_guidance_scale=4.0, fs=15, timestep_spacing="uniform_trailing",
              x_T=ins["x_T"].cuda())
    run = wm.DiffusionRunner(pm, lambda im: img if float(im.abs().sum()) > 0 else uci, uct, enc, ae.decode_first_stage)
    texts = [text * s for s in (1.0, 0.9, 1.1, 0.95, 1.05)]
    video = run.generate_multiround(texts, frame0, frame0[None, :, 0], **kw)
    assert video.shape == (1, 1, 3, 12 * 4 + 16, 576, 1024) and seen == [1, 4, 4, 4, 4]
    assert bool(torch.isfinite(video).all()) and float(video.abs().max()) < 10.0
    single = run.generate(texts[0], frame0, frame0[None, :, 0], **kw)
    assert torch.equal(video[0, 0][:, :12], single[0, 0][:, :12])
    stds = [float(video[0, 0][:, 12 * r:12 * r + 12].float().std()) for r in range(5)]
    assert (fp8_calls[0] > 0) == fp8  # (warm-up / capture forwards walk the op table: levels 0 and 1 take the fp8 kernel)
    print(f"\n[parity] multiround FULL width 5 rounds x 576x1024 bf16{' + fp8 attention (levels 0-1)' if fp8 else ''}: 64 frames, "
          f"per-round frame std {['%.3f' % s for s in stds]}")
    assert all(0.05 < s < 1.5 for s in stds)
    run.sampler.close()
    del pm, ae
    torch.cuda.empty_cache()
