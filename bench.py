"""bench.py - denoising steps/sec of the DDIM / 3-D U-Net hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--dtype bf16|f16] [--only 320x512|576x1024]

One "step" = one DDIM step of a 16-frame clip with classifier-free guidance = 2 U-Net forwards
(25.2 TFLOP at 320x512, 104.7 TFLOP at 576x1024) + the fused update kernel.  Inputs (latent, contexts,
weights) are synthetic (seeded) and resident in HBM before the timed region.  ONE run measures BOTH
BASELINE resolutions on the same 1.44 B-parameter U-Net: the headline line (`value`, `ms_per_step`) is
BASELINE configs[1] (320x512), `res_576x1024` carries configs[2].  Prints ONE JSON line on rank 0; extra
objects:
  roofline            the kernel family with the largest share of the 320x512 step (chosen from the live
                      measurement, not hard-coded), timed launch by launch with HIP events on the launch stream
                      during eagerly issued steps of the same loop (the timed steps replay a HIP graph, whose
                      nodes events cannot bracket); `families` lists every family's time / TFLOP/s;
  roofline_attention  the spatial self-attention at N = 9216 tokens (72x128 latent, 16 frames x 5 heads),
                      FLOPs 4 N^2 64 heads frames per launch (attention.py:81-144), in the 576x1024 loop;
  cpu_baseline        the CPU oracle (f32 eager restatement of the reference) on the host cores.
  roofline.dominant_kernel   the single KERNEL (not family) with the largest summed time, with its own fraction;
  compute_scaling     one rank's share of a frame-sharded step measured on THIS GPU: the U-Net forward (graph replay) on
                      clips of T/N frames for N = 2, 4, 8 - the compute-side bound of the N-GPU speed-up.
`share_of_sequential_step` are shares of the EAGER, single-stream step the family times were taken in (they sum to 1 with
`other`); the timed steps replay a two-stream graph and are shorter than that sum.  `roofline.traffic` comes from the
committed PMC summary (profiles/r04/pmc_traffic.json, separate rocprofv3 --pmc passes) only when that file was
produced with the library sources of this run (digest match), else null - never a stale constant.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_STEP = {"320x512": 25.21e12, "576x1024": 104.67e12}  # BASELINE.md §2 (2 forwards)
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16/f16, MI355X_MICROARCH.md
T = 16

KERNELS = {
    "gemm": "pm_gemm + pm_ln_gemm (dense nn.Linear / 1x1: gemm_kernel<A_DENSE> 2-stage, gemm_ring_kernel<A_DENSE>, gemm_ringw_kernel<A_DENSE> 256x128, split-K reduce; ln_gemm_kernel = LayerNorm + projection at K = 320; r06: gemm_wide_stream_kernel = 256x256 tile as one assembly statement on the GEGLU / wide 16-bit projections, gemm_wide_kernel)",
    "conv3x3": "pm_conv2d_3x3 (gemm_ring_kernel<A_CONV3X3_FAST>; gemm_kernel for f32-operand / strided / upsampling convs)",
    "conv_t3": "pm_conv_temporal_k3 (gemm_kernel / gemm_ring_kernel<A_CONVT3>)",
    "attention": "pm_attention (attn_self_kernel: spatial self-attention; attn_kernel: text+image cross-attention)",
}


# the HBM-bound side kernels of a step, bracketed the same way (`other` used to be one bucket)
SIDE = {
    "groupnorm_apply": "pm_groupnorm_apply (gn_apply_kernel: normalise + affine (+ SiLU) of the f32 stream into a 16-bit operand)",
    "groupnorm_stats": "pm_groupnorm_stats + pm_groupnorm_finalize_colstats (statistics passes / finalize of the epilogue column sums)",
    "layernorm": "pm_layernorm (levels without the fused LayerNorm + projection panel kernel)",
    "attention_temporal": "pm_attention_temporal (tattn_kernel: 16-frame attention per pixel and head)",
    "split16": "pm_split16 / pm_split16_upsample2x (f32 stream -> 16-bit operand, hi | lo, written-out Upsample)",
    "small": "pm_gemv_f32 / pm_pack_input / pm_unpack_output / pm_ddim_update (embedding path, layout, the fused update)",
}


class TimedOps:
    """HipOps proxy that brackets every launch of the MFMA kernel families with HIP events on the launch
    stream and counts their algorithmic FLOPs (2 M N K; attention 4 Nq Nk 64 heads B)."""

    def __init__(self, ops):
        self._ops = ops
        self.enabled = False
        self.reset()
        self._install_side()

    def reset(self):
        self.ev = {k: [] for k in KERNELS}
        self.fl = {k: 0.0 for k in KERNELS}
        self.by = {k: 0.0 for k in KERNELS}  # algorithmic HBM bytes: every operand and the output once
        self.rt = {k: 0.0 for k in KERNELS}  # sum over launches of max(FLOPs / MFMA peak, bytes / HBM peak): seconds
        self.attn_big = ([], 0.0)  # (events, flops) of the self-attention launches with Nq == Nk >= 9216
        self.kern = {}  # dense family by KERNEL: name -> [events, flops]
        self.side = {k: [[], 0.0] for k in SIDE}  # side family -> [events, algorithmic bytes]
        self._depth = 0  # > 0 inside a bracketed op (its inner launches belong to the outer bracket)

    def __getattr__(self, k):
        return getattr(self._ops, k)

    def _side(self, fam, fn, *a, **kw):
        if not self.enabled or self._depth:
            return fn(*a, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = fn(*a, **kw)
        e1.record()
        esz = lambda t: t.numel() * t.element_size()
        outs = y if isinstance(y, tuple) else (y,)
        nb = (sum(esz(t) for t in list(a) + list(kw.values()) if torch.is_tensor(t))
              + sum(esz(t) for t in outs if torch.is_tensor(t) and kw.get("out") is None))
        self.side[fam][0].append((e0, e1))
        self.side[fam][1] += nb
        return y

    # (HipOps calls these on ITSELF from inside its composite ops - groupnorm(), conv3x3()'s split16, _stats_end - so the
    # brackets are installed on the wrapped object's methods, once, and see those inner calls too)
    def _install_side(self):
        o = self._ops
        if getattr(o, "_bench_side", False):
            return
        o._bench_side = True
        wrap = lambda fam, name: setattr(o, name, (lambda f: (lambda *a, **kw: self._side(fam, f, *a, **kw)))(getattr(o, name)))
        wrap("groupnorm_apply", "groupnorm_apply")
        wrap("groupnorm_stats", "groupnorm_stats")
        wrap("groupnorm_stats", "_stats_end")
        wrap("layernorm", "layernorm")
        wrap("attention_temporal", "attention_temporal")
        wrap("split16", "split16")
        wrap("split16", "split16_upsample2x")
        for name in ("gemv", "pack_input", "unpack_output", "ddim_update"):
            wrap("small", name)

    def _timed(self, fam, flops, fn, *a, **kw):
        if not self.enabled:
            return fn(*a, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self._depth += 1  # (a split16 / finalize launched from inside this op stays part of its bracket)
        try:
            y = fn(*a, **kw)
        finally:
            self._depth -= 1
        e1.record()
        self.ev[fam].append((e0, e1))
        self.fl[fam] += flops
        esz = lambda t: t.numel() * t.element_size()
        out = y[0] if isinstance(y, tuple) else y
        nbytes = (sum(esz(t) for t in list(a) + list(kw.values()) if torch.is_tensor(t) and t.numel() > 4096)
                  + (esz(out) if torch.is_tensor(out) and kw.get("out") is None else 0.0))
        self.by[fam] += nbytes
        self.rt[fam] += max(flops / (MFMA_PEAK_TFLOPS * 1e12), nbytes / 8e12)
        return y, (e0, e1)

    def _kern(self, name, ev, flops):
        k = self.kern.setdefault(name, [[], 0.0])
        k[0].append(ev)
        k[1] += flops

    def gemm(self, a, w, *args, **kw):
        fl = 2.0 * a.shape[0] * w.shape[0] * w.shape[1]
        r = self._timed("gemm", fl, self._ops.gemm, a, w, *args, **kw)
        if not self.enabled:
            return r
        # which kernel the library's own plan gives this call (pm_gemm_kernel_choice mirrors pm_gemm's decision)
        o = self._ops
        f32_loader = a.dtype == torch.float32 and not o.presplit
        act = kw.get("act", args[2] if len(args) > 2 else "none")
        k_eff = a.shape[1] * (2 if (kw.get("split_a") and a.dtype == torch.float32 and o.presplit) else 1)
        k_eff += (-k_eff) % 64
        ch = o.lib.pm_gemm_kernel_choice(a.shape[0], w.shape[0], k_eff, 2 if act == "geglu" else 0,
                                         1 if f32_loader else 0, o.ws_bytes)
        name = ("gemm_kernel<A_DENSE, f32 operand> (register-staged)" if f32_loader else
                "gemm_wide_stream_kernel (256x256 tile, 128x128 wave tiles, one assembly statement per workgroup)" if ch == 5 else
                "gemm_wide_kernel (256x256 tile, assembly K loop per tile)" if ch == 4 else
                "gemm256_kernel (256x256, 8 waves, ping-pong phases)" if ch == 2 else
                "gemm_ringw_kernel<A_DENSE> (256x128 ring, 4 loader + 4 consumer waves)" if ch == 3 else
                "gemm_ring_kernel<A_DENSE>" if ch == 1 else "gemm_kernel<A_DENSE> (128x128, 2 LDS stages, 2 workgroups/CU)")
        self._kern(name, r[1], fl)
        return r[0]

    def ln_gemm(self, x, gamma, beta, w, *args, **kw):
        # LayerNorm + projection: ONE launch (pm_ln_gemm) at the 320-wide level, counted with its GEMM FLOPs (the
        # normalisation rides in the same kernel); elsewhere the pm_layernorm + pm_gemm pair, whose GEMM is timed
        code = 2 if kw.get("act") == "geglu" else 0
        o = self._ops
        # (r06: where the projection runs on gemm_wide_stream the op table itself takes the pair, HipOps.ln_gemm)
        stream_pair = (o.ln_pair_stream and kw.get("col_scale") is None
                       and o.lib.pm_gemm_kernel_choice(x.shape[0], w.shape[0], x.shape[1], code, 0, o.ws_bytes) == 5)
        if not o.fused_ln or stream_pair or not o.lib.pm_ln_gemm_supported(x.shape[0], w.shape[0], x.shape[1], code):
            return self.gemm(o.layernorm(x, gamma, beta), w, *args, **kw)
        fl = 2.0 * x.shape[0] * w.shape[0] * w.shape[1]
        r = self._timed("gemm", fl, self._ops.ln_gemm, x, gamma, beta, w, *args, **kw)
        if not self.enabled:
            return r
        self._kern("ln_gemm_kernel (LayerNorm + projection panel kernel)", r[1], fl)
        return r[0]

    def conv3x3(self, x, wp, bias, F, H, W, **kw):
        hv, wv = (2 * H, 2 * W) if kw.get("upsample") else (H, W)
        s, pad = kw.get("stride", 1), kw.get("pad_lo", 1)
        m = F * ((hv + pad - 2) // s + 1) * ((wv + pad - 2) // s + 1)
        r = self._timed("conv3x3", 2.0 * m * wp.shape[0] * wp.shape[1], self._ops.conv3x3, x, wp, bias, F, H, W, **kw)
        return r[0] if self.enabled else r

    def conv_t3(self, x, wp, *args, **kw):
        r = self._timed("conv_t3", 2.0 * x.shape[0] * wp.shape[0] * wp.shape[1], self._ops.conv_t3, x, wp, *args, **kw)
        return r[0] if self.enabled else r

    def attention(self, q, k1, v1, heads, k2=None, v2=None, *args, **kw):
        B, nq, _ = q.shape
        nk = k1.shape[1] + (0 if k2 is None else k2.shape[1])
        fl = 4.0 * nq * nk * 64 * heads * B
        r = self._timed("attention", fl, self._ops.attention, q, k1, v1, heads, k2, v2, *args, **kw)
        if not self.enabled:
            return r
        if k2 is None and nq == k1.shape[1] and nq >= 9216:
            self.attn_big[0].append(r[1])
            self.attn_big = (self.attn_big[0], self.attn_big[1] + fl)
        return r[0]

    def attention_fp8(self, q, k, v, heads, *args, **kw):
        B, nq, _ = q.shape
        fl = 4.0 * nq * k.shape[1] * 64 * heads * B
        r = self._timed("attention", fl, self._ops.attention_fp8, q, k, v, heads, *args, **kw)
        if not self.enabled:
            return r
        if nq == k.shape[1] and nq >= 9216:
            self.attn_big[0].append(r[1])
            self.attn_big = (self.attn_big[0], self.attn_big[1] + fl)
        return r[0]

    def summary(self):
        out = {}
        for fam in KERNELS:
            ms = sum(a.elapsed_time(b) for a, b in self.ev[fam])
            out[fam] = {"ms": ms, "launches": len(self.ev[fam]), "flops": self.fl[fam], "bytes": self.by[fam],
                        "roofline_s": self.rt[fam]}
        ev, fl = self.attn_big
        big = {"ms": sum(a.elapsed_time(b) for a, b in ev), "launches": len(ev), "flops": fl}
        kern = {name: {"ms": sum(a.elapsed_time(b) for a, b in evs), "launches": len(evs), "flops": kfl}
                for name, (evs, kfl) in self.kern.items()}
        side = {name: {"ms": sum(a.elapsed_time(b) for a, b in evs), "launches": len(evs), "bytes": nb}
                for name, (evs, nb) in self.side.items()}
        return out, big, kern, side


def cpu_baseline(unet, res, ins):
    """One U-Net forward of the oracle (f32 eager restatement of the reference) on the host cores."""
    from oracle import unet_ref
    sd = {k: v.detach().float().cpu() for k, v in unet.state_dict().items()}
    x = torch.cat([ins["x_T"], ins["c_concat"]], 1).float().cpu()
    ctx = ins["c_crossattn"].float().cpu()
    # eager PyTorch on many small ops scales negatively past ~16 threads (measured on the GPU box's
    # 2 x 64-core EPYC: 16 threads 3.8 s, 32: 5.2 s, 64: 11.6 s, 128: 25.7 s for the same forward)
    prev = torch.get_num_threads()
    torch.set_num_threads(min(16, os.cpu_count() or 16))
    t0 = time.time()
    unet_ref.unet_forward(sd, x, torch.tensor([500]), ctx, torch.tensor([15]))
    dt = time.time() - t0
    used = torch.get_num_threads()
    torch.set_num_threads(prev)
    return {"value": 1.0 / (2.0 * dt), "unit": "steps/s", "cores": used, "kind": "port",
            "sample": f"1 of the 2 U-Net forwards of one CFG DDIM step at {res} (f32 oracle, {dt:.1f} s), x2 per step"}


def parity_mode_leg(dev):
    """The contract-conformant configuration beside the bf16 headline: f16 operands with [hi | lo] norm
    outputs at the sites that buy error (HipOps(parity="selective")), BASELINE config 1 - 320x512, 10 CFG-4 DDIM steps, eta 0 -
    sampler -> first-stage decode on the kernels, FRAMES against the committed fixture of the REAL reference's chain
    (tests/golden/frames_full_40x64_s10_eta0.npz: data only; seeds 20230211 / 123 = oracle/golden_recipe.py's), and the step
    time of that loop next to the default f16 mode's on the same U-Net weights.  Nothing under oracle/ is used here."""
    import numpy as np
    from open_pandora_amd import factory, synth
    from open_pandora_amd.autoencoder import AutoencoderKL
    from open_pandora_amd.ddim import DDIMSampler
    from open_pandora_amd.ops_hip import HipOps
    path = os.path.join(ROOT, "tests", "golden", "frames_full_40x64_s10_eta0.npz")
    if not os.path.exists(path):
        return None
    g = np.load(path)
    h, w = 40, 64
    ins = synth.synth_inputs(h, w, T, seed=123)
    cond = {"c_crossattn": [ins["c_crossattn"].to(dev)], "c_concat": [ins["c_concat"].to(dev)]}
    uc = {"c_crossattn": [ins["uc_crossattn"].to(dev)], "c_concat": [ins["c_concat"].to(dev)]}
    kw = dict(batch_size=1, shape=(4, T, h, w), conditioning=cond, verbose=False, unconditional_guidance_scale=4.0,
              unconditional_conditioning=uc, eta=0.0, fs=torch.tensor([15], device=dev), timestep_spacing="uniform_trailing",
              x_T=ins["x_T"].to(dev))

    def rel(t, key):
        y = t.detach().float().cpu().reshape(-1)
        if f"{key}/full" in g:
            ref = torch.from_numpy(g[f"{key}/full"]).reshape(-1)
        else:
            ref = torch.from_numpy(g[f"{key}/slice"])
            y = y[::int(g[f"{key}/stride"])][:ref.numel()]
        return float((y.double() - ref.double()).norm() / ref.double().norm())

    out = {}
    pm = None
    for name, parity in (("default", False), ("selective", "selective")):
        ops = HipOps(torch.float16, dev, parity=parity)
        if pm is None:
            pm = factory.build_diffusion("320x512", ops, seed=20230211)
        else:
            pm.model.diffusion_model.bind(ops)  # the same weights behind the other op table (re-packed)
        smp = DDIMSampler(pm)
        z, _ = smp.sample(S=10, **kw)           # (warm-up + graph capture + the result)
        ms = 1e9
        for _ in range(2):                      # the same 10-step loop again, graph-replayed: best of two
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            smp.sample(S=10, **kw)
            torch.cuda.synchronize()
            ms = min(ms, 1e3 * (time.perf_counter() - t0) / 10)
        smp.close()
        ae = AutoencoderKL()
        ae.load_state_dict({k: synth.synth_tensor(k, tuple(v.shape), 20230211, dev) for k, v in ae.state_dict().items()})
        frames = ae.bind(HipOps(torch.float16, dev, parity=bool(parity))).decode_first_stage(z)
        out[name] = {"ms_per_step": ms, "steps_per_s": 1e3 / ms, "latent_rel_err_fixture": rel(z, "latent"),
                     "frames_rel_err_fixture": rel(frames, "frames")}
        del ae, frames
    del pm
    torch.cuda.empty_cache()
    sel = out["selective"]
    res = {"dtype": "f16", "parity": "selective", "config": "320x512, 16 frames, 10 CFG-4 DDIM steps, eta 0 (BASELINE configs[0]'s loop)",
           "steps_per_s": sel["steps_per_s"], "ms_per_step": sel["ms_per_step"],
           "frames_rel_err_fixture": sel["frames_rel_err_fixture"], "latent_rel_err_fixture": sel["latent_rel_err_fixture"],
           "tolerance": 1e-3, "within_tolerance": sel["frames_rel_err_fixture"] <= 1e-3,
           "margin_to_tolerance": 1.0 - sel["frames_rel_err_fixture"] / 1e-3,
           "step_vs_default_f16": sel["ms_per_step"] / out["default"]["ms_per_step"], "default_f16": out["default"],
           "fixture": "tests/golden/frames_full_40x64_s10_eta0.npz (the real reference's DDIMSampler.sample -> decode_first_stage, f32 CPU)"}
    # r06: the site list was chosen ON that fixture's seeds; the held-out pair (other weights, other
    # inputs, same reference chain; not used for any tuning) says what the configuration does without fitting: margins, not a flag
    hp = os.path.join(ROOT, "tests", "golden", "frames_full_40x64_s10_eta0_w777001_i456.npz")
    if os.path.exists(hp):
        g = np.load(hp)  # (rel() above reads `g`)
        ins = synth.synth_inputs(h, w, T, seed=int(g["input_seed"]))
        cond = {"c_crossattn": [ins["c_crossattn"].to(dev)], "c_concat": [ins["c_concat"].to(dev)]}
        uc = {"c_crossattn": [ins["uc_crossattn"].to(dev)], "c_concat": [ins["c_concat"].to(dev)]}
        kw.update(conditioning=cond, unconditional_conditioning=uc, x_T=ins["x_T"].to(dev))
        wseed = int(g["weight_seed"])
        ops = HipOps(torch.float16, dev, parity="selective")
        pm = factory.build_diffusion("320x512", ops, seed=wseed)
        smp = DDIMSampler(pm)
        z, _ = smp.sample(S=10, **kw)
        smp.close()
        ae = AutoencoderKL()
        ae.load_state_dict({k: synth.synth_tensor(k, tuple(v.shape), wseed, dev) for k, v in ae.state_dict().items()})
        frames = ae.bind(HipOps(torch.float16, dev, parity=True)).decode_first_stage(z)
        e_z, e_f = rel(z, "latent"), rel(frames, "frames")
        res["held_out_seeds"] = {"weight_seed": wseed, "input_seed": int(g["input_seed"]), "latent_rel_err_fixture": e_z,
                                 "frames_rel_err_fixture": e_f, "within_tolerance": e_f <= 1e-3,
                                 "margin_to_tolerance": 1.0 - e_f / 1e-3,
                                 "fixture": "tests/golden/frames_full_40x64_s10_eta0_w777001_i456.npz (same chain, seeds no site was chosen on)"}
        del pm, ae, frames
        torch.cuda.empty_cache()
    return res


def committed_traffic(res, fam="gemm"):
    """HBM-side bytes per launch of a kernel family from the committed PMC summary (separate rocprofv3 --pmc passes over
    one eager forward, gfx950 FETCH_SIZE correction applied there: tools/pmc_traffic.py) - only when it was taken with
    the library sources of THIS run; otherwise null."""
    path = os.path.join(ROOT, "profiles", "r06", "pmc_traffic.json")
    try:
        from open_pandora_amd import build as _b
        with open(path) as f:
            t = json.load(f)
        if t.get("lib_digest") != _b._digest():
            return None
        return t[res][fam]["bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        return None


def compute_scaling(pm_of, unet, ops, dev, reps=6):
    """One rank's kernels of a frame-sharded step, on this one GPU (VERDICT r02 #3a): the U-Net forward, replayed as a
    HIP graph, on clips of 16 / N frames.  N-way frame shards run exactly these shapes per rank (the temporal blocks
    re-shard to all 16 frames x P/N pixels: the same token count), so t(16) / t(16/N) bounds the speed-up of the
    compute side; exchanges come on top."""
    from open_pandora_amd import factory, synth
    from open_pandora_amd.ddim import _ForwardGraph
    out = {}
    for res in ("320x512", "576x1024"):
        h, w = factory.RESOLUTIONS[res]["image_size"]
        pm = pm_of(res)
        row = {}
        for n in (1, 2, 4, 8):
            t = T // n
            ins = synth.synth_inputs(h, w, t, seed=123)
            cond = {"c_crossattn": [ins["c_crossattn"].to(dev)], "c_concat": [ins["c_concat"].to(dev)]}
            uc = {"c_crossattn": [ins["uc_crossattn"].to(dev)], "c_concat": [ins["c_concat"].to(dev)]}
            x, ts, fs = ins["x_T"].to(dev), torch.full((1,), 500, device=dev, dtype=torch.long), torch.tensor([15], device=dev)
            entry = {"frames": t}
            for tag, u in (("one_forward_ms", None), ("cfg_pair_two_streams_ms", uc), ("cfg_pair_batched_ms", uc)):
                # (batched: the pair as ONE forward over 2 x t frames, PANDORA_CFG_BATCH=1 - weights read once, grids twice as full)
                prev = os.environ.get("PANDORA_CFG_BATCH")
                os.environ["PANDORA_CFG_BATCH"] = "1" if tag == "cfg_pair_batched_ms" else "0"
                try:
                    g = _ForwardGraph(pm, x, ts, cond, u, fs, {})
                finally:
                    if prev is None:
                        os.environ.pop("PANDORA_CFG_BATCH", None)
                    else:
                        os.environ["PANDORA_CFG_BATCH"] = prev
                g(x, ts)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    g(x, ts)
                torch.cuda.synchronize()
                entry[tag] = 1e3 * (time.perf_counter() - t0) / reps
                del g
            row[str(n)] = entry
        base1, base2 = row["1"]["one_forward_ms"], row["1"]["cfg_pair_two_streams_ms"]
        for n, e in row.items():
            e["speedup_one_forward"] = base1 / e["one_forward_ms"]
            e["speedup_cfg_pair"] = base2 / e["cfg_pair_two_streams_ms"]
            e["speedup_cfg_pair_batched"] = row["1"]["cfg_pair_batched_ms"] / e["cfg_pair_batched_ms"]
        # the two 8-GPU decompositions of frame_parallel.make_hybrid, compute side only (1-GPU step = cfg pair on 16 frames)
        for n, e in row.items():
            e["cfg_pair_best_ms"] = min(e["cfg_pair_two_streams_ms"], e["cfg_pair_batched_ms"])
        base2 = row["1"]["cfg_pair_best_ms"]  # (the 1-GPU step = the faster of the two forms on 16 frames)
        row["projection_8gpu"] = {
            "cfg_pair_x_4_frame_shards": base2 / row["4"]["one_forward_ms"],
            "8_frame_shards_both_branches_per_rank": base2 / row["8"]["cfg_pair_two_streams_ms"],
            "8_frame_shards_both_branches_batched_per_rank": base2 / row["8"]["cfg_pair_batched_ms"],
            "note": "1-GPU step time / one rank's kernel time at its shard size: an upper bound, exchanges not included"}
        row["projection_4gpu"] = {"cfg_pair_x_2_frame_shards": base2 / row["2"]["one_forward_ms"]}
        row["projection_2gpu"] = {"cfg_pair": base2 / row["1"]["one_forward_ms"]}
        out[res] = row
    return out


def attention_ceiling(ops, dtype, rounds=3, reps=5):
    """What this chip sustains for the attention kernel's per-tile instruction mix with NO global traffic (the ceiling
    probe of csrc/attn.hip, pm_debug_attn_variant 11: K/V tiles resident in LDS, same MFMAs / softmax stream / LDS reads),
    measured beside the production kernel on the same random N = 9216 tensors, interleaved.  The probe's output is not an
    attention result; only its time is used."""
    from open_pandora_amd.ops_hip import HipOps
    # (the probes exist only in the diagnostics build of the library, include/pandora_mi355x_diag.h: both arms of this A/B run on
    # that build - same kernel code as the shipped library for variant 0 - through an op table of its own)
    ops = HipOps(dtype, ops.device, diag=True)
    F, N, heads = 16, 9216, 5
    C = heads * 64
    qkv = torch.randn(F, N, 3 * C, device=ops.device, dtype=dtype)
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    fl = 4.0 * N * N * 64 * heads * F
    best = {0: 1e9, 11: 1e9}
    for r in range(rounds + 1):
        for var in (0, 11):
            ops.lib.pm_debug_attn_variant(var)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ops.attention(q, k, v, heads)
            e1.record()
            torch.cuda.synchronize()
            if r:
                best[var] = min(best[var], e0.elapsed_time(e1) / reps)
    ops.lib.pm_debug_attn_variant(0)
    return {"tflops": fl / best[11] / 1e9, "frac_of_peak": fl / best[11] / 1e9 / MFMA_PEAK_TFLOPS,
            "production_kernel_same_tensors_tflops": fl / best[0] / 1e9,
            "production_over_ceiling": best[11] / best[0],
            "what": "the production kernel's per-tile instruction stream with K/V resident in LDS (no global traffic): "
                    "profiles/r03/attention_ceiling.txt, profiles/r04/attention_shapes.txt"}


def attention_fp8_roofline(ops, dtype, rounds=3, reps=5):
    """BASELINE configs[4]'s named kernel in the default line (VERDICT r05 #6c): pm_attention_fp8 - the e4m3 pack pass AND the
    block-scaled-MFMA attention kernel, every launch inside the HIP-event bracket - on random N = 9216 tensors (16 frames x 5
    heads x head dim 64 = U-Net level 0 at 576x1024), interleaved with the 16-bit kernel on the same tensors.  TF/s-equivalent
    = the attention's 4 N^2 d FLOPs / the whole call; priced against the dense FP8 peak (5 PF)."""
    F, N, heads = 16, 9216, 5
    C = heads * 64
    qkv = torch.randn(F, N, 3 * C, device=ops.device, dtype=dtype)
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    fl = 4.0 * N * N * 64 * heads * F
    best = {"fp8": 1e9, "16bit": 1e9}
    for r in range(rounds + 1):
        for name, fn in (("fp8", lambda: ops.attention_fp8(q, k, v, heads)), ("16bit", lambda: ops.attention(q, k, v, heads))):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            if r:
                best[name] = min(best[name], e0.elapsed_time(e1) / reps)
    ach = fl / best["fp8"] / 1e9
    return {"bound": "mfma", "kernel": "pm_attention_fp8 = attn_fp8_pack_kernel (q, k rows and V^T in MFMA operand order as e4m3) + "
                                       "attn_fp8_kernel (v_mfma_scale_f32_32x32x64_f8f6f4, unit scales): N = 9216 x 16 frames x 5 heads, d 64",
            "achieved": ach, "peak": 2.0 * MFMA_PEAK_TFLOPS, "unit": "TFLOP/s-equivalent (the 16-bit algorithm's FLOPs / the whole call)",
            "frac": ach / (2.0 * MFMA_PEAK_TFLOPS), "ms_per_call_incl_pack": best["fp8"],
            "same_tensors_16bit_kernel_tflops": fl / best["16bit"] / 1e9, "speedup_over_16bit": best["16bit"] / best["fp8"],
            "timed": "HIP events around `reps` whole calls (pack + attention), best of 3 interleaved rounds, random data"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--only", default=None, choices=[None, "320x512", "576x1024"],
                    help="kernel-work runs: measure one resolution (the driver line always carries both)")
    ap.add_argument("--res", default=None, help=argparse.SUPPRESS)  # (round-1 spelling of --only)
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "off"])
    ap.add_argument("--emulate-shard", default="auto", choices=["auto", "off", "only"],
                    help="compute_scaling: the U-Net forward on 16/N-frame clips (N = 2, 4, 8) on this one GPU; "
                         "'only' prints just that object")
    ap.add_argument("--fp8-attention", action="store_true",
                    help="spatial self-attention on pm_attention_fp8 (block-scaled e4m3 MFMA): BASELINE configs[4]")
    ap.add_argument("--multiround", type=int, default=0,
                    help="also time an N-round autoregressive 576x1024 generation (configs[4]: 5 rounds = 10 s of video): "
                         "per round AE-encode 4 frames, 50 CFG DDIM steps, AE-decode 16 frames")
    ap.add_argument("--parity", action="store_true",
                    help="HipOps(parity=True): every GroupNorm / LayerNorm output as [hi | lo] 16-bit parts (2x the MFMA work on the "
                         "ops they feed) - the configuration whose FRAMES meet the north-star's 1e-3 (tests/test_frames_gpu.py); "
                         "use with --dtype f16: its step time goes on record next to the bf16 production number")
    ap.add_argument("--parity-mode", default="auto", choices=["auto", "off"],
                    help="default run at 1 GPU: also the f16 selective-parity configuration - step time + FRAMES error against "
                         "the committed reference fixture - as `parity_mode` (the contract-conformant number beside the bf16 one)")
    ap.add_argument("--split", default="hybrid", choices=["hybrid", "frames-kv"],
                    help="N > 1: 'hybrid' = cond / uncond branch pair x N/2 frame shards, temporal blocks re-sharded frames <-> pixels "
                         "(frame_parallel.make_hybrid's default); 'frames-kv' = the north-star's literal split - N frame shards, "
                         "both branches per rank, K|V all-gather for the temporal attention")
    ap.add_argument("--rehearsal-width", type=int, default=0, help=argparse.SUPPRESS)  # (functional rehearsals of the N > 1 code
    #   path on ONE GPU, tests/test_bench_rehearsal_gpu.py: a U-Net of that many base channels; the line says so and is not a
    #   measurement of the named model)
    a = ap.parse_args()
    only = a.only or a.res

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world} (launch N>1 with torch.distributed.run)"
    local_rank %= max(1, torch.cuda.device_count())  # (a 1-GPU box can rehearse N ranks over gloo)
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("PANDORA_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(dev))
        else:
            dist.init_process_group(backend)

    from open_pandora_amd import factory, synth
    from open_pandora_amd.ddim import DDIMSampler
    from open_pandora_amd.ddpm import LatentVisualDiffusion
    from open_pandora_amd.ops_hip import HipOps

    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    ops = TimedOps(HipOps(dt, dev, fp8_attention=a.fp8_attention, parity=a.parity))
    pm0 = factory.build_diffusion("320x512", ops,  # the two shipped configs share the U-Net (1 516 tensors)
                                  unet_overrides=dict(model_channels=a.rehearsal_width) if a.rehearsal_width else None)
    unet = pm0.model.diffusion_model
    fp = cfgp = None
    mode = "1 GPU"
    if world > 1:
        from open_pandora_amd.frame_parallel import make_hybrid
        # peer mailboxes sized once for the largest boundary frame of the run (f32 [72*128, 320]): no re-creation between
        # the two resolutions, whose recorded graphs hold the mailbox addresses
        fp, cfgp = make_hybrid(T, ops=ops, halo_bytes=4 * 72 * 128 * 320,
                               **(dict(use_cfg=False, kv_gather=True) if a.split == "frames-kv" else {}))
        unet.bind(ops, fp)
        fw = 1 if fp is None else fp.world
        mode = (f"{'cond/uncond branch pair x ' if cfgp is not None else ''}{fw}-way frame shards "
                f"({T // fw} frames/GPU), RCCL: 1 output exchange/step"
                + ("" if fp is None else " + (T,H,W)-GN all-reduce, temporal-conv halo exchange, frames<->pixels "
                                         "all-to-all around each TemporalTransformer"))
        # both partners of a CFG pair must draw the same DDIM noise for their (shared) frame shard
        torch.manual_seed(1234 + (0 if fp is None else fp.rank))
        torch.cuda.manual_seed(1234 + (0 if fp is None else fp.rank))

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    _pm = {"320x512": pm0}

    def pm_of(res):
        if res not in _pm:  # same U-Net behind the other yaml's shell (base_scale of the dynamic rescale, image size)
            r = dict(factory.RESOLUTIONS[res])
            r.pop("default_fs")
            _pm[res] = LatentVisualDiffusion(unet, linear_start=0.00085, linear_end=0.012, timesteps=1000,
                                             parameterization="v", rescale_betas_zero_snr=True, conditioning_key="hybrid",
                                             use_dynamic_rescale=True, scale_factor=0.18215, channels=4, **r)
        return _pm[res]

    if a.emulate_shard == "only":
        assert world == 1
        print(json.dumps({"compute_scaling": compute_scaling(pm_of, unet, ops, dev), "dtype": a.dtype}), flush=True)
        return

    def run_resolution(res, steps, warmup):
        h, w = factory.RESOLUTIONS[res]["image_size"]
        pm = pm_of(res)
        ins = synth.synth_inputs(h, w, T, seed=123)
        cond = {"c_crossattn": [ins["c_crossattn"].to(dev)], "c_concat": [ins["c_concat"].to(dev)]}
        uc = {"c_crossattn": [ins["uc_crossattn"].to(dev)], "c_concat": [ins["c_concat"].to(dev)]}
        x = ins["x_T"].to(dev)
        if fp is not None:
            cond["c_concat"] = [fp.shard_frames(cond["c_concat"][0])]
            uc["c_concat"] = [fp.shard_frames(uc["c_concat"][0])]
            x = fp.shard_frames(x)
        fs = torch.tensor([15], device=dev)
        S = 50
        smp = DDIMSampler(pm, cfg_parallel=cfgp)
        smp.make_schedule(S, "uniform_trailing", 1.0, verbose=False)
        order = list(reversed(range(S)))  # index of the i-th loop iteration

        def run(n, start):
            nonlocal x
            for j in range(n):
                index = order[(start + j) % S]
                step = int(smp.ddim_timesteps[index])
                ts = torch.full((1,), step, device=dev, dtype=torch.long)
                x, _ = smp.p_sample_ddim(x, cond, ts, index, unconditional_guidance_scale=4.0,
                                         unconditional_conditioning=uc, fs=fs, step=step, want_x0=False)

        run(warmup, 0)
        barrier()
        t0 = time.perf_counter()
        run(steps, warmup)
        barrier()
        elapsed = time.perf_counter() - t0
        # Roofline leg: the timed steps replay a HIP graph (events cannot bracket nodes of a captured graph), so
        # the kernel families are timed right after, live, with HIP events around every one of their launches on
        # the launch stream during two more, eagerly launched steps of the same loop.
        smp.use_graph = False
        if fp is not None:
            for k in fp.calls:
                fp.calls[k] = 0
        fwd_count = [0]
        run(1, warmup + steps)
        fwd_count[0] = 3 * (1 if cfgp is not None else 2)  # the 1 + 2 eager steps below, forwards per step on this rank
        ops.reset()
        ops.enabled = True
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        run(2, warmup + steps + 1)
        s1.record()
        torch.cuda.synchronize()
        ops.enabled = False
        seq_step_ms = s0.elapsed_time(s1) / 2.0  # one eager, single-stream step (what the family times are parts of)
        fams, big, kern, side = ops.summary()
        if world > 1:
            tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            elapsed = tt.item()
        assert torch.isfinite(x).all(), "latent went non-finite"
        step_s = elapsed / steps
        fam_out = {}
        for k, v in fams.items():
            tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0.0
            tbs = v["bytes"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0.0
            fam_out[k] = {"ms_per_step": v["ms"] / 2.0, "launches_per_step": v["launches"] // 2,
                          "tflops": tf, "frac_of_peak": tf / MFMA_PEAK_TFLOPS,
                          # the same launches against the OTHER roofline: algorithmic bytes (operands + output once)
                          "algorithmic_TBps": tbs, "frac_of_hbm_8TBps": tbs / 8.0,
                          # per-launch roofline: every launch priced at max(FLOPs / 2.5 PF, bytes / 8 TB/s)
                          "frac_of_per_launch_roofline": (v["roofline_s"] / (v["ms"] * 1e-3)) if v["ms"] > 0 else 0.0,
                          "share_of_sequential_step": (v["ms"] / 2.0) / seq_step_ms}
        # the HBM-bound side kernels, family by family (their brackets exclude launches made from inside an MFMA-family op,
        # e.g. conv3x3's own split16 pass, which stay in that family's time), then what is left: gaps of the eager issue
        kern_ms = sum(v["ms"] for v in fams.values()) + sum(v["ms"] for v in side.values())
        for k, v in side.items():
            tbs = v["bytes"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0.0
            fam_out[k] = {"ms_per_step": v["ms"] / 2.0, "launches_per_step": v["launches"] // 2, "bound": "hbm",
                          "algorithmic_TBps": tbs, "frac_of_hbm_8TBps": tbs / 8.0, "kernel": SIDE[k],
                          "share_of_sequential_step": (v["ms"] / 2.0) / seq_step_ms}
        for k in fam_out:  # shares of the summed KERNEL time: independent of how the step's launches are issued / overlapped
            fam_out[k]["share_of_kernel_time"] = fam_out[k]["ms_per_step"] * 2.0 / kern_ms
        fam_out["gaps"] = {"ms_per_step": seq_step_ms - kern_ms / 2.0,
                           "share_of_sequential_step": 1.0 - kern_ms / 2.0 / seq_step_ms,
                           "what": "idle time between the launches of an eagerly issued, single-stream step (host issue + kernel "
                                   "boundaries); the timed steps replay a two-stream HIP graph and do not pay it"}
        fam_out["timed_step"] = {"ms_per_step": 1e3 * step_s, "kernel_ms_per_step": kern_ms / 2.0,
                                 "what": "the headline step (graph replay, cond / uncond forwards on two streams) against the summed "
                                         "kernel time of one step: < 1 means the two forwards overlap on the device",
                                 "step_over_kernel_time": 1e3 * step_s / (kern_ms / 2.0)}
        if fp is not None and multi_gpu is not None and "exchanges_per_forward" not in multi_gpu:
            # the eager steps above walked the Python forward: 3 steps x forwards-per-step of this rank since the reset below
            multi_gpu["exchanges_per_forward"] = {k: v / max(1, fwd_count[0]) for k, v in fp.calls.items()}
        smp.close()  # graphs (and the exchange buffers recorded between them) go before the process group does
        return {"res": res, "latent": [T, h, w], "steps": steps, "elapsed": elapsed, "step_s": step_s,
                "seq_step_ms": seq_step_ms, "families": fam_out, "raw": fams, "attn_big": big, "kern": kern, "x": x,
                "ins": ins}

    # N > 1: what the run actually was (VERDICT r03 #6b) - the ranks a real all-reduce saw, and the exchanges one rank's
    # forward issues, counted by FrameParallel during the warm-up / recording forwards of the first resolution
    multi_gpu = None
    if world > 1:
        import torch.distributed as dist
        one = torch.ones(1, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(one)
        multi_gpu = {"rccl_ranks_seen": int(one.item()), "backend": dist.get_backend(),
                     "frame_group_world": 1 if fp is None else fp.world, "cfg_pair": cfgp is not None,
                     "peer_mailbox": bool(fp is not None and fp.mailbox is not None),
                     "scaling_curve_measured": False,
                     "note": "no multi-GPU node has been available to this build: nothing here is a measured scaling curve; the "
                             "driver computes efficiency from its own per-N runs"}
    results = {}
    if only in (None, "320x512"):
        results["320x512"] = run_resolution("320x512", a.steps, a.warmup)
    if only in (None, "576x1024"):
        s1024 = a.steps if only else max(3, min(a.steps, 10))
        results["576x1024"] = run_resolution("576x1024", s1024, min(a.warmup, 2) if not only else a.warmup)
    head = results.get("320x512") or results["576x1024"]

    # first-stage decode of the clip (the step after the loop): reported beside the loop
    decode_ms = None
    if world == 1 and "320x512" in results:
        from open_pandora_amd.autoencoder import AutoencoderKL
        h, w = head["latent"][1:]
        with torch.device("meta"):
            ae = AutoencoderKL()
        ae.load_state_dict({k: synth.synth_tensor(k, tuple(v.shape), 20230211, dev) for k, v in ae.state_dict().items()},
                           assign=True)
        ae.bind(ops)
        z = (0.18215 * head["x"]).contiguous()
        ae.decode_first_stage(z)  # warm-up (weight packing)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        frames = ae.decode_first_stage(z)
        torch.cuda.synchronize()
        decode_ms = 1e3 * (time.perf_counter() - t1)
        assert frames.shape == (1, 3, T, 8 * h, 8 * w) and torch.isfinite(frames).all()

    # BASELINE configs[4]: N-round autoregressive generation at 576x1024 through the caller surface
    # (wm.DiffusionRunner.generate_multiround = ChatWM.generate_video_mutliround, model.py:1094-1129), 1 GPU
    multi = None
    if a.multiround and world == 1:
        from open_pandora_amd import wm
        from open_pandora_amd.autoencoder import AutoencoderKL
        from open_pandora_amd.ddpm import LatentVisualDiffusion
        r = dict(factory.RESOLUTIONS["576x1024"])
        r.pop("default_fs")
        pm2 = LatentVisualDiffusion(unet, linear_start=0.00085, linear_end=0.012, timesteps=1000, parameterization="v",
                                    rescale_betas_zero_snr=True, conditioning_key="hybrid", use_dynamic_rescale=True,
                                    scale_factor=0.18215, channels=4, **r)
        with torch.device("meta"):
            ae = AutoencoderKL()
        ae.load_state_dict({k: synth.synth_tensor(k, tuple(v.shape), 20230211, dev) for k, v in ae.state_dict().items()},
                           assign=True)
        ae.bind(ops)
        ins = synth.synth_inputs(72, 128, T, seed=123)
        text, img = ins["c_crossattn"][:, :77].to(dev), ins["c_crossattn"][:, 77:].to(dev)
        uct, uci = ins["uc_crossattn"][:, :77].to(dev), ins["uc_crossattn"][:, 77:].to(dev)
        runner = wm.DiffusionRunner(pm2, lambda im: img if float(im.abs().sum()) > 0 else uci, uct,
                                    ae.encode_first_stage, ae.decode_first_stage)
        frame0 = synth.uniform_pm1(3 * 576 * 1024, 123, "bench/frame0", dev).reshape(3, 1, 576, 1024)
        kw = dict(n_samples=1, ddim_steps=50, ddim_eta=1.0, unconditional_guidance_scale=4.0, fs=15,
                  timestep_spacing="uniform_trailing")
        runner.generate_multiround([text], frame0, frame0[None, :, 0], **dict(kw, ddim_steps=2))  # warm-up (graph, AE packing)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        video = runner.generate_multiround([text] * a.multiround, frame0, frame0[None, :, 0], **kw)
        torch.cuda.synchronize()
        dtm = time.perf_counter() - t0
        assert video.shape == (1, 1, 3, 12 * (a.multiround - 1) + 16, 576, 1024) and torch.isfinite(video).all()
        multi = {"rounds": a.multiround, "frames": int(video.shape[3]), "seconds": dtm,
                 "sec_per_round": dtm / a.multiround, "resolution": "576x1024", "ddim_steps": 50, "n_gpus": 1,
                 "includes": "AE encode of the conditioning frames + 50 CFG DDIM steps + AE decode of 16 frames per round"}

    if rank == 0:
        def roof(r):
            """the family with the largest measured share of the step"""
            fam = max(r["raw"], key=lambda k: r["raw"][k]["ms"])
            v = r["raw"][fam]
            ach = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0.0
            dom = None
            if r["kern"]:
                kn = max(r["kern"], key=lambda k: r["kern"][k]["ms"])
                kv = r["kern"][kn]
                ktf = kv["flops"] / (kv["ms"] * 1e-3) / 1e12 if kv["ms"] > 0 else 0.0
                dom = {"kernel": kn, "ms_per_step": kv["ms"] / 2.0, "launches_per_step": kv["launches"] // 2,
                       "achieved": ktf, "unit": "TFLOP/s", "frac": ktf / MFMA_PEAK_TFLOPS,
                       "share_of_sequential_step": kv["ms"] / 2.0 / r["seq_step_ms"],
                       "all_dense_kernels": {n: {"ms_per_step": x["ms"] / 2.0, "tflops": (x["flops"] / (x["ms"] * 1e-3) / 1e12
                                                                                        if x["ms"] > 0 else 0.0)}
                                             for n, x in r["kern"].items()}}
            return {"bound": "mfma", "kernel": KERNELS[fam], "achieved": ach, "peak": MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS, "traffic": committed_traffic(r["res"], fam),
                    "algorithmic_bytes_per_launch": (r["families"][fam]["algorithmic_TBps"] * 1e12 * v["ms"] * 1e-3
                                                     / max(1, v["launches"])),
                    "dominant_kernel": dom, "sequential_step_ms": r["seq_step_ms"],
                    "hbm_view": {"achieved": r["families"][fam]["algorithmic_TBps"], "peak": 8.0, "unit": "TB/s",
                                 "frac": r["families"][fam]["frac_of_hbm_8TBps"],
                                 "frac_of_per_launch_roofline": r["families"][fam]["frac_of_per_launch_roofline"],
                                 "note": "algorithmic bytes of the same launches / the same time: the family mixes "
                                         "MFMA-bound and HBM-bound shapes (DESIGN.md section 3)"},
                    "launches": v["launches"], "avg_launch_ms": v["ms"] / max(1, v["launches"]),
                    "share_of_sequential_step": r["families"][fam]["share_of_sequential_step"],
                    "chosen_by": "largest summed HIP-event time among the MFMA kernel families of this run",
                    "families": r["families"]}

        res = head["res"]
        out = {
            "metric": "denoising_steps_per_sec", "value": 1.0 / head["step_s"], "unit": "steps/s",
            "n_gpus": world, "steps": head["steps"], "warmup": a.warmup, "ms_per_step": 1e3 * head["step_s"],
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": a.dtype,
            "data": "synthetic",
            "config": {"workload": f"{res} (latent {T}x{head['latent'][1]}x{head['latent'][2]}), 16 frames, 50-step DDIM "
                                   f"schedule (eta 1.0, uniform_trailing), cfg 4.0 => 2 U-Net forwards/step, 1.44 B-parameter "
                                   f"U-Net = BASELINE configs[{1 if res == '320x512' else 2}]; the same run also measures "
                                   f"{'576x1024 (configs[2]) -> res_576x1024' if only is None else 'only this resolution'}",
                       "latent": head["latent"], "parallelism": mode},
            "sec_per_2s_video": 50.0 * head["step_s"],
            "ae_decode_ms_16_frames": decode_ms,
            "sec_per_2s_video_incl_decode": None if decode_ms is None else 50.0 * head["step_s"] + 1e-3 * decode_ms,
            "whole_step_mfma_frac": FLOP_PER_STEP[res] / head["step_s"] / 1e12 / MFMA_PEAK_TFLOPS / world,
            "roofline": roof(head),
        }
        r2 = results.get("576x1024")
        if r2 is not None and r2 is not head:
            out["res_576x1024"] = {
                "steps_per_s": 1.0 / r2["step_s"], "ms_per_step": 1e3 * r2["step_s"], "steps": r2["steps"],
                "sec_per_2s_video": 50.0 * r2["step_s"], "latent": r2["latent"],
                "whole_step_mfma_frac": FLOP_PER_STEP["576x1024"] / r2["step_s"] / 1e12 / MFMA_PEAK_TFLOPS / world,
                "roofline": roof(r2)}
        if r2 is not None and r2["attn_big"]["launches"]:
            b = r2["attn_big"]
            ach = b["flops"] / (b["ms"] * 1e-3) / 1e12
            out["roofline_attention"] = {
                "bound": "mfma", "kernel": ("pm_attention_fp8 (attn_fp8_pack_kernel + attn_fp8_kernel, e4m3 on the block-scaled MFMA; "
                                            "priced against the DENSE BF16 peak like the bf16 kernel: the fp8 dense peak is 2x)"
                                            if a.fp8_attention else "pm_attention (attn_self16_kernel: v_mfma_f32_16x16x32, denominators on the matrix pipe)") +
                                           ": spatial self-attention, N = 9216 tokens x 16 frames x 5 heads x head dim 64 "
                                           "(576x1024, U-Net level 0)",
                "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS,
                "traffic": committed_traffic("attention_n9216", "attention"), "launches": b["launches"],
                "avg_launch_ms": b["ms"] / b["launches"], "flops_per_launch": b["flops"] / b["launches"]}
            if world == 1 and not a.fp8_attention:
                out["roofline_attention"]["ceiling"] = attention_ceiling(ops, dt)
                out["roofline_attention_fp8"] = attention_fp8_roofline(ops, dt)
        if a.parity:
            out["config"]["numerics"] = ("HipOps(parity=True): norm outputs carried as [hi | lo] 16-bit parts (the 1e-3-frames "
                                         "configuration, tests/test_frames_gpu.py::test_frames_*_parity_mode)")
        if world > 1:
            out["multi_gpu"] = dict(multi_gpu, split=a.split)
        if a.rehearsal_width:
            out["rehearsal"] = f"U-Net of {a.rehearsal_width} base channels: a functional rehearsal, NOT a measurement of the named model"
            out["value"] = None
        if a.fp8_attention:
            out["config"]["attention"] = "fp8 (e4m3) operands on v_mfma_scale_f32_32x32x64_f8f6f4 for the spatial self-attention"
        if multi is not None:
            out["config4_multiround"] = multi
        if a.emulate_shard == "auto" and world == 1 and only is None:
            out["compute_scaling"] = compute_scaling(pm_of, unet, ops, dev)
        if a.parity_mode == "auto" and world == 1 and only is None and not a.rehearsal_width:
            out["parity_mode"] = parity_mode_leg(dev)
        if a.cpu_baseline == "auto" and world == 1:
            out["cpu_baseline"] = cpu_baseline(unet, res, head["ins"])
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
