"""bench.py - denoising steps/sec of the DDIM / 3-D U-Net hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--res 320x512|576x1024] [--dtype bf16|f16]

One "step" = one DDIM step of a 16-frame clip with classifier-free guidance = 2 U-Net forwards
(25.2 TFLOP at 320x512, 104.7 TFLOP at 576x1024) + the fused update kernel.  Inputs (latent, contexts,
weights) are synthetic (seeded) and resident in HBM before the timed region.  Prints ONE JSON line on
rank 0 (see the driver contract); extra objects: `roofline` (dominant kernel, HIP-event timed inside the
timed region) and `cpu_baseline` (the CPU oracle = reference eager path restated, on the host cores).
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


class TimedOps:
    """HipOps proxy that brackets every launch of the dominant kernel (conv3x3 implicit GEMM) with
    HIP events on the launch stream and counts its algorithmic FLOPs."""

    def __init__(self, ops):
        self._ops = ops
        self.events, self.flops, self.enabled = [], 0.0, False

    def __getattr__(self, k):
        return getattr(self._ops, k)

    def conv3x3(self, x, wp, bias, F, H, W, **kw):
        if not self.enabled:
            return self._ops.conv3x3(x, wp, bias, F, H, W, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = self._ops.conv3x3(x, wp, bias, F, H, W, **kw)
        e1.record()
        self.events.append((e0, e1))
        out = y[0] if isinstance(y, tuple) else y  # (out, GroupNorm totals) when stats are fused
        self.flops += 2.0 * out.shape[0] * out.shape[1] * wp.shape[1]
        return y

    def summary(self):
        ms = sum(a.elapsed_time(b) for a, b in self.events)
        n = max(1, len(self.events))
        return ms, n, self.flops


def cpu_baseline(pm, res, ins, cond):
    """One U-Net forward of the oracle (f32 eager restatement of the reference) on the host cores."""
    from oracle import unet_ref
    unet = pm.model.diffusion_model
    sd = {k: v.detach().float().cpu() for k, v in unet.state_dict().items()}
    x = torch.cat([ins["x_T"], ins["c_concat"]], 1).float().cpu()
    ctx = cond["c_crossattn"][0].float().cpu()
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--res", default="320x512")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "off"])
    a = ap.parse_args()

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
    from open_pandora_amd.ops_hip import HipOps

    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    ops = TimedOps(HipOps(dt, dev))
    pm = factory.build_diffusion(a.res, ops)
    h, w = factory.RESOLUTIONS[a.res]["image_size"]
    T = 16
    fp = cfgp = None
    mode = "1 GPU"
    if world > 1:
        from open_pandora_amd.frame_parallel import make_hybrid
        fp, cfgp = make_hybrid(T)
        pm.model.diffusion_model.bind(ops, fp)
        fw = 1 if fp is None else fp.world
        mode = (f"{'cond/uncond branch pair x ' if cfgp is not None else ''}{fw}-way frame shards "
                f"({T // fw} frames/GPU), RCCL: 1 output exchange/step"
                + ("" if fp is None else " + (T,H,W)-GN all-reduce, temporal-conv halo P2P, temporal K/V all-gather"))
        # both partners of a CFG pair must draw the same DDIM noise for their (shared) frame shard
        torch.manual_seed(1234 + (0 if fp is None else fp.rank))
        torch.cuda.manual_seed(1234 + (0 if fp is None else fp.rank))
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

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    run(a.warmup, 0)
    barrier()
    t0 = time.perf_counter()
    run(a.steps, a.warmup)
    barrier()
    elapsed = time.perf_counter() - t0
    # Roofline leg: the timed steps replay a HIP graph (events cannot bracket nodes of a captured
    # graph), so the dominant kernel is timed right after, live, with HIP events around every one of
    # its launches on the launch stream during two more eagerly launched steps of the same loop.
    smp.use_graph = False
    run(1, a.warmup + a.steps)
    ops.enabled = True
    run(2, a.warmup + a.steps + 1)
    torch.cuda.synchronize()
    ops.enabled = False
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = tt.item()
    assert torch.isfinite(x).all(), "latent went non-finite"

    # first-stage decode of the clip (the step after the loop): reported beside the loop
    decode_ms = None
    if world == 1:
        from open_pandora_amd.autoencoder import AutoencoderKL
        with torch.device("meta"):
            ae = AutoencoderKL()
        ae.load_state_dict({k: synth.synth_tensor(k, tuple(v.shape), 20230211, dev) for k, v in ae.state_dict().items()},
                           assign=True)
        ae.bind(ops)
        z = (0.18215 * x).contiguous()
        ae.decode_first_stage(z)  # warm-up (weight packing)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        frames = ae.decode_first_stage(z)
        torch.cuda.synchronize()
        decode_ms = 1e3 * (time.perf_counter() - t1)
        assert frames.shape == (1, 3, T, 8 * h, 8 * w) and torch.isfinite(frames).all()

    if rank == 0:
        ms, n, fl = ops.summary()
        traffic = None  # per-launch HBM-side bytes of the dominant kernel, from the committed PMC passes
        tj = os.path.join(ROOT, "profiles", "r01", "traffic.json")
        if a.res == "320x512" and os.path.exists(tj):
            with open(tj) as f:
                traffic = json.load(f)["conv3x3"]["traffic_bytes"]
        ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        out = {
            "metric": "denoising_steps_per_sec", "value": a.steps / elapsed, "unit": "steps/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": a.dtype,
            "data": "synthetic",
            "config": {"workload": f"{a.res}, 16 frames, 50-step DDIM schedule (eta 1.0, uniform_trailing), "
                                   f"cfg 4.0 => 2 U-Net forwards/step, 1.44 B-parameter U-Net (BASELINE configs[1] at 320x512)",
                       "latent": [T, h, w], "parallelism": mode},
            "sec_per_2s_video": 50.0 * elapsed / a.steps,
            "ae_decode_ms_16_frames": decode_ms,
            "sec_per_2s_video_incl_decode": None if decode_ms is None else 50.0 * elapsed / a.steps + 1e-3 * decode_ms,
            "whole_step_mfma_frac": FLOP_PER_STEP[a.res] / (elapsed / a.steps) / 1e12 / MFMA_PEAK_TFLOPS / world,
            "roofline": {"bound": "mfma", "kernel": "pm_conv2d_3x3 (gemm_ring_kernel<A_CONV3X3_FAST>; gemm_kernel for the f32-operand and strided/upsampling convs)",
                         "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS,
                         "traffic": traffic, "launches": n, "avg_launch_ms": ms / n,
                         "share_of_step_time": (ms * 1e-3 / 2.0) / (elapsed / a.steps)},
        }
        if a.cpu_baseline == "auto" and world == 1:
            out["cpu_baseline"] = cpu_baseline(pm, a.res, ins, {"c_crossattn": [ins["c_crossattn"]]})
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
