"""Per-shape time table of one eagerly launched U-Net forward (run on the GPU box):
every op-table call is bracketed with HIP events and grouped by (op, M, N, K, flags).
usage: python tools/shape_profile.py [--res 320x512] [--dtype bf16] [--reps 3]"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pandora_amd import factory, synth  # noqa: E402
from open_pandora_amd.ops_hip import HipOps  # noqa: E402

PEAK_F, PEAK_B = 2.5e15, 6.3e12  # dense bf16 MFMA, achievable HBM


class Prof:
    def __init__(self, ops):
        self._ops, self.rec, self.on = ops, [], False

    def __getattr__(self, name):
        fn = getattr(self._ops, name)
        if not callable(fn) or name.startswith("_") or name in ("groupnorm_nchunks",):
            return fn

        def wrapped(*a, **kw):
            if not self.on:
                return fn(*a, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = fn(*a, **kw)
            e1.record()
            self.rec.append((self._key(name, a, kw, y), e0, e1))
            return y
        return wrapped

    @staticmethod
    def _key(name, a, kw, y):
        out = y[0] if isinstance(y, tuple) else y
        esz = lambda t: t.numel() * t.element_size()
        if name in ("gemm", "conv3x3", "conv_t3"):
            x, w = a[0], a[1]
            M, N, K = out.shape[0], out.shape[1], w.shape[1]
            res = kw.get("residual", a[3] if name == "gemm" and len(a) > 3 else None)
            byts = esz(x) + esz(w) + esz(out) + (esz(res) if res is not None else 0)
            tag = f"{name} M={M} N={w.shape[0]} K={K} a={str(x.dtype)[6:]} o={str(out.dtype)[6:]}" \
                  f"{' act=' + kw['act'] if kw.get('act') else ''}{' res' if res is not None else ''}{' stats' if kw.get('stats') else ''}"
            return tag, 2.0 * M * w.shape[0] * K, byts
        if name == "ln_gemm":  # LayerNorm + projection (one kernel at K = 320, else the pm_layernorm + pm_gemm pair)
            x, w = a[0], a[3]
            byts = esz(x) + esz(w) + esz(out)
            return f"ln_gemm M={x.shape[0]} N={w.shape[0]} K={w.shape[1]} a={str(x.dtype)[6:]} o={str(out.dtype)[6:]}" \
                   f"{' act=' + kw['act'] if kw.get('act') else ''}", 2.0 * x.shape[0] * w.shape[0] * w.shape[1], byts
        fl = 0.0
        if name in ("attention", "attention_fp8"):  # (F, Nq, C) x (F, Nk, C): QK^T and PV, 2 FLOP per MAC (an MFMA-bound op:
            q, k = a[0], a[1]                        #  its floor is FLOPs / peak, not its operand bytes)
            fl = 4.0 * q.shape[0] * q.shape[1] * k.shape[1] * q.shape[2]
        elif name == "attention_temporal":           # (T, P, C): every pixel attends over the T frames
            q = a[0]
            fl = 4.0 * q.shape[0] * q.shape[0] * q.shape[1] * q.shape[2]
        tens = [t for t in list(a) + list(kw.values()) if isinstance(t, torch.Tensor)]
        big = max(tens, key=lambda t: t.numel()) if tens else None
        byts = sum(esz(t) for t in tens) + (esz(out) if isinstance(out, torch.Tensor) else 0)
        return f"{name} {tuple(big.shape) if big is not None else ''} {str(big.dtype)[6:] if big is not None else ''}", fl, byts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", default="320x512")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--frames", type=int, default=16, help="clip length: 16 / N = one rank's share of an N-way frame-sharded step")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    ops = Prof(HipOps(dt, "cuda:0"))
    pm = factory.build_diffusion(a.res, ops)
    h, w = factory.RESOLUTIONS[a.res]["image_size"]
    ins = synth.synth_inputs(h, w, a.frames, seed=123)
    cond = {"c_crossattn": [ins["c_crossattn"].cuda()], "c_concat": [ins["c_concat"].cuda()]}
    x = ins["x_T"].cuda()
    ts = torch.full((1,), 500, device="cuda", dtype=torch.long)
    fs = torch.tensor([15], device="cuda")
    for _ in range(2):
        pm.apply_model(x, ts, cond, fs=fs)
    torch.cuda.synchronize()
    ops.on = True
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(a.reps):
        pm.apply_model(x, ts, cond, fs=fs)
    t1.record()
    torch.cuda.synchronize()
    agg = collections.OrderedDict()
    for (tag, fl, by), e0, e1 in ops.rec:
        r = agg.setdefault(tag, [0, 0.0, fl, by])
        r[0] += 1
        r[1] += e0.elapsed_time(e1)
    tot = sum(r[1] for r in agg.values()) / a.reps
    print(f"# {a.res} {a.dtype} {a.frames} frames: eager forward {t0.elapsed_time(t1) / a.reps:.2f} ms, sum of op brackets {tot:.2f} ms")
    print(f"{'op':78s} {'n':>4s} {'ms/fwd':>7s} {'us':>7s} {'TF/s':>6s} {'GB/s':>6s} {'floor_us':>8s} {'x':>5s}")
    floor_tot = 0.0
    for tag, (n, ms, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        n //= a.reps
        us = 1e3 * ms / (n * a.reps)
        floor = 1e6 * max(fl / PEAK_F, by / PEAK_B)
        floor_tot += floor * n
        print(f"{tag:78s} {n:4d} {ms / a.reps:7.3f} {us:7.1f} {fl / us / 1e6:6.0f} {by / us / 1e3:6.0f} {floor:8.1f} {us / max(floor, 1e-3):5.1f}")
    print(f"# sum of floors {floor_tot / 1e3:.2f} ms/fwd")


if __name__ == "__main__":
    main()
