"""gemm_wide (256x256 tile, 128x128 wave tiles, assembly main loop) against the other pm_gemm kernels and the vendor library:
correctness of every epilogue flavour against an f32 product, then interleaved timing rounds in ONE process (MI355X devices and
clocks differ between processes), random data.  Runs on the diagnostics build (pm_debug_gemm_wide switches the kernel choice).
usage: python tools/wide_probe.py [--quick] [--reps 20] [--rounds 3]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pandora_amd import capi, packing  # noqa: E402
from open_pandora_amd.ops_hip import HipOps  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--quick", action="store_true")
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--skip-check", action="store_true")
ap.add_argument("--stream", action="store_true", help="gemm_wide_stream: parity + timing against the other kernels and the vendor library")
ap.add_argument("--stamps", action="store_true", help="in-kernel stamps of loop variant 7 (prologue / loop / epilogue cycles, clock)")
ap.add_argument("--variants", default="", help="comma list of loop variants (diagnostics build) timed on the first shapes")
args = ap.parse_args()


def rel(a, b):
    a, b = HipOps.totals_f32(a).double(), HipOps.totals_f32(b).double()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


def check(dtype):
    ops = HipOps(dtype, "cuda:0", diag=True)
    g = torch.Generator(device="cuda").manual_seed(1)
    rn = lambda *s, dt=dtype, sc=1.0: (torch.randn(*s, device="cuda", generator=g) * sc).to(dt)
    bad = 0
    for (M, N, K) in [(512, 512, 128), (256, 256, 640), (300, 520, 192), (1024, 768, 1280), (2560, 1280, 1280), (777, 1000, 320)]:
        a, w = rn(M, K), rn(N, K, sc=K ** -0.5)
        bias = rn(N, dt=torch.float32)
        res16, res32 = rn(M, N), rn(M, N, dt=torch.float32)
        want = a.float() @ w.float().t()
        cases = [("plain", dict(), want), ("bias", dict(bias=bias), want + bias),
                 ("bias+res16", dict(bias=bias, residual=res16), want + bias + res16.float()),
                 ("stream f32 + res32", dict(bias=bias, residual=res32, stream=True), want + bias + res32),
                 ("silu", dict(bias=bias, act="silu"), torch.nn.functional.silu(want + bias)),
                 ("gelu", dict(bias=bias, act="gelu"), torch.nn.functional.gelu(want + bias)),
                 ("stream + stats", dict(bias=bias, stream=True, stats=(1, 32)) if (M % 64 == 0 and N % 32 == 0) else None, want + bias)]
        if N % 32 == 0:
            wp, bp = packing.pack_geglu(w.cpu(), bias.cpu())
            xv, gate = (want + bias).chunk(2, dim=-1)
            cases.append(("geglu", dict(_w=wp.cuda(), bias=bp.cuda(), act="geglu"), xv * torch.nn.functional.gelu(gate)))
        for name, kw, ref in cases:
            if kw is None:
                continue
            kw = dict(kw)
            ww = kw.pop("_w", w)
            outs = {}
            for mode in (0, 2):
                ops.lib.pm_debug_gemm_wide(mode)
                y = ops.gemm(a, ww, **kw)
                tot = None
                if isinstance(y, tuple):
                    y, tot = y
                torch.cuda.synchronize()
                outs[mode] = (y.float(), tot)
            e0, e2 = rel(outs[0][0], ref), rel(outs[2][0], ref)
            line = f"[check {str(dtype)[6:]}] M={M} N={N} K={K} {name:20s} other {e0:.2e} wide {e2:.2e}"
            if outs[0][1] is not None:
                line += f" stats {rel(outs[2][1], outs[0][1]):.1e}"
                if rel(outs[2][1], outs[0][1]) > 1e-5:
                    bad += 1
            if not (e2 <= max(1.3 * e0, 1e-6)):
                bad += 1
                line += "   <-- MISMATCH"
            print(line, flush=True)
    ops.lib.pm_debug_gemm_wide(1)
    return bad


def bench(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


def timing():
    ops = HipOps(torch.bfloat16, "cuda:0", diag=True)
    shapes = [(8192, 8192, 8192), (9216, 10240, 1280), (36864, 5120, 640), (9216, 3840, 1280), (2560, 3840, 1280),
              (10240, 5120, 640), (2560, 10240, 1280), (9216, 1280, 5120), (36864, 640, 2560), (36864, 1280, 11520),
              (147456, 2560, 320), (4096, 4096, 4096)]
    if args.quick:
        shapes = shapes[:4]
    for M, N, K in shapes:
        a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * K ** -0.5
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        wt = w.t()
        t = {0: [], 2: [], "lib": []}
        for _ in range(args.rounds):
            for mode in (0, 2):
                ops.lib.pm_debug_gemm_wide(mode)
                t[mode].append(bench(lambda: ops.gemm(a, w, out=out), args.reps))
            t["lib"].append(bench(lambda: torch.matmul(a, wt, out=out), args.reps))
        fl = 2.0 * M * N * K
        m = {k: min(v) for k, v in t.items()}
        print(f"M={M:6d} N={N:5d} K={K:5d}: other {m[0]:8.1f} us {fl / m[0] / 1e6:5.0f} TF/s | wide {m[2]:8.1f} us {fl / m[2] / 1e6:5.0f} TF/s"
              f" | lib {m['lib']:8.1f} us {fl / m['lib'] / 1e6:5.0f} TF/s | wide/other {m[2] / m[0]:5.2f} wide/lib {m[2] / m['lib']:5.2f}",
              flush=True)
    ops.lib.pm_debug_gemm_wide(1)


def stream():
    """gemm_wide_stream forced (pm_debug_gemm_wstream(2)) against the other kernels (both assembly kernels off) on whole-tile shapes"""
    bad = 0
    for dtype in (torch.bfloat16, torch.float16):
        ops = HipOps(dtype, "cuda:0", diag=True)
        g = torch.Generator(device="cuda").manual_seed(3)
        rn = lambda *s, dt=dtype, sc=1.0: (torch.randn(*s, device="cuda", generator=g) * sc).to(dt)
        for (M, N, K) in [(256, 256, 256), (512, 768, 320), (1024, 512, 640), (2048, 2048, 1280), (8192, 8192, 448), (36864, 5120, 640)]:
            a, w = rn(M, K), rn(N, K, sc=K ** -0.5)
            bias = rn(N, dt=torch.float32)
            wp, bp = packing.pack_geglu(w.cpu(), bias.cpu())
            want = a.float() @ w.float().t()
            xv, gate = (want + bias).chunk(2, dim=-1)
            cases = [("plain", w, dict(), want), ("bias", w, dict(bias=bias), want + bias),
                     ("geglu", wp.cuda(), dict(bias=bp.cuda(), act="geglu"), xv * torch.nn.functional.gelu(gate))]
            for name, ww, kw, ref in cases:
                outs = {}
                for mode in (0, 2):
                    ops.lib.pm_debug_gemm_wide(0)
                    ops.lib.pm_debug_gemm_wstream(mode)
                    outs[mode] = ops.gemm(a, ww, **kw).float()
                    torch.cuda.synchronize()
                same = torch.equal(outs[0], outs[2])
                e0, e2 = rel(outs[0], ref), rel(outs[2], ref)
                # plain: the same MFMA chain in the same order -> bit-equal; bias / GEGLU: the bias is the accumulators' INITIAL value
                # here (added first, not last): equal to f32 rounding order, judged against the f32 reference
                ok = same if name == "plain" else e2 <= 1.05 * e0 + 1e-6
                bad += 0 if ok else 1
                print(f"[stream check {str(dtype)[6:]}] M={M} N={N} K={K} {name:6s} vs f32 reference: other {e0:.3e} stream {e2:.3e} bit-equal {same}"
                      + ("" if ok else "   <-- MISMATCH"), flush=True)
    print(f"# stream mismatches: {bad}")
    ops = HipOps(torch.bfloat16, "cuda:0", diag=True)
    shapes = [(8192, 8192, 8192, "none"), (9216, 10240, 1280, "geglu"), (36864, 5120, 640, "geglu"), (147456, 2560, 320, "geglu"),
              (10240, 5120, 640, "geglu"), (2560, 10240, 1280, "geglu"), (40960, 2560, 320, "geglu"), (640, 10240, 1280, "geglu"),
              (9216, 3840, 1280, "none"), (2560, 3840, 1280, "none"), (9216, 1280, 5120, "none"), (2560, 1280, 5120, "none"),
              (2560, 1280, 1280, "none"), (10240, 768, 640, "none"), (36864, 1280, 2560, "none"), (4096, 4096, 4096, "none"),
              (40960, 512, 512, "none"), (2304, 10240, 1280, "geglu")]
    if args.quick:
        shapes = shapes[:4]
    for M, N, K, act in shapes:
        a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * K ** -0.5
        bias = torch.randn(N, device="cuda", dtype=torch.float32)
        out = torch.empty(M, N // 2 if act == "geglu" else N, device="cuda", dtype=torch.bfloat16)
        full = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        wt = w.t()
        t = {"other": [], "wide": [], "stream": [], "lib": []}
        for _ in range(args.rounds):
            for name, wd, ws in (("other", 0, 0), ("wide", 2, 0), ("stream", 0, 2)):
                ops.lib.pm_debug_gemm_wide(wd)
                ops.lib.pm_debug_gemm_wstream(ws)
                t[name].append(bench(lambda: ops.gemm(a, w, bias, act=act, out=out), args.reps))
            t["lib"].append(bench(lambda: torch.matmul(a, wt, out=full), args.reps))  # (plain product: the library fuses no GEGLU)
        fl = 2.0 * M * N * K
        m = {k: min(v) for k, v in t.items()}
        print(f"M={M:6d} N={N:5d} K={K:5d} {act:5s}: " + " | ".join(f"{k} {m[k]:7.1f} us {fl / m[k] / 1e6:5.0f}" for k in ("other", "wide", "stream", "lib"))
              + f" | stream/other {m['stream'] / m['other']:4.2f} stream/lib(plain) {m['stream'] / m['lib']:4.2f}", flush=True)
    ops.lib.pm_debug_gemm_wide(1)
    ops.lib.pm_debug_gemm_wstream(1)
    return bad


def variants():
    ops = HipOps(torch.bfloat16, "cuda:0", diag=True)
    vs = [int(v) for v in args.variants.split(",")]
    for M, N, K in [(8192, 8192, 8192), (9216, 10240, 1280), (36864, 5120, 640)]:
        a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * K ** -0.5
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        t = {v: [] for v in vs}
        for _ in range(args.rounds):
            for v in vs:
                ops.lib.pm_debug_gemm_wide(2 + 16 * v)
                t[v].append(bench(lambda: ops.gemm(a, w, out=out), args.reps))
        fl = 2.0 * M * N * K
        print(f"M={M:6d} N={N:5d} K={K:5d}: " + " | ".join(f"V{v} {min(t[v]):7.1f} us {fl / min(t[v]) / 1e6:5.0f}" for v in vs), flush=True)
    ops.lib.pm_debug_gemm_wide(1)


def stamps():
    import time
    ops = HipOps(torch.bfloat16, "cuda:0", diag=True)
    for variant, M, N, K in [(7, 8192, 8192, 8192), (7, 9216, 10240, 1280), (7, 36864, 5120, 640), (7, 4096, 4096, 4096),
                             (8, 9216, 10240, 1280), (9, 9216, 10240, 1280), (10, 9216, 10240, 1280), (11, 9216, 10240, 1280)]:
        if variant == 8:
            print("# epilogue ablations (timing only): variant 8 = accumulators read, no epilogue; 9 = the generic epilogue, stores masked off; 10 = the generic epilogue on every flavour (7 = lean where legal); 11 = lean without its stores")
        a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * K ** -0.5
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        buf = torch.zeros(256 * 8, dtype=torch.int64, device="cuda")
        ops.lib.pm_debug_gemm_wide(2 + 16 * variant)
        ops.lib.pm_debug_wide_stamps(buf.data_ptr())
        t0 = time.time()
        while time.time() - t0 < 2.0:  # (the clock under sustained load: MI355X_MICROARCH.md DVFS item 6)
            for _ in range(20):
                ops.gemm(a, w, out=out)
            torch.cuda.synchronize()
        buf.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.gemm(a, w, out=out)
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1)
        b = buf.view(256, 8).double().cpu()
        b = b[b[:, 4] > 0]
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        steps, pro, loop, epi = b[:, 4].sum(), b[:, 0].sum(), b[:, 1].sum(), b[:, 2].sum()
        per_wg_tiles = b[:, 4] / (K // 64)
        print(f"[V{variant}] M={M} N={N} K={K}: {us:.1f} us; per tile: prologue {pro / tiles:.0f} cyc, loop {loop / steps:.0f} cyc/K-step "
              f"(2048 = MFMA-bound), epilogue {epi / tiles:.0f} cyc; tiles per workgroup {per_wg_tiles.min():.0f}..{per_wg_tiles.max():.0f}; "
              f"clock ~ {(pro + loop + epi) / len(b) / us / 1e3:.2f} GHz (cycles of the busiest-average workgroup / wall)", flush=True)
    ops.lib.pm_debug_wide_stamps(0)
    ops.lib.pm_debug_gemm_wide(1)


if __name__ == "__main__":
    if args.stream:
        sys.exit(1 if stream() else 0)
    if args.stamps:
        stamps()
        sys.exit(0)
    if args.variants:
        variants()
        sys.exit(0)
    bad = 0
    if not args.skip_check:
        bad = check(torch.bfloat16) + check(torch.float16)
        print(f"# mismatches: {bad}")
    timing()
    sys.exit(1 if bad else 0)
