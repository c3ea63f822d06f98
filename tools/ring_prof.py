"""In-kernel cycle stamps of the ring GEMM (loader: vmcnt wait / barrier / issue; consumer: barrier /
compute / epilogue).  Builds a -DPM_RING_PROF copy of the library next to the product one.
usage: python tools/ring_prof.py   (on the GPU box)"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gpurun_out", "libpandora_prof.so")
if "--build" in sys.argv or "--rebuild" in sys.argv or not os.path.exists(LIB):
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    sys.path.insert(0, ROOT)
    from open_pandora_amd import build as _b
    src = [os.path.join(ROOT, "open-pandora_amd", "csrc", f) for f in _b.SOURCES]
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-std=c++17",
                    "-DPM_RING_PROF"] + src + ["-o", LIB], check=True)
    if "--build" in sys.argv:
        sys.exit(0)
os.environ["PANDORA_LIB"] = LIB
os.environ["PANDORA_GEMM_RING"] = "2"
os.environ["PANDORA_GEMM256"] = "0"
if "--ringw" in sys.argv:
    os.environ["PANDORA_GEMM_RINGW"] = "2"
import torch  # noqa: E402

sys.path.insert(0, ROOT)
from open_pandora_amd.ops_hip import HipOps  # noqa: E402

ops = HipOps(torch.bfloat16, "cuda:0")
lib = ops.lib
lib.pm_debug_ring_prof.argtypes = [ctypes.c_void_p]
lib.pm_debug_ring_prof.restype = None
prof = torch.zeros(256 * 8 * 4, dtype=torch.int64, device="cuda")
lib.pm_debug_ring_prof(prof.data_ptr())
P = lambda t: t.data_ptr()
def report(tag, us):
    d = prof.view(256, 8, 4).double().cpu()
    nb = int((d[:, 0, 3] > 0).sum())
    cons, load = d[:nb, :4], d[:nb, 4:]
    print(f"{tag}: {us:.1f} us, {nb} blocks, {cons[:, :, 3].mean():.1f} K-steps/block  (clock64 ticks per K-step, mean over waves)")
    print(f"   consumer: barrier {cons[:, :, 0].sum() / cons[:, :, 3].sum():7.0f}  compute {cons[:, :, 1].sum() / cons[:, :, 3].sum():7.0f}"
          f"  epilogue/tile-steps {cons[:, :, 2].sum() / cons[:, :, 3].sum():7.0f}  total/blk {(cons[:, :, :3].sum(2)).mean():9.0f}")
    print(f"   loader  : vmwait  {load[:, :, 0].sum() / load[:, :, 3].sum():7.0f}  barrier {load[:, :, 1].sum() / load[:, :, 3].sum():7.0f}"
          f"  issue {load[:, :, 2].sum() / load[:, :, 3].sum():7.0f}  total/blk {(load[:, :, :3].sum(2)).mean():9.0f}")


# temporal conv, unsplit (as the fused-statistics calls run it)
for F, Pp, C in ((16, 160, 1280), (16, 2560, 320)):
    x = torch.randn(F * Pp, C, device="cuda", dtype=torch.bfloat16)
    wt = torch.randn(C, 3 * C, device="cuda", dtype=torch.bfloat16) * 0.02
    out = torch.empty(F * Pp, C, device="cuda", dtype=torch.float32)
    for it in range(3):
        prof.zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.pm_conv_temporal_k3(P(x), C, None, None, P(wt), None, None, 0, P(out), C, F, Pp, C, C, P(ops.zero_page),
                                     2, ops.dt, None, 0, None, torch.cuda.current_stream().cuda_stream)
        e1.record()
        torch.cuda.synchronize()
        assert rc == 0, rc
    report(f"conv_t3 F={F} P={Pp} C={C}", e0.elapsed_time(e1) * 1e3)

# 3x3 conv (fast mode)
for F, H, W, C in ((16, 40, 64, 320), (16, 20, 32, 640)):
    x = torch.randn(F * H * W, C, device="cuda", dtype=torch.bfloat16)
    wp = torch.randn(C, 9 * C, device="cuda", dtype=torch.bfloat16) * 0.02
    out = torch.empty(F * H * W, C, device="cuda", dtype=torch.float32)
    for it in range(3):
        prof.zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.pm_conv2d_3x3(P(x), C, P(wp), None, None, 0, P(out), C, F, H, W, C, C, 1, 0, 1, P(ops.zero_page), 2, ops.dt,
                               None, 0, None, torch.cuda.current_stream().cuda_stream)
        e1.record()
        torch.cuda.synchronize()
        assert rc == 0, rc
    report(f"conv3x3 F={F} {H}x{W} C={C}", e0.elapsed_time(e1) * 1e3)

for M, N, K in ((8192, 8192, 8192), (9216, 1280, 5120), (36864, 1920, 640), (2560, 1280, 3840), (10240, 640, 2560), (40960, 320, 320), (40960, 960, 320), (40960, 320, 1280), (10240, 640, 640)):
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for it in range(3):
        prof.zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.pm_gemm(P(a), K, P(w), K, None, None, 0, P(out), N, M, N, K, 0, 0, ops.dt, None, 0, None,
                         torch.cuda.current_stream().cuda_stream)  # no workspace: unsplit
        e1.record()
        torch.cuda.synchronize()
        assert rc == 0
    us = e0.elapsed_time(e1) * 1e3
    d = prof.view(256, 8, 4).double().cpu()
    nb = int((d[:, 0, 3] > 0).sum())
    cons, load = d[:nb, :4], d[:nb, 4:]
    steps = cons[:, :, 3].mean()
    print(f"M={M} N={N} K={K}: {us:.1f} us, {nb} blocks, {steps:.1f} K-steps/block  (clock64 ticks per K-step, mean over waves)")
    print(f"   consumer: barrier {cons[:, :, 0].sum() / cons[:, :, 3].sum():7.0f}  compute {cons[:, :, 1].sum() / cons[:, :, 3].sum():7.0f}"
          f"  epilogue/tile-steps {cons[:, :, 2].sum() / cons[:, :, 3].sum():7.0f}  total/blk {(cons[:, :, :3].sum(2)).mean():9.0f}")
    print(f"   loader  : vmwait  {load[:, :, 0].sum() / load[:, :, 3].sum():7.0f}  barrier {load[:, :, 1].sum() / load[:, :, 3].sum():7.0f}"
          f"  issue {load[:, :, 2].sum() / load[:, :, 3].sum():7.0f}  total/blk {(load[:, :, :3].sum(2)).mean():9.0f}")
