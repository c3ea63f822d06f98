"""What shader clock does the chip hold under the GEMM / attention kernels?  Runs one op in a loop for a few seconds per case
and samples `rocm-smi --showclocks` beside it (the sclk line).  usage: python tools/clock_under_load.py"""
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pandora_amd.ops_hip import HipOps  # noqa: E402

ops = HipOps(torch.bfloat16, "cuda:0")


def sample(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
            m = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", txt) or re.search(r"sclk[^\n]*?(\d+)Mhz", txt)
            if m:
                out.append(int(m.group(1)))
        except Exception:  # noqa: BLE001
            pass
        time.sleep(0.2)


def case(name, fn, flops, seconds=3.0):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    stop, clocks = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, clocks))
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n += 20
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    mid = sorted(clocks)[len(clocks) // 2] if clocks else -1
    print(f"{name:46s}: {flops * n / dt / 1e12:7.0f} TF/s   sclk samples {min(clocks) if clocks else -1}..{max(clocks) if clocks else -1} MHz (median {mid}, {len(clocks)} samples)")


r = lambda *s: torch.randn(*s, device="cuda", dtype=torch.bfloat16)
a, w = r(8192, 8192), r(8192, 8192) * 0.01
os.environ.setdefault("X", "0")
case("idle (no kernel)", lambda: None, 0.0, 1.5)
case("pm_gemm 8192^3 (kernel by pm_gemm's choice)", lambda: ops.gemm(a, w), 2.0 * 8192 ** 3)
case("torch.matmul 8192^3 (hipBLASLt)", lambda: torch.matmul(a, w.t()), 2.0 * 8192 ** 3)
a2, w2 = r(9216, 5120), r(1280, 5120) * 0.01
case("pm_gemm 9216x1280x5120 (ring kernel)", lambda: ops.gemm(a2, w2), 2.0 * 9216 * 1280 * 5120)
x, wp = r(16 * 72 * 128, 320), r(320, 9 * 320) * 0.02
zb = torch.zeros(320, device="cuda")
case("conv3x3 147456x320x2880 (ring kernel)", lambda: ops.conv3x3(x, wp, zb, 16, 72, 128), 2.0 * 147456 * 320 * 2880)
qkv = r(16, 9216, 960)
case("attention N=9216", lambda: ops.attention(qkv[..., :320], qkv[..., 320:640], qkv[..., 640:], 5), 4.0 * 9216 * 9216 * 64 * 5 * 16)
big = torch.empty(1 << 28, device="cuda", dtype=torch.float32)
case("copy 1 GiB (HBM-bound)", lambda: big[: 1 << 27].copy_(big[1 << 27:]), 0.0)
