"""CPU: the C-ABI library builds/loads and exports every symbol include/*.h declares (no compute)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="pandora_mi355x.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from open_pandora_amd import build, capi
    path = build.build()
    lib = ctypes.CDLL(path)
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/pandora_mi355x.h but not exported"
    assert set(capi.SIGNATURES) == set(syms)
    loaded = capi.load()
    assert loaded.pm_abi_version() == 1
    assert loaded.pm_strerror(-2).decode().startswith("shape")


def test_shipped_library_has_no_diagnostics_and_reads_no_environment():
    """kernel-variant overrides, probe instantiations and the PANDORA_* tuning switches live in the
    -DPM_DIAG build only; the shipped library exports none of them and never calls getenv."""
    import subprocess
    from open_pandora_amd import build, capi
    lib = ctypes.CDLL(build.build())
    diag_syms = declared_symbols("pandora_mi355x_diag.h")
    assert diag_syms == sorted(capi.DIAG_SIGNATURES) and diag_syms
    for s in diag_syms:
        assert not hasattr(lib, s), f"{s} (diagnostics) is exported by the shipped library"
    und = subprocess.run(["nm", "-D", "--undefined-only", build.LIB], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in und, "the shipped library reads the environment"
    dlib = ctypes.CDLL(build.build(diag=True))
    for s in declared_symbols() + diag_syms:
        assert hasattr(dlib, s), f"{s} missing from the diagnostics build"
    und = subprocess.run(["nm", "-D", "--undefined-only", build.LIB_DIAG], capture_output=True, text=True, check=True).stdout
    assert "getenv" in und


def test_product_refuses_to_run_without_ops():
    import pytest
    import torch
    from open_pandora_amd.unet import UNetModel
    m = UNetModel(in_channels=8, model_channels=64, out_channels=4, num_res_blocks=1, attention_resolutions=[1],
                  channel_mult=[1], num_head_channels=64, context_dim=1024, use_linear=True,
                  use_relative_position=False, temporal_length=16)
    # inference (eval / no_grad, as WorldModel.generate runs it) without an op table: loud, no eager fallback.  (In
    # training mode with autograd on the module takes the differentiable path instead: tests/test_boundary_cpu.py.)
    with pytest.raises(RuntimeError, match="bind"):
        m.eval()(torch.zeros(1, 8, 16, 8, 8), torch.tensor([1]), context=torch.zeros(1, 333, 1024))
    with torch.no_grad(), pytest.raises(RuntimeError, match="bind"):
        m.train()(torch.zeros(1, 8, 16, 8, 8), torch.tensor([1]), context=torch.zeros(1, 333, 1024))


def test_hip_ops_fail_loudly_on_cpu():
    import pytest
    import torch
    from open_pandora_amd.ops_hip import HipOps
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(Exception):
        HipOps(torch.float16, "cuda:0")
