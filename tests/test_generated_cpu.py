"""The committed assembly of the two 256x256 GEMM kernels is what the generators print (csrc/*.inc are generated files:
tools/gen_wide_loop.py, tools/gen_wide_stream.py) - an edit of one without the other fails here, on the CPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("gen,inc", [("gen_wide_loop.py", "gemm_wide_loop.inc"), ("gen_wide_stream.py", "gemm_wide_stream.inc")])
def test_generated_assembly_is_current(gen, inc):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", gen)], capture_output=True, text=True, check=True).stdout
    with open(os.path.join(ROOT, "open-pandora_amd", "csrc", inc)) as f:
        assert f.read() == out, f"{inc} is stale: python tools/{gen} > open-pandora_amd/csrc/{inc}"
