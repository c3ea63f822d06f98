"""Print the reference-side ctypes binding stub of INTEGRATION.md section 2 from capi.SIGNATURES (the single
source the product itself binds with), so the documented stub cannot drift from the ABI again.
usage: python tools/gen_binding_stub.py  (tests/test_boundary_cpu.py checks INTEGRATION.md against this output)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pandora_amd import capi  # noqa: E402

NAMES = {ctypes.c_void_p: "c_void_p", ctypes.c_int64: "c_int64", ctypes.c_int: "c_int", ctypes.c_float: "c_float",
         ctypes.c_double: "c_double", ctypes.c_size_t: "c_size_t", ctypes.c_char_p: "c_char_p", None: "None"}


def stub():
    out = ["import ctypes", "from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_void_p",
           'lib = ctypes.CDLL("open-pandora_amd/libpandora_mi355x.so")']
    for name, (res, args) in capi.SIGNATURES.items():
        a = ", ".join(NAMES[t] for t in args)
        out.append(f"lib.{name}.restype, lib.{name}.argtypes = {NAMES[res]}, [{a}]  # {len(args)} arguments")
    out += [
        "",
        "def linear(x, w, bias, out, workspace=None):      # x [M,K], w [N,K] (nn.Linear layout), bf16, on the GPU",
        "    rc = lib.pm_gemm(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), bias.data_ptr(),",
        "                     None, 0,                                       # residual, ldr",
        "                     out.data_ptr(), out.stride(0), x.shape[0], w.shape[0], x.shape[1],",
        "                     0, 0, 2,                                       # act PM_ACT_NONE, flags 0, dtype PM_BF16",
        "                     workspace.data_ptr() if workspace is not None else None,",
        "                     workspace.numel() if workspace is not None else 0,   # split-K scratch (optional)",
        "                     None,                                          # colstats (fused GroupNorm sums): off",
        "                     torch.cuda.current_stream().cuda_stream)       # 19 arguments, as in the header",
        "    if rc: raise RuntimeError(lib.pm_strerror(rc).decode())",
    ]
    return "\n".join(out)


def write_doc():
    """replace the generated block of INTEGRATION.md in place"""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")
    doc = open(path).read()
    head, rest = doc.split("<!-- BEGIN GENERATED STUB -->\n```python\n")
    tail = rest.split("\n```\n<!-- END GENERATED STUB -->")[1]
    open(path, "w").write(head + "<!-- BEGIN GENERATED STUB -->\n```python\n" + stub() + "\n```\n<!-- END GENERATED STUB -->" + tail)


if __name__ == "__main__":
    if "--write" in sys.argv:
        write_doc()
    else:
        print(stub())
