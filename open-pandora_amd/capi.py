"""ctypes binding of libpandora_mi355x.so (include/pandora_mi355x.h).

The product path has no fallback: if the library is missing or a call fails, this raises.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
DIAG_LIB_PATH = os.path.join(HERE, "libpandora_mi355x_diag.so")  # the -DPM_DIAG build (include/pandora_mi355x_diag.h)
# PANDORA_DIAG_LIB=1: measurement runs (tools/, env A/Bs) load the diagnostics build in place of the shipped library - the
# only build that reads the PANDORA_* kernel-tuning switches; PANDORA_LIB: an explicit path (kernel experiments)
LIB_PATH = os.environ.get("PANDORA_LIB", DIAG_LIB_PATH if os.environ.get("PANDORA_DIAG_LIB") == "1"
                          else os.path.join(HERE, "libpandora_mi355x.so"))

PM_F16, PM_BF16, PM_F32 = 1, 2, 3
PM_OUT_HILO = 0x100
PM_TOTALS_I64, PM_FLAG_STATS_I64 = 0x200, 64
PM_FLAG_A_F32, PM_FLAG_OUT_F32, PM_FLAG_RES_F32, PM_FLAG_BIAS_IS_SCALE, PM_FLAG_A_LO, PM_FLAG_W_WRAP = 1, 2, 4, 8, 16, 32
PM_ACT_NONE, PM_ACT_SILU, PM_ACT_GEGLU, PM_ACT_GELU = 0, 1, 2, 3
ACT_CODES = {"none": PM_ACT_NONE, None: PM_ACT_NONE, "silu": PM_ACT_SILU, "geglu": PM_ACT_GEGLU, "gelu": PM_ACT_GELU}

# name -> (restype, argtypes); mirrors include/pandora_mi355x.h declaration by declaration
SIGNATURES = {
    "pm_strerror": (c_char_p, [c_int]),
    "pm_abi_version": (c_int, []),
    "pm_gemm": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p,
                        c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_void_p, c_size_t,
                        c_void_p, c_void_p]),
    "pm_groupnorm_finalize_colstats": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p]),
    "pm_gemm_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64, c_int]),
    "pm_gemm_colstats_rows": (c_int, [c_int64, c_int64, c_int64, c_int, c_size_t]),
    "pm_gemm_kernel_choice": (c_int, [c_int64, c_int64, c_int64, c_int, c_int, c_size_t]),
    "pm_conv2d_3x3": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                              c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_int,
                              c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "pm_conv_temporal_k3": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int64,
                                    c_int64, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "pm_conv_temporal_k3_clips": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64,
                                          c_int64, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "pm_groupnorm_nchunks": (c_int64, [c_int64, c_int64]),
    "pm_groupnorm_stats": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int64,
                                   c_int, c_int, c_void_p]),
    "pm_groupnorm_apply": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_double,
                                   c_float, c_int, c_int, c_int, c_void_p]),
    "pm_layernorm": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                             c_int64, c_float, c_int, c_int, c_void_p]),
    "pm_split16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_void_p]),
    "pm_split16_upsample2x": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_int, c_int,
                                      c_void_p]),
    "pm_ln_gemm": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_float, c_void_p, c_int64, c_void_p, c_void_p,
                           c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "pm_ln_gemm_supported": (c_int, [c_int64, c_int64, c_int64, c_int]),
    "pm_attention": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64,
                             c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_float,
                             c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_float, c_int,
                             c_void_p]),
    "pm_attention_fp8_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64, c_int64]),
    "pm_attention_fp8": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p,
                                 c_int64, c_int64, c_int64, c_int64, c_int64, c_float, c_int, c_void_p, c_size_t,
                                 c_void_p]),
    "pm_attention_generic": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p,
                                     c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_float, c_int, c_void_p]),
    "pm_attention_temporal": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p,
                                      c_int64, c_int64, c_int64, c_int64, c_int64, c_float, c_int,
                                      c_void_p]),
    "pm_gemv_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                            c_int, c_int, c_int, c_void_p]),
    "pm_ddim_update": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                               c_float, c_float, c_float, c_float, c_float, c_float, c_float, c_int,
                               c_void_p]),
    "pm_timestep_embedding": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "pm_pack_input": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64,
                              c_int, c_void_p]),
    "pm_unpack_output": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p]),
    "pm_latent_affine": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64,
                                 c_float, c_int, c_void_p]),
    "pm_softmax_rows": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_float, c_int,
                                c_void_p]),
    "pm_peer_mailbox_bytes": (c_size_t, [c_int, c_int64, c_int64]),
    "pm_peer_create": (c_int, [c_size_t, c_void_p, c_void_p, c_void_p]),
    "pm_peer_open": (c_int, [c_void_p, c_void_p]),
    "pm_peer_close": (c_int, [c_void_p]),
    "pm_peer_destroy": (c_int, [c_void_p]),
    "pm_peer_status": (c_int, [c_void_p, c_void_p, c_void_p]),
    "pm_peer_exchange": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
                                 c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_double, c_void_p]),
}

# include/pandora_mi355x_diag.h: only the diagnostics build exports these
DIAG_SIGNATURES = {
    "pm_debug_attn_variant": (None, [c_int]),
    "pm_debug_attn_stamps": (None, [c_void_p]),
    "pm_debug_gemm_wide": (None, [c_int]),
    "pm_debug_gemm_wstream": (None, [c_int]),
    "pm_debug_wide_stamps": (None, [c_void_p]),
    "pm_debug_erf": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
}

_lib = None
_diag = None


class PandoraKernelError(RuntimeError):
    pass


def load(path=None):
    """Load the shared library and attach the declared signatures. Raises if it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise PandoraKernelError(
            f"{path} not found: build it with `python open-pandora_amd/build.py` "
            "(there is no CPU or PyTorch fallback for the denoising path)")
    # (a process that will also use torch on the GPU must have imported torch BEFORE this point: torch bundles its own
    # libamdhip64.so.7 and this library has to resolve that SONAME to the copy torch runs on - HipOps and
    # __graft_entry__.build() do; loading the ROCm copy first leaves torch's streams in another runtime instance)
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def load_diag():
    """The diagnostics build (every export of the shipped library + DIAG_SIGNATURES) as a handle of its own, beside the
    shipped library in the same process.  Measurement code only (tools/, bench.py's ceiling leg, variant tests)."""
    global _diag
    if _diag is not None:
        return _diag
    if not os.path.exists(DIAG_LIB_PATH):
        raise PandoraKernelError(f"{DIAG_LIB_PATH} not found: build it with `python open-pandora_amd/build.py --diag`")
    lib = ctypes.CDLL(DIAG_LIB_PATH)
    for name, (res, args) in {**SIGNATURES, **DIAG_SIGNATURES}.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _diag = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().pm_strerror(rc).decode()
        raise PandoraKernelError(f"{what}: {msg} (code {rc})")
