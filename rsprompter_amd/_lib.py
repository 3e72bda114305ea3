"""ctypes binding of librsp_hip.so (the C ABI declared in include/rsp_hip.h).

The product path has NO fallback: if the library is missing, or a call returns
a non-zero status, a RuntimeError is raised.  Nothing here imports `oracle/`.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librsp_hip.so")

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_int64 = ctypes.c_int64
c_float = ctypes.c_float


class RspGemmDesc(ctypes.Structure):
    _fields_ = [
        ("A", c_void_p), ("Bhi", c_void_p), ("Blo", c_void_p), ("C", c_void_p),
        ("bias", c_void_p), ("res", c_void_p), ("a_rowmap", c_void_p), ("c_rowmap", c_void_p),
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("lda", c_int), ("ldc", c_int), ("ldr", c_int),
        ("res_mod", c_int), ("act", c_int),
        ("alpha", c_float), ("a_scale_log2", c_int),
        ("conv_k", c_int), ("conv_stride", c_int), ("conv_pad", c_int),
        ("conv_H", c_int), ("conv_W", c_int), ("conv_C", c_int),
        ("conv_Ho", c_int), ("conv_Wo", c_int),
    ]


# name -> (restype, argtypes); the CPU test checks every symbol is exported.
PROTOTYPES = {
    "rsp_abi_version": (c_int, []),
    "rsp_build_info": (ctypes.c_char_p, []),
    "rsp_split_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "rsp_gemm": (c_int, [ctypes.POINTER(RspGemmDesc), c_void_p]),
    "rsp_layernorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_int, c_void_p]),
    "rsp_vit_relpos": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "rsp_vit_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "rsp_preprocess": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                               ctypes.POINTER(c_float), ctypes.POINTER(c_float), c_int, c_float, c_void_p]),
    "rsp_patchify": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
}

_lib = None


def load():
    """Load librsp_hip.so (after torch, so both share one HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  -- makes libamdhip64.so.7 resident first
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m rsprompter_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no fallback path.")
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        raise RuntimeError(f"librsp_hip: {what} failed with status {status}")
