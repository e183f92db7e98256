"""ctypes binding of the C ABI in include/occnet_amd.h.

The product path has no CPU fallback: if libocc_amd.so is missing or an entry point is absent this
module raises at import/use time.  torch is imported first so the HIP runtime already loaded by
PyTorch-ROCm is the one our library binds to (same process, same streams).
"""
import ctypes
import os
import re

import torch  # noqa: F401  (loads libamdhip64 before our library)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libocc_amd.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "occnet_amd.h")


class OccAmdError(RuntimeError):
    """Raised when a C-ABI entry point reports failure (mirror of TORCH_CHECK exceptions)."""


class OccAmdUnsupported(OccAmdError):
    """The shape has no fused kernel; the caller must take the unfused HIP path."""


def declared_symbols(header=HEADER_PATH):
    """Names of every function declared in the public header."""
    with open(header) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(occ_[a-z0-9_]+)\s*\(", src)))


_lib = None
ABI = 3     # include/occnet_amd.h: bumped whenever a signature changes (2: range scales of the fp16 value rows; 3: their weight terms in device memory)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OccAmdError(
                f"{LIB_PATH} not found: build it with `python -m occnet_amd.build` "
                "(the MI355X path has no CPU fallback)")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.occ_last_error.restype = ctypes.c_char_p
        _lib.occ_abi_version.restype = ctypes.c_int
        if _lib.occ_abi_version() != ABI:
            got, _lib = _lib.occ_abi_version(), None
            raise OccAmdError(f"{LIB_PATH} has C ABI {got}, this package binds ABI {ABI}: rebuild it "
                              "(`python -m occnet_amd.build`)")
    return _lib


def check(rc, what):
    if rc == 0:
        return
    msg = lib().occ_last_error().decode()
    if rc == -3:
        raise OccAmdUnsupported(f"{what}: {msg}")
    raise OccAmdError(f"{what} failed (code {rc}): {msg}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def stream_ptr(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


i32 = ctypes.c_int
i64 = ctypes.c_int64
f32 = ctypes.c_float
