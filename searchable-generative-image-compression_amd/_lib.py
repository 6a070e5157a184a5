"""ctypes binding of libsgic.so (C ABI in include/sgic.h).  Fails loudly: no library -> ImportError,
no GPU -> RuntimeError at the first op.  There is deliberately no CPU fallback."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("SGIC_LIB") or os.path.join(_HERE, "libsgic.so")   # SGIC_LIB: A/B builds of the library (tools)

if not os.path.exists(_SO):
    raise ImportError(f"{_SO} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(hipcc --offload-arch=gfx950). The product path has no CPU fallback.")

lib = C.CDLL(_SO)
lib.sgic_last_error.restype = C.c_char_p
lib.sgic_pack12_size.restype = C.c_size_t
lib.sgic_pack12_size.argtypes = [C.c_size_t]

c_void_p, c_int, c_float, c_size_t = C.c_void_p, C.c_int, C.c_float, C.c_size_t


class LaunchOpts(C.Structure):
    """sgic_launch_opts (include/sgic.h): per-call launch options of the GEMM-family / attention entry points"""
    _fields_ = [("tile_mode", C.c_int), ("attn_mode", C.c_int), ("profiler", C.c_void_p), ("w_packed", C.c_int), ("a_packed", C.c_int)]


def launch_opts(tile_mode=0, attn_mode=0, profiler=None, w_packed=0, a_packed=0):
    """-> a by-reference ctypes argument, or NULL when every field is at its default"""
    if not tile_mode and not attn_mode and not profiler and not w_packed and not a_packed:
        return c_void_p(0)
    return C.byref(LaunchOpts(int(tile_mode), int(attn_mode), profiler, int(w_packed), int(a_packed)))


class SgicError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        raise SgicError(f"libsgic {what} failed rc={rc}: {lib.sgic_last_error().decode()}")


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("sgic_amd needs an MI355X (gfx950) GPU: the hot path is HIP-only, no CPU fallback")


try:   # the raw HIP stream of torch's current stream without building a Stream object (8 us -> 0.3 us per launch)
    _raw_stream = torch._C._cuda_getCurrentRawStream
    _get_device = torch._C._cuda_getDevice
except AttributeError:   # a torch without the private hooks
    _raw_stream = None


def stream():
    """HIP stream of torch's CURRENT stream on the current device (honours `with torch.cuda.stream(...)`)"""
    if _raw_stream is not None:
        return c_void_p(_raw_stream(_get_device()))
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """device (or host) pointer of a contiguous tensor / numpy array; None -> NULL"""
    if t is None:
        return c_void_p(0)
    if isinstance(t, torch.Tensor):
        assert t.is_contiguous(), "sgic kernels take contiguous tensors"
        return c_void_p(t.data_ptr())
    return c_void_p(t.ctypes.data)


_NO_STREAM = {"sgic_pmf_to_quantized_cdf", "sgic_cdf_table_create", "sgic_profiler_create", "sgic_profiler_begin", "sgic_profiler_end"}


def call(name, *args):
    """call lib.<name>(*args, current HIP stream) -> check return code.  Every device entry point of
    the C ABI takes the stream as its LAST argument; it is appended here."""
    fn = getattr(lib, name)
    conv = []
    for a in args:
        if isinstance(a, (torch.Tensor,)) or a is None or hasattr(a, "ctypes"):
            conv.append(ptr(a))
        elif isinstance(a, bool):
            conv.append(c_int(int(a)))
        elif isinstance(a, int):
            conv.append(c_int(a))
        elif isinstance(a, float):
            conv.append(c_float(a))
        else:
            conv.append(a)
    if name not in _NO_STREAM:
        conv.append(stream())
    check(fn(*conv), name)
