"""Drop-in for the reference pybind module entropy.MLCodec_CXX (src/cpp/ops/ops.cpp:84-91)."""
import ctypes as C

import numpy as np

from .._lib import check, lib


def pmf_to_quantized_cdf(pmf, precision=16):
    """list[float] -> list[int] of len(pmf)+1, same steal-from-smallest rule (ops.cpp:24-82)."""
    p = np.ascontiguousarray(np.asarray(pmf, dtype=np.float32))
    out = np.zeros(len(p) + 1, dtype=np.uint32)
    check(lib.sgic_pmf_to_quantized_cdf(p.ctypes.data_as(C.c_void_p), C.c_int(len(p)), C.c_int(int(precision)),
                                        out.ctypes.data_as(C.c_void_p)), "pmf_to_quantized_cdf")
    return out.tolist()
