"""CPU: libsgic.so loads and exports every symbol include/sgic.h declares (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "sgic.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sgic_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    so = os.path.join(ROOT, "searchable-generative-image-compression_amd", "libsgic.so")
    lib = ctypes.CDLL(so)
    names = _declared()
    assert len(names) >= 10
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    lib.sgic_version.restype = ctypes.c_int
    assert lib.sgic_version() >= 100


def test_host_table_builder_matches_golden(golden_dir):
    import numpy as np
    import sgic_amd  # noqa
    from sgic_amd.entropy.MLCodec_CXX import pmf_to_quantized_cdf
    k = np.load(os.path.join(golden_dir, "pmf_kats.npz"))
    for i in range(int(k["n"])):
        assert pmf_to_quantized_cdf(k[f"pmf_{i}"].tolist(), 16) == k[f"cdf_{i}"].tolist()
