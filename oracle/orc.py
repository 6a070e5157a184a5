"""ctypes binding of oracle/liboracle.so (the plain-C CPU restatement).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    src = os.path.join(_HERE, "sgic_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_rans_encode.restype = C.c_long
        L.orc_pack12.restype = C.c_size_t
        L.orc_pack12_size.restype = C.c_size_t
        L.orc_pack12_size.argtypes = [C.c_size_t]
        L.orc_torchac_uniform_encode.restype = C.c_size_t
        L.orc_dec_sizeof.restype = C.c_size_t
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def pmf_to_quantized_cdf(pmf, precision=16):
    pmf = np.ascontiguousarray(pmf, dtype=np.float32)
    out = np.zeros(len(pmf) + 1, dtype=np.uint32)
    rc = lib().orc_pmf_to_quantized_cdf(_p(pmf, C.c_float), C.c_int(len(pmf)), C.c_int(precision), _p(out, C.c_uint32))
    if rc:
        raise ValueError(f"pmf_to_quantized_cdf rc={rc}")
    return out


class Table:
    def __init__(self, cdf, sizes, offsets):
        self.cdf = np.ascontiguousarray(cdf, dtype=np.int32)
        self.sizes = np.ascontiguousarray(sizes, dtype=np.int32)
        self.offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        self.rows, self.cols = self.cdf.shape

    def args(self):
        return (_p(self.cdf, C.c_int32), C.c_int(self.rows), C.c_int(self.cols), _p(self.sizes, C.c_int32),
                _p(self.offsets, C.c_int32))


def rans_encode(sym, idx, table):
    sym = np.ascontiguousarray(sym, dtype=np.int16).reshape(-1)
    idx = np.ascontiguousarray(idx, dtype=np.int16).reshape(-1)
    cap = len(sym) * 14 + 64
    out = np.zeros(cap, dtype=np.uint8)
    n = lib().orc_rans_encode(_p(sym, C.c_int16), _p(idx, C.c_int16), C.c_size_t(len(sym)), *table.args(),
                              _p(out, C.c_uint8), C.c_size_t(cap))
    if n < 0:
        raise ValueError(f"rans_encode rc={n}")
    return out[:n].tobytes()


class Decoder:
    """Stateful cursor like the reference RansDecoder (set_stream, then successive decode_stream)."""

    def __init__(self, stream, table):
        self.table = table
        self._buf = np.frombuffer(bytes(stream), dtype=np.uint8).copy()
        self._st = C.create_string_buffer(lib().orc_dec_sizeof())
        rc = lib().orc_rans_dec_init(self._st, _p(self._buf, C.c_uint8), C.c_size_t(len(self._buf)))
        if rc:
            raise ValueError(f"dec_init rc={rc}")

    def decode(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int16).reshape(-1)
        out = np.zeros(len(idx), dtype=np.int16)
        rc = lib().orc_rans_decode(self._st, _p(idx, C.c_int16), C.c_size_t(len(idx)), *self.table.args(),
                                   _p(out, C.c_int16))
        if rc:
            raise ValueError(f"rans_decode rc={rc}")
        return out


def pack12(idx):
    idx = np.ascontiguousarray(idx, dtype=np.int16).reshape(-1)
    out = np.zeros(lib().orc_pack12_size(len(idx)), dtype=np.uint8)
    n = lib().orc_pack12(_p(idx, C.c_int16), C.c_size_t(len(idx)), _p(out, C.c_uint8))
    return out[:n].tobytes()


def torchac_uniform_encode(idx, lp=4097):
    idx = np.ascontiguousarray(idx, dtype=np.int16).reshape(-1)
    cap = len(idx) * 2 + 16
    out = np.zeros(cap, dtype=np.uint8)
    n = lib().orc_torchac_uniform_encode(_p(idx, C.c_int16), C.c_size_t(len(idx)), C.c_int(lp), _p(out, C.c_uint8),
                                         C.c_size_t(cap))
    return out[:n].tobytes()


def unpack12(data, n):
    buf = np.frombuffer(bytes(data) + b"\0\0", dtype=np.uint8).copy()
    out = np.zeros(n, dtype=np.int16)
    lib().orc_unpack12(_p(buf, C.c_uint8), C.c_size_t(n), _p(out, C.c_int16))
    return out


def quant_step(y, scales, means, k, thr, y_hat_so_far):
    """One image, NCHW (C,H,W) fp32 arrays; returns (sym (C/4,H,W) int16, idx int16); updates y_hat_so_far."""
    y = np.ascontiguousarray(y, dtype=np.float32)
    scales = np.ascontiguousarray(scales, dtype=np.float32)
    means = np.ascontiguousarray(means, dtype=np.float32)
    assert y_hat_so_far.dtype == np.float32 and y_hat_so_far.flags.c_contiguous
    Cc, H, W = y.shape
    sym = np.zeros((Cc // 4, H, W), dtype=np.int16)
    idx = np.zeros((Cc // 4, H, W), dtype=np.int16)
    lib().orc_quant_step(_p(y, C.c_float), _p(scales, C.c_float), _p(means, C.c_float), C.c_int(Cc), C.c_int(H),
                         C.c_int(W), C.c_int(k), C.c_float(-1.0 if thr is None else thr),
                         _p(y_hat_so_far, C.c_float), _p(sym, C.c_int16), _p(idx, C.c_int16))
    return sym, idx


def resize_bicubic_u8(img_chw, oh, ow):
    img = np.ascontiguousarray(img_chw, dtype=np.uint8)
    c, h, w = img.shape
    assert c == 3
    out = np.zeros((3, oh, ow), dtype=np.uint8)
    lib().orc_resize_bicubic_u8(_p(img, C.c_uint8), C.c_int(h), C.c_int(w), _p(out, C.c_uint8), C.c_int(oh), C.c_int(ow))
    return out
