"""Minimal ctypes binding of the system libzstd for the 512-byte CLIP code (compress.py:66,78 uses
python-zstandard's ZstdCompressor(level=19); that package is not in the image, the C library is).
Host-side, third-party codec: the compressed BYTES depend on the libzstd version, the
decoded payload does not."""
import ctypes as C
import ctypes.util

_lib = None


def _load():
    global _lib
    if _lib is None:
        name = ctypes.util.find_library("zstd") or "libzstd.so.1"
        L = C.CDLL(name)
        L.ZSTD_compressBound.restype = C.c_size_t
        L.ZSTD_compressBound.argtypes = [C.c_size_t]
        L.ZSTD_compress.restype = C.c_size_t
        L.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
        L.ZSTD_decompress.restype = C.c_size_t
        L.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.ZSTD_isError.restype = C.c_uint
        L.ZSTD_isError.argtypes = [C.c_size_t]
        L.ZSTD_getFrameContentSize.restype = C.c_ulonglong
        L.ZSTD_getFrameContentSize.argtypes = [C.c_void_p, C.c_size_t]
        L.ZSTD_createCCtx.restype = C.c_void_p
        L.ZSTD_compressCCtx.restype = C.c_size_t
        L.ZSTD_compressCCtx.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
        _lib = L
    return _lib


class Compressor:
    def __init__(self, level=19):
        self.level = level
        self.L = _load()
        self.ctx = C.c_void_p(self.L.ZSTD_createCCtx())

    def compress(self, data: bytes) -> bytes:
        cap = self.L.ZSTD_compressBound(len(data))
        buf = C.create_string_buffer(cap)
        n = self.L.ZSTD_compressCCtx(self.ctx, buf, cap, data, len(data), self.level)
        if self.L.ZSTD_isError(n):
            raise RuntimeError("zstd compress failed")
        return buf.raw[:n]


def decompress(data: bytes) -> bytes:
    L = _load()
    size = L.ZSTD_getFrameContentSize(data, len(data))
    if size >= (1 << 62):
        size = 1 << 20
    buf = C.create_string_buffer(int(size) or 1)
    n = L.ZSTD_decompress(buf, int(size), data, len(data))
    if L.ZSTD_isError(n):
        raise RuntimeError("zstd decompress failed")
    return buf.raw[:n]
