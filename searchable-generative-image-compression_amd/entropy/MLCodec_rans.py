"""Drop-in for the reference pybind module entropy.MLCodec_rans (src/cpp/py_rans/py_rans.cpp:261-281):
RansEncoder / RansDecoder with the same methods, numpy in / numpy out, but the coding itself runs in
the HIP kernels of csrc/entropy.hip (one workgroup per stream).  This per-image facade exists for
drop-in parity; the codec's batched path calls the same kernels on device-resident tensors directly
(sgic_amd.bottleneck) and never touches numpy.

Differences from the reference, all deliberate:
  * streamPart must be 1 and multiThread is ignored (the codec uses ec_thread=False, stream_part=1,
    models/sq_bottleneck.py:57) -> ValueError otherwise;
  * flush() on an empty / all-skipped symbol list returns 01 00 00 80 00 instead of corrupting the heap
    (rans.cpp:165-168), and decode over-reads are bounds-checked (RuntimeError) instead of silent.
"""
import ctypes as C

import numpy as np
import torch

from .._lib import call, check, lib, require_gpu


class _Tables:
    def __init__(self):
        self.handles = []

    def add(self, cdfs, cdfs_sizes, offsets):
        cdfs = np.ascontiguousarray(cdfs, dtype=np.int32)
        sizes = np.ascontiguousarray(cdfs_sizes, dtype=np.int32).reshape(-1)
        offs = np.ascontiguousarray(offsets, dtype=np.int32).reshape(-1)
        assert cdfs.ndim == 2 and cdfs.shape[0] == len(sizes) == len(offs)
        h = C.c_void_p()
        check(lib.sgic_cdf_table_create(cdfs.ctypes.data_as(C.c_void_p), C.c_int(cdfs.shape[0]), C.c_int(cdfs.shape[1]),
                                        sizes.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p),
                                        C.byref(h)), "cdf_table_create")
        self.handles.append(h)
        return len(self.handles) - 1

    def clear(self):
        for h in self.handles:
            lib.sgic_cdf_table_destroy(h)
        self.handles = []

    def __del__(self):
        try:
            self.clear()
        except Exception:
            pass


class RansEncoder:
    def __init__(self, multiThread=False, streamPart=1):
        if streamPart != 1:
            raise ValueError("sgic RansEncoder supports streamPart == 1 only (what the codec uses)")
        require_gpu()
        self._t = _Tables()
        self._sym, self._idx, self._group = [], [], None
        self._stream = np.zeros(0, dtype=np.uint8)

    def add_cdf(self, cdfs, cdfs_sizes, offsets):
        return self._t.add(cdfs, cdfs_sizes, offsets)

    def empty_cdf_buffer(self):
        self._t.clear()

    def reset(self):
        self._sym, self._idx, self._group = [], [], None

    def encode_with_indexes(self, symbols, indexes, cdf_group_index):
        s = np.ascontiguousarray(symbols, dtype=np.int16).reshape(-1)
        i = np.ascontiguousarray(indexes, dtype=np.int16).reshape(-1)
        if len(s) != len(i):
            raise ValueError("symbols and indexes must have the same length")
        if self._group is not None and self._group != cdf_group_index:
            raise ValueError("one cdf group per stream")
        self._t.handles[cdf_group_index]  # IndexError like .at()
        self._group = cdf_group_index
        self._sym.append(s)
        self._idx.append(i)

    def flush(self):
        n = int(sum(len(a) for a in self._sym))
        dev = torch.device("cuda", torch.cuda.current_device())
        if self._group is None:
            self._stream = np.frombuffer(bytes([1, 0, 0, 0x80, 0]), dtype=np.uint8).copy()
            return
        sym = torch.from_numpy(np.concatenate(self._sym) if n else np.zeros(0, np.int16)).to(dev)
        idx = torch.from_numpy(np.concatenate(self._idx) if n else np.zeros(0, np.int16)).to(dev)
        cap = 2 * n + 64
        while True:
            out = torch.empty(cap, dtype=torch.uint8, device=dev)
            meta = torch.zeros(3, dtype=torch.int32, device=dev)
            call("sgic_rans_encode_batch", self._t.handles[self._group], sym, idx, 1, n, out, cap, meta[0:1], meta[1:2],
                 meta[2:3])
            off, ln, err = meta.cpu().tolist()
            if err == -3:  # SGIC_ENOSPC: bypass-heavy stream, retry with the hard upper bound
                cap = 16 * n + 64
                continue
            if err:
                raise IndexError("cdf index out of range")
            self._stream = out[off:off + ln].cpu().numpy()
            return

    def get_encoded_stream(self):
        return self._stream.copy()


class RansDecoder:
    def __init__(self, streamPart=1):
        if streamPart != 1:
            raise ValueError("sgic RansDecoder supports streamPart == 1 only (what the codec uses)")
        require_gpu()
        self._t = _Tables()
        self._dev = None

    def add_cdf(self, cdfs, cdfs_sizes, offsets):
        return self._t.add(cdfs, cdfs_sizes, offsets)

    def empty_cdf_buffer(self):
        self._t.clear()

    def set_stream(self, encoded):
        buf = np.ascontiguousarray(encoded, dtype=np.uint8).reshape(-1)
        dev = torch.device("cuda", torch.cuda.current_device())
        self._buf = torch.from_numpy(buf.copy()).to(dev)
        self._len = torch.tensor([len(buf)], dtype=torch.int32, device=dev)
        self._state = torch.zeros(4, dtype=torch.int32, device=dev)
        self._cap = max(len(buf), 1)
        call("sgic_rans_decode_init_batch", self._buf, self._cap, None, self._len, 1, self._state)

    def decode_stream(self, indexes, cdf_group_index):
        idx = np.ascontiguousarray(indexes, dtype=np.int16).reshape(-1)
        dev = self._buf.device
        d_idx = torch.from_numpy(idx).to(dev)
        out = torch.empty(len(idx), dtype=torch.int16, device=dev)
        call("sgic_rans_decode_batch", self._t.handles[cdf_group_index], self._buf, self._cap, None, self._len, 1,
             self._state, d_idx, len(idx), len(idx), out, len(idx))
        if int(self._state[2].item()) != 0:
            raise RuntimeError("rANS decode ran past the end of the stream / bad index (corrupt bitstream)")
        return out.cpu().numpy()
