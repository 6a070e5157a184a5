"""FAISS IndexFlatIP on-disk format (what `faiss.write_index(IndexFlatIP)` emits; pinned by the reference's
IO/faiss/index.faiss) and exact inner-product search.  faiss itself is not in the image; the file layout is
"IxFI", i32 d, i64 ntotal, i64 2^20, i64 2^20, u8 is_trained, i32 metric(0 = inner product), u64 count,
fp32 data (compress.py:89-114)."""
import os
import struct

import numpy as np


def write_index_flat_ip(path, vecs):
    v = np.ascontiguousarray(vecs, dtype=np.float32)
    n, d = v.shape
    with open(path, "wb") as f:
        f.write(b"IxFI")
        f.write(struct.pack("<i", d))
        f.write(struct.pack("<q", n))
        f.write(struct.pack("<qq", 1 << 20, 1 << 20))
        f.write(struct.pack("<B", 1))
        f.write(struct.pack("<i", 0))
        f.write(struct.pack("<Q", n * d))
        f.write(v.tobytes())


def read_index_flat_ip(path):
    data = open(path, "rb").read()
    assert data[:4] == b"IxFI", "not an IndexFlatIP file"
    d, = struct.unpack_from("<i", data, 4)
    n, = struct.unpack_from("<q", data, 8)
    cnt, = struct.unpack_from("<Q", data, 37)
    assert cnt == n * d
    return np.frombuffer(data, dtype=np.float32, count=n * d, offset=45).reshape(n, d).copy()


class FaissDB:
    """compress.py:89-114: re-normalise (+1e-12), append, persist index.faiss + ids.txt"""

    def __init__(self, index_dir, dim):
        os.makedirs(index_dir, exist_ok=True)
        self.index_path = os.path.join(index_dir, "index.faiss")
        self.ids_path = os.path.join(index_dir, "ids.txt")
        self._base = read_index_flat_ip(self.index_path) if os.path.exists(self.index_path) else np.zeros((0, dim), np.float32)
        self._new = []          # rows appended since the last merge (a 10 k corpus must not re-copy the matrix per add)
        self.ids = []
        if os.path.exists(self.ids_path):
            with open(self.ids_path, "r", encoding="utf-8") as f:
                self.ids = [ln.strip() for ln in f if ln.strip()]

    def add(self, vec_unit, doc_id):
        v = np.asarray(vec_unit, dtype=np.float32).copy()[None, :]
        v /= np.linalg.norm(v, axis=1, keepdims=True) + 1e-12
        self._new.append(v.astype(np.float32))
        self.ids.append(doc_id)

    @property
    def vecs(self):
        if self._new:
            self._base = np.concatenate([self._base] + self._new, axis=0)
            self._new = []
        return self._base

    def persist(self):
        write_index_flat_ip(self.index_path, self.vecs)
        with open(self.ids_path, "w", encoding="utf-8") as f:
            for i in self.ids:
                f.write(i + "\n")

    def search(self, q, k):
        """exact inner-product top-k (IndexFlatIP.search): -> (scores (nq,k), indices (nq,k))"""
        q = np.atleast_2d(np.asarray(q, dtype=np.float32))
        s = q @ self.vecs.T
        k = min(k, self.vecs.shape[0])
        idx = np.argsort(-s, axis=1, kind="stable")[:, :k]
        return np.take_along_axis(s, idx, axis=1), idx
