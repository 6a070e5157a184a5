"""`.c2df` v2 container -- byte-identical to the reference's src/filemaker.py:4-173 (magic "C2DF", LE u16
version, JSON header, typed TLV entries; BYTES/STR/JSON/NP payloads carry their length twice: the outer
entry length and the inner one).  Host-side integer work; no kernel involved."""
import json
import struct
from pathlib import Path

import numpy as np

T_BYTES, T_STR, T_INT, T_FLOAT, T_JSON, T_NP, T_NONE, T_BOOL = range(8)
_FIXED = (T_INT, T_FLOAT, T_BOOL, T_NONE)


def _np_payload(arr):
    dt = arr.dtype.str.encode("utf-8")
    data = arr.tobytes(order="C")
    parts = [struct.pack("<B", len(dt)), dt, struct.pack("<B", arr.ndim)]
    parts += [struct.pack("<I", int(d)) for d in arr.shape]
    parts += [struct.pack("<I", len(data)), data]
    return b"".join(parts)


def _dump_entry(key, val):
    """type dispatch in the reference's order (filemaker.py:20-73)"""
    if key in {"z_indeices_shape", "h_indices_shape", "y_shape", "x_shape"} or key.endswith("_shape"):
        return T_NP, _np_payload(np.asarray(val, dtype=np.int32))
    if key in {"token_length", "num_tokens", "n_tokens"} or key.endswith("_length"):
        return T_INT, struct.pack("<q", int(val))
    if val is None:
        return T_NONE, b""
    if isinstance(val, bool):
        return T_BOOL, struct.pack("<B", 1 if val else 0)
    if isinstance(val, int):
        return T_INT, struct.pack("<q", val)
    if isinstance(val, float):
        return T_FLOAT, struct.pack("<d", val)
    if isinstance(val, (bytes, bytearray, memoryview)):
        b = bytes(val)
        return T_BYTES, struct.pack("<I", len(b)) + b
    if isinstance(val, str):
        b = val.encode("utf-8")
        return T_STR, struct.pack("<I", len(b)) + b
    arr = None
    if isinstance(val, np.ndarray):
        arr = val
    elif hasattr(val, "detach") and hasattr(val, "cpu"):
        arr = val.detach().cpu().contiguous().numpy()
    if arr is not None:
        return T_NP, _np_payload(arr)
    if isinstance(val, (list, dict)):
        jb = json.dumps(val, ensure_ascii=False).encode("utf-8")
        return T_JSON, struct.pack("<I", len(jb)) + jb
    s = str(val).encode("utf-8")
    return T_STR, struct.pack("<I", len(s)) + s


def pack_c2df(enc_result: dict, header: dict) -> bytes:
    out = [b"C2DF", struct.pack("<H", int(header.get("version", 2)))]
    hb = json.dumps(header, ensure_ascii=False).encode("utf-8")
    out += [struct.pack("<I", len(hb)), hb, struct.pack("<I", len(enc_result))]
    for k, v in enc_result.items():
        kb = k.encode("utf-8")
        t, payload = _dump_entry(k, v)
        out += [struct.pack("<H", len(kb)), kb, struct.pack("<B", t)]
        if t not in _FIXED:
            out.append(struct.pack("<I", len(payload)))
        out.append(payload)
    return b"".join(out)


def _load_entry(t, payload):
    if t == T_NONE:
        return None
    if t == T_BOOL:
        return bool(payload[0])
    if t == T_INT:
        return struct.unpack_from("<q", payload, 0)[0]
    if t == T_FLOAT:
        return struct.unpack_from("<d", payload, 0)[0]
    if t in (T_BYTES, T_STR, T_JSON):
        n, = struct.unpack_from("<I", payload, 0)
        b = payload[4:4 + n]
        return b if t == T_BYTES else (b.decode("utf-8") if t == T_STR else json.loads(b.decode("utf-8")))
    if t == T_NP:
        off = 0
        dl = payload[off]
        off += 1
        dt = payload[off:off + dl].decode("utf-8")
        off += dl
        nd = payload[off]
        off += 1
        shape = []
        for _ in range(nd):
            shape.append(struct.unpack_from("<I", payload, off)[0])
            off += 4
        n, = struct.unpack_from("<I", payload, off)
        off += 4
        return np.frombuffer(payload[off:off + n], dtype=np.dtype(dt)).reshape(shape)
    raise ValueError(f"unknown type code: {t}")


def unpack_c2df(src):
    data = Path(src).read_bytes() if isinstance(src, (str, Path)) else bytes(src)
    assert data[:4] == b"C2DF", "bad magic"
    off = 6
    hlen, = struct.unpack_from("<I", data, off)
    off += 4
    header = json.loads(data[off:off + hlen].decode("utf-8")) if hlen > 0 else {}
    off += hlen
    n, = struct.unpack_from("<I", data, off)
    off += 4
    enc = {}
    for _ in range(n):
        kl, = struct.unpack_from("<H", data, off)
        off += 2
        key = data[off:off + kl].decode("utf-8")
        off += kl
        t = data[off]
        off += 1
        if t in _FIXED:
            ln = {T_INT: 8, T_FLOAT: 8, T_BOOL: 1, T_NONE: 0}[t]
        else:
            ln, = struct.unpack_from("<I", data, off)
            off += 4
        enc[key] = _load_entry(t, data[off:off + ln])
        off += ln
    return enc, header
