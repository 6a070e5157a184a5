#!/usr/bin/env python3
"""Counterpart of the reference src/search.py: query a FAISS IndexFlatIP built by compress.py.
`query-c2df` needs no CLIP model (zstd-decode the embedded u8 code, q/255*2-1, l2-normalise; search.py:20-41);
`query-image` / `query-text` run the MI355X CLIP image / text towers (clip.py).  The BPE tokenizer is open_clip's
(third-party, its vocabulary file is not in the reference tree): `query-text` uses `open_clip.get_tokenizer` when the
package is importable and otherwise takes ready-made ids via --token_ids.  The search itself is exact inner product: one fp32 MFMA GEMM (queries x
database^T) + a top-k kernel on the GPU."""
import argparse
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch


class NotSearchable(ValueError):
    """a .c2df whose embedded CLIP code is absent or inconsistent with its own metadata"""


def unit_rows(x, floor=1e-9):
    """rows scaled to unit length; a zero row stays zero instead of dividing by zero"""
    x = np.asarray(x, dtype=np.float32)
    length = np.sqrt(np.sum(x * x, axis=-1, keepdims=True))
    return x / np.maximum(length, floor)


def codes_to_unit(codes_u8):
    """inverse of the compress side's u8 quantiser (compress.py:77, `round((z*0.5+0.5)*255)`): code c -> c/255*2-1 in the
    reference's operation order (search.py:21: divide, then scale, then shift -- `c * (2/255) - 1` differs by one fp32 ulp for 111
    of the 256 codes, enough to reorder near-ties), then back onto the unit sphere (the quantiser moved the vector off it by up
    to half a step per coordinate)"""
    return unit_rows((np.asarray(codes_u8).astype(np.float32) / np.float32(255.0)) * np.float32(2.0) - np.float32(1.0))


def embedded_clip_vector(path):
    """query-c2df (search.py:20-41): the CLIP vector a .c2df carries -- entry `clip_stream` is a zstd frame of `dim` u8
    codes, `clip_meta["dim"]` says how many.  -> (unit vector (dim,) fp32, container header)"""
    from .filemaker import unpack_c2df
    from .zstd import decompress
    entries, header = unpack_c2df(path)
    path = path if isinstance(path, (str, os.PathLike)) else "<c2df bytes>"    # unpack_c2df also takes the file's bytes
    stream, meta = entries.get("clip_stream"), entries.get("clip_meta")
    if stream is None or meta is None:
        raise NotSearchable(f"{path}: container has no embedded CLIP code (entries clip_stream / clip_meta), it cannot be used as a query")
    want = int((meta or {}).get("dim", 0) or 0)
    if want <= 0:
        raise NotSearchable(f"{path}: clip_meta carries no positive 'dim'")
    codes = np.frombuffer(decompress(stream), dtype=np.uint8)
    if codes.size != want:
        raise NotSearchable(f"{path}: clip_stream decodes to {codes.size} codes but clip_meta.dim says {want}")
    return codes_to_unit(codes), header


# names of the reference script (search.py:16-41), kept so that code written against it keeps importing
l2n = unit_rows
dequantize_clip_u8 = codes_to_unit
decode_clip_from_c2df = embedded_clip_vector

# the two on-disk layouts the reference's tools produce: build.py writes (faiss.index, paths.json), compress.py writes
# (index.faiss, ids.txt) -- search.py:65-88 accepts either
_INDEX_LAYOUTS = (("faiss.index", "paths.json", lambda t: list(json.loads(t))),
                  ("index.faiss", "ids.txt", lambda t: [ln.strip() for ln in t.splitlines() if ln.strip()]))


def load_index(index_dir):
    """-> (database (n, d) fp32, doc ids [n]) from whichever layout is present in index_dir"""
    from .faiss_io import read_index_flat_ip
    root = Path(index_dir)
    for index_name, ids_name, parse in _INDEX_LAYOUTS:
        fi, fp = root / index_name, root / ids_name
        if fi.exists() and fp.exists():
            vecs, ids = read_index_flat_ip(str(fi)), parse(fp.read_text(encoding="utf-8"))
            if len(ids) != vecs.shape[0]:
                raise ValueError(f"{root}: {index_name} holds {vecs.shape[0]} vectors but {ids_name} lists {len(ids)} ids")
            return vecs, ids
    raise FileNotFoundError(f"no FAISS index in {root}: expected " + " or ".join(f"{a} + {b}" for a, b, _ in _INDEX_LAYOUTS))


def tokenize(text, ctx=77, token_ids=None):
    """search.py:93-94 `tokenizer([query])` -> (1, ctx) int64.  open_clip's tokenizer when available; `token_ids`
    (already BPE-encoded, <start> ... <end>) is the offline route."""
    if token_ids is not None:
        ids = [int(t) for t in (token_ids.split(",") if isinstance(token_ids, str) else token_ids)]
        if not 0 < len(ids) <= ctx:
            raise ValueError(f"need 1..{ctx} token ids, got {len(ids)}")
        out = np.zeros((1, ctx), dtype=np.int64)
        out[0, :len(ids)] = ids
        return out
    try:
        import open_clip
    except ImportError as e:
        raise RuntimeError("query-text needs open_clip's BPE tokenizer (pip package open_clip_torch) or --token_ids") from e
    return np.asarray(open_clip.get_tokenizer("ViT-B-32")([text]), dtype=np.int64)


def encode_text(tokens, text_model):
    """search.py:92-97 encode_text: unit-norm (1, D) fp32 on the host"""
    return text_model.encode_text(tokens).cpu().numpy().astype("float32")


def search_gpu(q, vecs, topk, device="cuda:0"):
    """exact IndexFlatIP.search on the GPU: -> (scores (nq,k), ids (nq,k))"""
    from . import ops
    dev = torch.device(device)
    k = max(1, min(topk, vecs.shape[0]))
    dq = torch.from_numpy(np.ascontiguousarray(q, dtype=np.float32)).to(dev)
    db = torch.from_numpy(np.ascontiguousarray(vecs, dtype=np.float32)).to(dev)
    scores = ops.gemm(dq, db, w_const=False)
    s, i = ops.topk_rows(scores, k)
    return s.cpu().numpy(), i.cpu().numpy()


def do_search(q, vecs, paths, topk=10):
    sim, ids = search_gpu(q, vecs, topk)
    return [(paths[i], float(sim[0, j])) for j, i in enumerate(ids[0]) if i != -1]


def main(argv=None):
    ap = argparse.ArgumentParser(description="query-text / query-image / query-c2df")
    sub = ap.add_subparsers(dest="cmd", required=True)
    for name, arg in (("query-text", "--text"), ("query-image", "--image"), ("query-c2df", "--c2df")):
        p = sub.add_parser(name)
        p.add_argument("--index_dir", type=Path, required=True)
        p.add_argument(arg, type=str, required=True)
        p.add_argument("--topk", type=int, default=10)
        p.add_argument("--clip_ckpt", type=str, default=None)
        if name == "query-text":
            p.add_argument("--token_ids", type=str, default=None, help="comma-separated BPE ids (offline tokenizer bypass)")
    args = ap.parse_args(argv)
    vecs, paths = load_index(args.index_dir)
    if args.cmd == "query-c2df":
        q = decode_clip_from_c2df(args.c2df)[0][None, :]
    elif args.cmd == "query-image":
        from . import weights as W
        from .codec import ClipCodec
        from .compress import load_image, load_state
        from .config import CLIP_B32
        csd = load_state(args.clip_ckpt, W.clip_spec, CLIP_B32, 4321)
        q = ClipCodec(csd, CLIP_B32, "cuda:0").image_to_unit_vec(load_image(args.image))[None, :]
    else:
        from . import weights as W
        from .clip import ClipTextHIP
        from .compress import load_state
        from .config import CLIP_B32
        toks = tokenize(args.text, CLIP_B32.ctx, args.token_ids)
        tsd = load_state(args.clip_ckpt, W.clip_text_spec, CLIP_B32, 4321)
        q = encode_text(toks, ClipTextHIP(tsd, CLIP_B32, "cuda:0"))
    print(json.dumps([{"path": p, "score": s} for p, s in do_search(q, vecs, paths, args.topk)], ensure_ascii=False, indent=2))
    return 0


if __name__ == "__main__":
    sys.exit(main())
