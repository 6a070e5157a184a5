#!/usr/bin/env python3
"""Counterpart of the reference src/search.py: query a FAISS IndexFlatIP built by compress.py.
`query-c2df` needs no CLIP model (zstd-decode the embedded u8 code, q/255*2-1, l2-normalise; search.py:20-41);
`query-image` / `query-text` run the MI355X CLIP image / text towers (clip.py).  The BPE tokenizer is open_clip's
(third-party, its vocabulary file is not in the reference tree): `query-text` uses `open_clip.get_tokenizer` when the
package is importable and otherwise takes ready-made ids via --token_ids.  The search itself is exact inner product: one fp32 MFMA GEMM (queries x
database^T) + a top-k kernel on the GPU."""
import argparse
import json
import sys
from pathlib import Path

import numpy as np
import torch


def l2n(x, axis=-1, eps=1e-9):
    n = np.linalg.norm(x, axis=axis, keepdims=True)
    return x / np.maximum(n, eps)


def dequantize_clip_u8(q):
    z = (q.astype(np.float32) / 255.0) * 2.0 - 1.0
    return l2n(z.astype(np.float32))


def decode_clip_from_c2df(path):
    from .filemaker import unpack_c2df
    from .zstd import decompress
    enc, header = unpack_c2df(path)
    if "clip_stream" not in enc or "clip_meta" not in enc:
        raise ValueError(f"{path} No 'clip_stream' or 'clip_meta' was found, this file can't be used to search!")
    dim = int((enc["clip_meta"] or {}).get("dim", 0))
    if dim <= 0:
        raise ValueError(f"{path} Invalid clip_meta.dim")
    q = np.frombuffer(decompress(enc["clip_stream"]), dtype=np.uint8)
    if q.size != dim:
        raise ValueError(f"{path} Dimension didn't match: q={q.size}, dim={dim}")
    return dequantize_clip_u8(q).astype("float32"), header


def load_index(index_dir):
    from .faiss_io import read_index_flat_ip
    index_dir = Path(index_dir)
    if (index_dir / "faiss.index").exists() and (index_dir / "paths.json").exists():
        vecs = read_index_flat_ip(str(index_dir / "faiss.index"))
        paths = json.loads((index_dir / "paths.json").read_text(encoding="utf-8"))
    elif (index_dir / "index.faiss").exists() and (index_dir / "ids.txt").exists():
        vecs = read_index_flat_ip(str(index_dir / "index.faiss"))
        paths = [ln.strip() for ln in (index_dir / "ids.txt").read_text(encoding="utf-8").splitlines() if ln.strip()]
    else:
        raise FileNotFoundError(f"Can't find FAISS index in {index_dir}")
    return vecs, paths


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
    scores = ops.gemm(dq, db)
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
