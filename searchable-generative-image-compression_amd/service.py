"""Resident in-process service: the endpoints of the reference's webapp.py (`/compress`, `/decompress`,
`/search/stream/{text,image,c2df}`, `/file`, previews; webapp.py:63-325) served by ONE long-lived object that owns the
model, instead of a `subprocess.run(["python", "./src/compress.py", ...])` per request that pays the model build
(45 s on the reference) every time (webapp.py:127-137,182-192,242,272,303).

`ResidentService` is transport-free: every endpoint is a method that takes the request's payload (file name + bytes, or
the JSON body) and returns a `Response` (status, body bytes, media type, file name, headers) or, for the streaming search
endpoints, an iterator of NDJSON lines -- exactly the bodies / headers / status codes the reference app produces, so a
FastAPI / Starlette / http.server adapter is a few lines per route (`http_handler()` below is the stdlib one).  The
tests drive the methods directly: no network stack involved.

Single launching thread by design (one process per GPU): a lock serialises the GPU sections; requests for images of
equal geometry could be micro-batched on top of this object (encode_batch / decode_batch take B > 1) -- not done here.
"""
import datetime
import hashlib
import io
import json
import os
import threading
import time
import zipfile
from pathlib import Path
from urllib.parse import quote

import numpy as np
import torch

IMAGE_EXTS = {".png", ".jpg", ".jpeg", ".webp", ".bmp"}
_MEDIA = {".png": "image/png", ".jpg": "image/jpeg", ".jpeg": "image/jpeg", ".webp": "image/webp", ".bmp": "image/bmp"}


class ServiceError(Exception):
    """what the reference raises as fastapi.HTTPException(status_code, detail)"""

    def __init__(self, status, detail):
        super().__init__(f"{status}: {detail}")
        self.status, self.detail = int(status), detail


class Response:
    def __init__(self, body, media_type, filename=None, headers=None, status=200):
        self.status, self.body, self.media_type, self.filename, self.headers = status, body, media_type, filename, dict(headers or {})


def timing_headers(elapsed_ms, stage):
    """webapp.py:40-47"""
    return {"X-SIC-Stage": stage, "X-SIC-Elapsed-MS": str(int(elapsed_ms)), "X-SIC-Elapsed-S": f"{elapsed_ms / 1000:.3f}",
            "X-SIC-Server-Clock": datetime.datetime.now(datetime.timezone.utc).replace(tzinfo=None).isoformat() + "Z",
            "Access-Control-Expose-Headers": "X-SIC-Stage, X-SIC-Elapsed-MS, X-SIC-Elapsed-S, X-SIC-Server-Clock, "
                                             "Content-Disposition, Content-Type"}


def media_type_of(path):
    return _MEDIA.get(Path(path).suffix.lower(), "application/octet-stream")      # .c2df and everything else: octet-stream


def ndjson(obj):
    return (json.dumps(obj, ensure_ascii=False) + "\n").encode("utf-8")


class ResidentService:
    def __init__(self, ckpt_path=None, clip_ckpt=None, device="cuda:0", small=False, index_dir=None, preview_cache=None,
                 media_roots=None):
        from . import weights as W
        from .codec import ClipCodec, Codec
        from .compress import load_state
        from .config import CLIP_B32, CLIP_TINY, LARGE, SMALL
        t0 = time.perf_counter()
        self.device = torch.device(device)
        self.cfg, self.ccfg = (SMALL, CLIP_TINY) if small else (LARGE, CLIP_B32)
        sd = load_state(ckpt_path, W.full_spec, self.cfg, 1234)
        csd = load_state(clip_ckpt, lambda c: W.clip_spec(c) + W.clip_text_spec(c), self.ccfg, 4321)
        self.model = Codec(sd, self.cfg, self.device)                                  # built ONCE, then resident
        self.model.hybrid_codec.quantize_feat.force_zero_thres = 0.12                  # compress.py:238-239
        self.model.hybrid_codec.quantize_feat.update(force=True)
        self.clipc = ClipCodec(csd, self.ccfg, self.device)
        self._csd, self._text = csd, None
        self.index_dir = Path(index_dir or os.getenv("INDEX_DIR", "./IO/faiss")).resolve()
        self.preview_cache = Path(preview_cache or os.getenv("PREVIEW_CACHE", "./cache/previews")).resolve()
        self.preview_cache.mkdir(parents=True, exist_ok=True)
        self.media_roots = [Path(p).resolve() for p in (media_roots or [os.getenv("MEDIA_ROOT", "./"), "./data", "./IO"])] + \
            [self.index_dir, self.index_dir.parent]
        self._lock = threading.Lock()
        self._index = {}                     # index_dir -> (stamp, device matrix, ids)
        self.build_seconds = time.perf_counter() - t0

    # ------------------------------------------------------------------ codec endpoints
    def _encode(self, img_chw):
        """the body of the reference's compress loop for one image (compress.py:252-280) -> .c2df bytes"""
        from . import ops
        from .entropy.compression_model import get_padding_size
        from .filemaker import pack_c2df
        H, W = int(img_chw.shape[1]), int(img_chw.shape[2])
        pad = get_padding_size(H, W, p=256)
        x = img_chw.to(self.device).float().contiguous()[None]
        enc = self.model.encode_only(ops.pad_replicate(x, *pad))
        unit, q = self.clipc.batch_to_codes(x)
        enc["clip_stream"] = self.clipc.compress_codes(q[0].cpu().numpy())
        enc["clip_meta"] = self.clipc.meta(self.ccfg.embed_dim)
        header = {"version": 2, "model_id": enc["clip_meta"]["model_id"], "embed_dim": int(self.ccfg.embed_dim),
                  "quant_type": "u8_symmetric_-1_1", "image_hw": [H, W], "padding": [int(v) for v in pad]}
        return pack_c2df(enc, header), unit[0].cpu().numpy()

    @staticmethod
    def _image_from_bytes(data):
        from PIL import Image
        a = np.array(Image.open(io.BytesIO(data)).convert("RGB"), dtype=np.uint8)
        return torch.from_numpy(a).permute(2, 0, 1).float().div(255.0) * 2.0 - 1.0            # compress.py:160-164

    def compress(self, filename, data):
        """POST /compress (webapp.py:113-165): one uploaded image -> its .c2df as application/octet-stream"""
        t0 = time.perf_counter()
        try:
            img = self._image_from_bytes(data)
            with self._lock:
                blob, _ = self._encode(img)
        except ServiceError:
            raise
        except Exception as e:   # noqa: BLE001 -- the reference maps a failed run to 500 "Inference failed"
            raise ServiceError(500, f"Inference failed: {e}") from e
        return Response(blob, "application/octet-stream", filename=f"{Path(filename).stem}.c2df",
                        headers=timing_headers((time.perf_counter() - t0) * 1000, "compress"))

    def _decode_png(self, data):
        from .decompress import to_u8_hwc
        from .filemaker import unpack_c2df
        from PIL import Image
        enc, header = unpack_c2df(data)
        keys = ("z_bit_stream", "h_bit_stream", "img_shape", "feat_shape", "stack_shape", "token_length", "z_indices_shape")
        x_hat = self.model.decode_batch([{k: enc[k] for k in keys}])
        pl, pr, pt, pb = header.get("padding", [0, 0, 0, 0])
        H, W = x_hat.shape[2] - pt - pb, x_hat.shape[3] - pl - pr                            # decompress.py:110-112
        arr = to_u8_hwc(x_hat[0, :, pt:pt + H, pl:pl + W].clamp(-1, 1) * 0.5 + 0.5)
        buf = io.BytesIO()
        Image.fromarray(arr).save(buf, format="PNG")
        return buf.getvalue()

    def decompress(self, filename, data):
        """POST /decompress (webapp.py:167-226): one uploaded .c2df -> the reconstructed PNG"""
        t0 = time.perf_counter()
        try:
            with self._lock:
                png = self._decode_png(data)
        except Exception as e:   # noqa: BLE001
            raise ServiceError(500, f"Inference failed: {e}") from e
        return Response(png, "image/png", filename=f"{Path(filename).stem}.png",
                        headers=timing_headers((time.perf_counter() - t0) * 1000, "decompress"))

    def compress_many(self, files):
        """the reference zips the outputs when a job produced several (webapp.py:151-165); files = [(name, bytes), ...]"""
        t0 = time.perf_counter()
        buf = io.BytesIO()
        with zipfile.ZipFile(buf, "w", compression=zipfile.ZIP_DEFLATED) as zf:
            for name, data in files:
                zf.writestr(f"bitstreams/{Path(name).stem}.c2df", self.compress(name, data).body)
        return Response(buf.getvalue(), "application/zip", filename="c2df.zip",
                        headers=timing_headers((time.perf_counter() - t0) * 1000, "compress"))

    # ------------------------------------------------------------------ files and previews
    def serve_file(self, path):
        """GET /file (webapp.py:67-74)"""
        p = Path(path).resolve()
        if not p.exists() or not p.is_file():
            raise ServiceError(404, "File not found")
        if p.suffix.lower() not in IMAGE_EXTS and p.suffix.lower() != ".c2df":
            raise ServiceError(403, "Forbidden file type")
        return Response(p.read_bytes(), media_type_of(p), filename=p.name)

    def resolve_media_path(self, raw):
        """webapp.py:24-38: the path itself, else the first file of that name under the media roots"""
        try:
            p = Path(raw).expanduser()
        except Exception:   # noqa: BLE001
            return None
        if p.exists() and p.is_file():
            return p.resolve()
        name = Path(raw).name
        for root in self.media_roots:
            try:
                for cand in root.rglob(name):
                    if cand.is_file() and (cand.suffix.lower() in IMAGE_EXTS or cand.suffix.lower() == ".c2df"):
                        return cand.resolve()
            except OSError:
                continue
        return None

    def preview_url_for_path(self, path):
        """webapp.py:76-111: images are served as they are; a .c2df is decoded ONCE into the preview cache (keyed by
        path|mtime|size) -- in process, not by a decompress.py subprocess"""
        p = self.resolve_media_path(path)
        if not p:
            return ""
        suf = p.suffix.lower()
        if suf in IMAGE_EXTS:
            return f"/file?path={quote(str(p))}"
        if suf == ".c2df":
            st = p.stat()
            key = hashlib.sha1((str(p.resolve()) + f"|{int(st.st_mtime)}|{st.st_size}").encode("utf-8")).hexdigest()
            out_png = self.preview_cache / f"{key}.png"
            if not out_png.exists():
                try:
                    with self._lock:
                        png = self._decode_png(p.read_bytes())
                    tmp = out_png.with_suffix(f".{os.getpid()}.tmp")
                    tmp.write_bytes(png)
                    os.replace(tmp, out_png)
                except Exception:   # noqa: BLE001 -- like the reference: fall back to the raw file
                    return f"/file?path={quote(str(p))}"
            return f"/file?path={quote(str(out_png))}"
        return ""

    # ------------------------------------------------------------------ search endpoints
    def _load_index(self, index_dir):
        from .search import load_index
        d = Path(index_dir or self.index_dir).resolve()
        stamp = tuple(sorted((f.name, f.stat().st_mtime_ns, f.stat().st_size) for f in d.iterdir())) if d.is_dir() else None
        hit = self._index.get(str(d))
        if hit is None or hit[0] != stamp:
            vecs, ids = load_index(d)
            hit = (stamp, torch.from_numpy(np.ascontiguousarray(vecs, dtype=np.float32)).to(self.device), ids)
            self._index[str(d)] = hit                                  # the database stays resident in HBM between queries
        return hit[1], hit[2]

    def _search(self, q, topk, index_dir):
        from . import ops
        db, ids = self._load_index(index_dir)
        k = max(1, min(int(topk), db.shape[0]))
        with self._lock:
            dq = torch.from_numpy(np.ascontiguousarray(q, dtype=np.float32)).to(self.device)
            s, i = ops.topk_rows(ops.gemm(dq, db, w_const=False), k)                   # exact inner product + top-k (IndexFlatIP.search)
            s, i = s.cpu().numpy(), i.cpu().numpy()
        return [{"path": ids[j], "score": float(s[0, r])} for r, j in enumerate(i[0]) if j != -1]

    def _stream(self, start_meta, make_query, topk, index_dir):
        """the NDJSON protocol shared by the three search endpoints (webapp.py:236-255)"""
        t0 = time.perf_counter()
        ms = lambda: int((time.perf_counter() - t0) * 1000)   # noqa: E731
        yield ndjson(dict({"type": "meta", "stage": "start"}, **start_meta, topk=int(topk)))
        try:
            items = self._search(make_query(), topk, index_dir)
            yield ndjson({"type": "meta", "stage": "searched", "count": len(items), "elapsed_ms": ms()})
            for it in items:
                yield ndjson({"type": "item", "path": it["path"], "score": float(it["score"]),
                              "preview_url": self.preview_url_for_path(it["path"])})
            yield ndjson({"type": "done", "elapsed_ms": ms()})
        except Exception as e:   # noqa: BLE001 -- the protocol reports errors in-band
            yield ndjson({"type": "error", "detail": str(e)})

    def search_text(self, body):
        """POST /search/stream/text (webapp.py:228-257); body = {"text", "topk", "index_dir"[, "token_ids"]}"""
        text = (body.get("text") or "").strip()
        topk = int(body.get("topk") or 10)
        if not text:
            raise ServiceError(400, "text is required")

        def query():
            from .clip import ClipTextHIP
            from .search import encode_text, tokenize
            toks = tokenize(text, self.ccfg.ctx, body.get("token_ids"))
            with self._lock:
                if self._text is None:
                    self._text = ClipTextHIP(self._csd, self.ccfg, self.device)
                return encode_text(toks, self._text)

        return self._stream({"query_type": "text", "query": text}, query, topk, body.get("index_dir"))

    def search_image(self, filename, data, topk=10, index_dir=None):
        """POST /search/stream/image (webapp.py:259-287)"""
        def query():
            img = self._image_from_bytes(data)
            with self._lock:
                return self.clipc.image_to_unit_vec(img)[None, :]

        return self._stream({"query_type": "image", "filename": filename}, query, topk, index_dir)

    def search_c2df(self, filename, data, topk=10, index_dir=None):
        """POST /search/stream/c2df (webapp.py:289-317): needs no model -- the query is the embedded CLIP code"""
        def query():
            from .search import embedded_clip_vector
            return embedded_clip_vector(data)[0][None, :]

        return self._stream({"query_type": "c2df", "filename": filename}, query, topk, index_dir)


# ---------------------------------------------------------------------- optional stdlib transport
def http_handler(service):
    """-> a `http.server.BaseHTTPRequestHandler` subclass that routes the reference's URLs to `service` (stdlib only:
    multipart bodies are parsed with `email`).  `ThreadingHTTPServer(("127.0.0.1", 8000), http_handler(svc)).serve_forever()`"""
    import email
    import email.policy
    from http.server import BaseHTTPRequestHandler
    from urllib.parse import parse_qs, urlparse

    class Handler(BaseHTTPRequestHandler):
        def _upload(self):
            n = int(self.headers.get("Content-Length", "0"))
            raw = b"Content-Type: " + self.headers.get("Content-Type", "").encode() + b"\r\n\r\n" + self.rfile.read(n)
            msg = email.message_from_bytes(raw, policy=email.policy.HTTP)
            for part in msg.iter_parts():
                if part.get_filename():
                    return part.get_filename(), part.get_payload(decode=True)
            raise ServiceError(400, "multipart field 'file' is required")

        def _send(self, r):
            self.send_response(r.status)
            self.send_header("Content-Type", r.media_type)
            self.send_header("Content-Length", str(len(r.body)))
            if r.filename:
                self.send_header("Content-Disposition", f'attachment; filename="{r.filename}"')
            for k, v in r.headers.items():
                self.send_header(k, v)
            self.end_headers()
            self.wfile.write(r.body)

        def _send_stream(self, it):
            self.send_response(200)
            self.send_header("Content-Type", "application/x-ndjson")
            self.end_headers()
            for line in it:
                self.wfile.write(line)
                self.wfile.flush()

        def _route(self, method):
            u = urlparse(self.path)
            qs = {k: v[0] for k, v in parse_qs(u.query).items()}
            try:
                if method == "GET" and u.path == "/file":
                    return self._send(service.serve_file(qs.get("path", "")))
                if method == "POST" and u.path == "/compress":
                    return self._send(service.compress(*self._upload()))
                if method == "POST" and u.path == "/decompress":
                    return self._send(service.decompress(*self._upload()))
                if method == "POST" and u.path == "/search/stream/text":
                    body = json.loads(self.rfile.read(int(self.headers.get("Content-Length", "0"))) or b"{}")
                    return self._send_stream(service.search_text(body))
                if method == "POST" and u.path in ("/search/stream/image", "/search/stream/c2df"):
                    fn = service.search_image if u.path.endswith("image") else service.search_c2df
                    return self._send_stream(fn(*self._upload(), topk=int(qs.get("topk", 10)), index_dir=qs.get("index_dir")))
                raise ServiceError(404, "Not Found")
            except ServiceError as e:
                self._send(Response(json.dumps({"detail": e.detail}).encode(), "application/json", status=e.status))

        def do_GET(self):   # noqa: N802
            self._route("GET")

        def do_POST(self):   # noqa: N802
            self._route("POST")

    return Handler
