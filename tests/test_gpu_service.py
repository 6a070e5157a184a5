"""GPU: the resident service (sgic_amd.service) driven without any network stack -- endpoint semantics of the
reference's webapp.py (status codes, media types, file names, timing headers, NDJSON protocol), model built once."""
import io
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _png(h, w, seed):
    from PIL import Image
    from sgic_amd.data import synth_images
    x = synth_images(1, 256 * ((h + 255) // 256), 256 * ((w + 255) // 256), seed)[0, :, :h, :w]
    buf = io.BytesIO()
    Image.fromarray(((x * 0.5 + 0.5) * 255).round().byte().permute(1, 2, 0).numpy()).save(buf, format="PNG")
    return buf.getvalue()


@pytest.fixture(scope="module")
def svc(tmp_path_factory):
    import sgic_amd  # noqa
    from sgic_amd import compress
    from sgic_amd.service import ResidentService
    root = tmp_path_factory.mktemp("svc")
    src = root / "imgs"
    src.mkdir()
    for i in range(6):
        (src / f"im{i}.png").write_bytes(_png(256, 256, 900 + i))
    assert compress.main(["--dataset_dir", str(src), "--save_dir", str(root / "out"), "--small", "--batch_size", "4"]) == 0
    s = ResidentService(small=True, index_dir=str(root / "out" / "faiss"), preview_cache=str(root / "previews"),
                        media_roots=[str(root)])
    return s, root


def _lines(it):
    return [json.loads(ln) for ln in b"".join(it).decode().splitlines()]


def test_compress_and_decompress_endpoints(svc):
    from PIL import Image
    from sgic_amd.filemaker import unpack_c2df
    s, root = svc
    data = _png(200, 300, 5)
    r = s.compress("photo.final.png", data)
    assert r.status == 200 and r.media_type == "application/octet-stream" and r.filename == "photo.final.c2df"
    assert r.headers["X-SIC-Stage"] == "compress" and int(r.headers["X-SIC-Elapsed-MS"]) >= 0 and r.headers["X-SIC-Server-Clock"].endswith("Z")
    enc, hdr = unpack_c2df(r.body)
    assert hdr["image_hw"] == [200, 300] and hdr["padding"] == [0, 212, 0, 56] and "clip_stream" in enc
    # the resident path and the CLI driver write the same bytes for the same file
    d = root / "one"
    d.mkdir(exist_ok=True)
    (d / "photo.final.png").write_bytes(data)
    from sgic_amd import compress
    assert compress.main(["--dataset_dir", str(d), "--save_dir", str(root / "one_out"), "--small"]) == 0
    assert (root / "one_out" / "bitstreams" / "photo.final.c2df").read_bytes() == r.body
    # decompress: PNG of the original (unpadded) size
    r2 = s.decompress("photo.final.c2df", r.body)
    assert r2.media_type == "image/png" and r2.filename == "photo.final.png" and r2.headers["X-SIC-Stage"] == "decompress"
    assert Image.open(io.BytesIO(r2.body)).size == (300, 200)
    from sgic_amd.service import ServiceError
    with pytest.raises(ServiceError) as e:
        s.decompress("junk.c2df", b"not a container")
    assert e.value.status == 500 and e.value.detail.startswith("Inference failed")
    with pytest.raises(ServiceError) as e:
        s.compress("junk.png", b"\x89PNG not really")
    assert e.value.status == 500
    z = s.compress_many([("a.png", data), ("b.png", _png(256, 256, 6))])
    import zipfile
    assert z.media_type == "application/zip" and sorted(zipfile.ZipFile(io.BytesIO(z.body)).namelist()) == ["bitstreams/a.c2df", "bitstreams/b.c2df"]


def test_search_streams_follow_the_ndjson_protocol(svc):
    s, root = svc
    c2df = (root / "out" / "bitstreams" / "im3.c2df").read_bytes()
    ev = _lines(s.search_c2df("im3.c2df", c2df, topk=4))
    assert ev[0] == {"type": "meta", "stage": "start", "query_type": "c2df", "filename": "im3.c2df", "topk": 4}
    assert ev[1]["type"] == "meta" and ev[1]["stage"] == "searched" and ev[1]["count"] == 4
    items = [e for e in ev if e["type"] == "item"]
    assert len(items) == 4 and os.path.basename(items[0]["path"]) == "im3.c2df" and items[0]["score"] > 0.99
    assert [it["score"] for it in items] == sorted((it["score"] for it in items), reverse=True)
    assert ev[-1]["type"] == "done" and ev[-1]["elapsed_ms"] >= 0
    # previews: a .c2df hit is decoded once into the cache and served from there afterwards
    assert items[0]["preview_url"].startswith("/file?path=") and items[0]["preview_url"].endswith(".png")
    from urllib.parse import unquote
    png_path = unquote(items[0]["preview_url"].split("=", 1)[1])
    stamp = os.stat(png_path).st_mtime_ns
    assert s.preview_url_for_path(items[0]["path"]) == items[0]["preview_url"] and os.stat(png_path).st_mtime_ns == stamp
    f = s.serve_file(png_path)
    assert f.media_type == "image/png" and f.body[:4] == b"\x89PNG"
    # image query: the uploaded image of corpus item 2 retrieves item 2 first (same CLIP tower as the compress side)
    ev = _lines(s.search_image("q.png", (root / "imgs" / "im2.png").read_bytes(), topk=3))
    items = [e for e in ev if e["type"] == "item"]
    assert os.path.basename(items[0]["path"]) == "im2.c2df" and ev[0]["query_type"] == "image"
    # text query: BPE ids supplied (open_clip's tokenizer is not in the image); empty text is a 400 like the reference
    ev = _lines(s.search_text({"text": "an apple", "topk": 2, "token_ids": "49406,320,3055,49407"}))
    assert ev[0]["query"] == "an apple" and [e["type"] for e in ev] == ["meta", "meta", "item", "item", "done"]
    from sgic_amd.service import ServiceError
    with pytest.raises(ServiceError) as e:
        s.search_text({"text": "  "})
    assert e.value.status == 400 and e.value.detail == "text is required"
    # errors inside a stream are reported in-band
    ev = _lines(s.search_c2df("x.c2df", c2df, topk=3, index_dir=str(root / "nowhere")))
    assert ev[0]["type"] == "meta" and ev[-1]["type"] == "error" and "nowhere" in ev[-1]["detail"]
    ev = _lines(s.search_text({"text": "needs a tokenizer"}))
    assert ev[-1]["type"] == "error" and "token" in ev[-1]["detail"].lower()


def test_file_endpoint_status_codes(svc):
    from sgic_amd.service import ServiceError
    s, root = svc
    with pytest.raises(ServiceError) as e:
        s.serve_file(str(root / "missing.png"))
    assert e.value.status == 404
    (root / "secret.txt").write_text("x")
    with pytest.raises(ServiceError) as e:
        s.serve_file(str(root / "secret.txt"))
    assert e.value.status == 403
    r = s.serve_file(str(root / "out" / "bitstreams" / "im0.c2df"))
    assert r.media_type == "application/octet-stream" and r.filename == "im0.c2df"
    assert s.preview_url_for_path("no_such_file_anywhere.png") == ""
    assert s.preview_url_for_path("im1.png").endswith("im1.png")          # found by name under the media roots


@pytest.mark.perf
def test_model_is_built_once_and_requests_are_fast(svc):
    """the point of the resident object: after the one-off build, a request is milliseconds, not a model build"""
    import time
    s, root = svc
    data = (root / "imgs" / "im0.png").read_bytes()
    s.compress("warm.png", data)
    t0 = time.perf_counter()
    for _ in range(5):
        s.compress("im0.png", data)
    per = (time.perf_counter() - t0) / 5
    assert per < 0.5 and per < s.build_seconds, (per, s.build_seconds)


def test_http_adapter_routes_without_a_socket(svc):
    """the stdlib handler maps the reference's URLs to the service; driven with in-memory request/response files"""
    from sgic_amd.service import http_handler
    s, root = svc
    H = http_handler(s)
    body = (b"--BOUND\r\nContent-Disposition: form-data; name=\"file\"; filename=\"up.png\"\r\nContent-Type: image/png\r\n\r\n"
            + (root / "imgs" / "im1.png").read_bytes() + b"\r\n--BOUND--\r\n")
    req = (b"POST /compress HTTP/1.1\r\nHost: x\r\nContent-Type: multipart/form-data; boundary=BOUND\r\nContent-Length: "
           + str(len(body)).encode() + b"\r\n\r\n" + body)

    class _Sock:
        def __init__(self, data):
            self.r, self.w = io.BytesIO(data), io.BytesIO()

        def makefile(self, mode, *a, **k):
            return self.r if "r" in mode else self.w

        def sendall(self, b):
            self.w.write(b)

    sock = _Sock(req)
    h = H.__new__(H)
    h.request, h.client_address, h.server = sock, ("127.0.0.1", 0), None
    h.rfile, h.wfile = sock.r, sock.w
    h.handle_one_request()
    out = sock.w.getvalue()
    head, _, payload = out.partition(b"\r\n\r\n")
    assert head.startswith(b"HTTP/1.0 200") or head.startswith(b"HTTP/1.1 200")
    assert b"X-SIC-Stage: compress" in head and b'filename="up.c2df"' in head and payload[:4] == b"C2DF"
