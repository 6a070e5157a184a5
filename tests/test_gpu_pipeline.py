"""GPU: the drivers end to end on a small synthetic corpus (SMALL/TINY architectures): compress.py ->
.c2df + clip_vecs + FAISS index -> search (top-k parity vs numpy exact search) -> decompress.py -> PNGs."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_compress_search_decompress_cli(tmp_path):
    from PIL import Image
    import sgic_amd  # noqa
    from sgic_amd import compress, decompress, search
    from sgic_amd.data import synth_images
    from sgic_amd.faiss_io import read_index_flat_ip
    from sgic_amd.filemaker import unpack_c2df
    src = tmp_path / "imgs"
    src.mkdir()
    sizes = [(256, 256)] * 5 + [(200, 300)] * 2 + [(512, 256)]
    for i, (h, w) in enumerate(sizes):
        x = synth_images(1, 256 * ((h + 255) // 256), 256 * ((w + 255) // 256), 100 + i)[0, :, :h, :w]
        Image.fromarray(((x * 0.5 + 0.5) * 255).round().byte().permute(1, 2, 0).numpy()).save(src / f"im{i:02d}.png")
    out = tmp_path / "out"
    assert compress.main(["--dataset_dir", str(src), "--save_dir", str(out), "--small", "--batch_size", "4"]) == 0
    files = sorted(os.listdir(out / "bitstreams"))
    assert files == [f"im{i:02d}.c2df" for i in range(len(sizes))]
    enc, hdr = unpack_c2df(out / "bitstreams" / "im05.c2df")
    assert hdr["image_hw"] == [200, 300] and hdr["padding"] == [0, 212, 0, 56] and hdr["version"] == 2
    assert list(enc.keys()) == ["z_bit_stream", "h_bit_stream", "img_shape", "feat_shape", "stack_shape", "token_length",
                                "z_indices_shape", "clip_stream", "clip_meta"]
    vecs = read_index_flat_ip(str(out / "faiss" / "index.faiss"))
    ids = (out / "faiss" / "ids.txt").read_text().splitlines()
    assert vecs.shape == (len(sizes), 64) and len(ids) == len(sizes)
    for i in range(len(sizes)):
        v = np.load(out / "clip_vecs" / f"im{i:02d}.npy")
        assert np.allclose(vecs[i], v / (np.linalg.norm(v) + 1e-12), atol=1e-6)
    # query-c2df: the embedded u8 code of image 3 must retrieve image 3 first; GPU top-k == numpy exact search
    q, _ = search.decode_clip_from_c2df(out / "bitstreams" / "im03.c2df")
    s, i = search.search_gpu(q[None], vecs, 5)
    ref = np.argsort(-(q[None] @ vecs.T), axis=1, kind="stable")[:, :5]
    assert i[0, 0] == 3 and np.array_equal(i, ref)
    assert np.allclose(s[0], (q[None] @ vecs.T)[0, ref[0]], atol=1e-5)
    # query-text plumbing: tokenizer bypass -> text tower -> top-k (synthetic weights: ranks are arbitrary, shapes are not)
    with pytest.raises(ValueError):
        search.tokenize("x", 77, ",".join(["1"] * 78))
    toks = search.tokenize("ignored", 77, "49406,320,1125,49407")
    assert toks.shape == (1, 77) and toks[0, 3] == 49407 and toks[0, 4:].sum() == 0
    # decompress
    assert decompress.main(["--dataset_dir", str(out / "bitstreams"), "--save_dir", str(out), "--small"]) == 0
    for i, (h, w) in enumerate(sizes):
        im = Image.open(out / "results" / f"im{i:02d}.png")
        assert im.size == (w, h)


def test_search_topk_parity_10k_corpus(tmp_path):
    """BASELINE.json configs[3] at its size: a 10 000 x 512 IndexFlatIP (written in the FAISS on-disk layout and read
    back), 16 queries, top-10: GPU GEMM + top-k kernel vs numpy exact search (search.py:113-120).  Indices must be
    identical (ties -> lower index, as IndexFlatIP); scores within 1e-5 (fp32 dot products of unit vectors)."""
    import sgic_amd  # noqa
    from sgic_amd import search
    from sgic_amd.faiss_io import FaissDB, read_index_flat_ip
    rng = np.random.default_rng(7)
    db = rng.standard_normal((10000, 512), dtype=np.float32)
    db /= np.linalg.norm(db, axis=1, keepdims=True)
    db[4321] = db[1234]                      # an exact duplicate: a tie that must resolve to the lower index
    fdb = FaissDB(str(tmp_path), 512)
    for i in range(db.shape[0]):
        fdb.add(db[i], f"bitstreams/im{i:05d}.c2df")
    fdb.persist()
    vecs, paths = search.load_index(tmp_path)
    assert vecs.shape == (10000, 512) and len(paths) == 10000 and np.array_equal(vecs, read_index_flat_ip(str(tmp_path / "index.faiss")))
    q = np.concatenate([vecs[[1234, 17, 9999]], rng.standard_normal((13, 512), dtype=np.float32)])
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    s, i = search.search_gpu(q, vecs, 10)
    full = q.astype(np.float64) @ vecs.astype(np.float64).T
    ref = np.argsort(-full, axis=1, kind="stable")[:, :10]
    assert i[0, 0] == 1234 and i[0, 1] == 4321 and i[1, 0] == 17 and i[2, 0] == 9999
    # fp32 rounding can swap neighbours whose fp64 scores differ by < 1e-6; everywhere else the order is exact
    for r in range(q.shape[0]):
        if not np.array_equal(i[r], ref[r]):
            assert set(i[r]) == set(ref[r]) or np.abs(np.sort(full[r, i[r]])[::-1] - np.sort(full[r, ref[r]])[::-1]).max() < 1e-6, r
        assert np.allclose(s[r], full[r, i[r]], atol=1e-5)


def test_u8_ingest_kernel_bit_exact_vs_torch():
    """sgic_u8hwc_to_f32chw_pad == transforms.ToTensor()(img) * 2.0 - 1.0 followed by F.pad(replicate) (compress.py:161-164,
    258-261), bit for bit, for every byte value and ragged geometries"""
    import sgic_amd  # noqa
    from sgic_amd import ops
    rng = np.random.default_rng(3)
    for (B, H, W, pad) in [(3, 200, 300, (0, 212, 0, 56)), (2, 256, 256, (0, 0, 0, 0)), (1, 1, 1, (0, 255, 0, 255)), (2, 17, 33, (3, 5, 2, 7))]:
        a = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
        a.reshape(-1)[:256] = np.arange(256, dtype=np.uint8)[:a.size]
        ref = torch.from_numpy(a).permute(0, 3, 1, 2).float().div(255.0) * 2.0 - 1.0
        ref = torch.nn.functional.pad(ref, pad, mode="replicate")
        got = ops.u8hwc_to_f32chw_pad(torch.from_numpy(a).cuda(), *pad).cpu()
        assert got.shape == ref.shape and torch.equal(got, ref), (B, H, W, pad)


def test_streaming_compress_equals_per_image_encode_only(tmp_path):
    """the streamed, batched CLI writes, for every file, the bytes `encode_only` + `ClipCodec` give for that image alone
    (reference loop, compress.py:248-291): batching / ingest / pipelining change nothing in the .c2df"""
    from PIL import Image
    import sgic_amd  # noqa
    from sgic_amd import compress, ops
    from sgic_amd import weights as W
    from sgic_amd.codec import ClipCodec, Codec
    from sgic_amd.config import CLIP_TINY, SMALL
    from sgic_amd.data import synth_images
    from sgic_amd.entropy.compression_model import get_padding_size
    from sgic_amd.filemaker import unpack_c2df
    src = tmp_path / "imgs"
    src.mkdir()
    sizes = [(256, 256), (200, 300), (256, 256), (256, 256), (200, 300), (300, 520), (256, 256)]
    for i, (h, w) in enumerate(sizes):
        x = synth_images(1, 768, 768, 500 + i)[0, :, :h, :w]
        Image.fromarray(((x * 0.5 + 0.5) * 255).round().byte().permute(1, 2, 0).numpy()).save(src / f"f{i}.png")
    out = tmp_path / "out"
    assert compress.main(["--dataset_dir", str(src), "--save_dir", str(out), "--small", "--batch_size", "3", "--prefetch", "2"]) == 0
    sd = W.synth_weights(W.encoder_spec(SMALL) + W.codec_misc_spec(SMALL) + W.bottleneck_spec(SMALL), seed=1234)
    model = Codec(sd, SMALL, "cuda:0")
    model.hybrid_codec.quantize_feat.force_zero_thres = 0.12
    model.hybrid_codec.quantize_feat.update(force=True)
    clipc = ClipCodec(W.synth_weights(W.clip_spec(CLIP_TINY), seed=4321), CLIP_TINY, "cuda:0")
    for i, (h, w) in enumerate(sizes):
        img = compress.load_image(str(src / f"f{i}.png")).cuda()[None]
        pad = get_padding_size(h, w, p=256)
        ref = model.encode_only(torch.nn.functional.pad(img, pad, mode="replicate"))
        enc, hdr = unpack_c2df(out / "bitstreams" / f"f{i}.c2df")
        assert enc["h_bit_stream"] == ref["h_bit_stream"] and enc["z_bit_stream"] == ref["z_bit_stream"], i
        assert tuple(enc["img_shape"]) == tuple(ref["img_shape"]) and hdr["padding"] == list(pad) and hdr["image_hw"] == [h, w]
        v = clipc.image_to_unit_vec(img[0])
        assert np.allclose(np.load(out / "clip_vecs" / f"f{i}.npy"), v, atol=1e-6), i


def test_cli_edge_cases_empty_corpus_and_broken_file(tmp_path):
    """an empty dataset directory is a clean no-op (no index written, like compress.py:297); a file that is not an image
    fails the run loudly (in the header pass, before any GPU work) instead of hanging or writing a partial index"""
    from PIL import Image
    import sgic_amd  # noqa
    from sgic_amd import compress
    from sgic_amd.data import synth_images
    empty, out0 = tmp_path / "empty", tmp_path / "out0"
    empty.mkdir()
    assert compress.main(["--dataset_dir", str(empty), "--save_dir", str(out0), "--small"]) == 0
    assert not (out0 / "faiss" / "index.faiss").exists() and os.listdir(out0 / "bitstreams") == []
    src, out1 = tmp_path / "src", tmp_path / "out1"
    src.mkdir()
    for i in range(3):
        x = synth_images(1, 256, 256, 40 + i)[0]
        Image.fromarray(((x * 0.5 + 0.5) * 255).round().byte().permute(1, 2, 0).numpy()).save(src / f"ok{i}.png")
    (src / "zz_not_an_image.png").write_bytes(b"this is not a PNG")
    with pytest.raises(Exception):
        compress.main(["--dataset_dir", str(src), "--save_dir", str(out1), "--small", "--batch_size", "2"])
    assert not (out1 / "faiss" / "index.faiss").exists()
