"""Host side of the streaming ingest (sgic_amd.ingest) -- no GPU: geometry planning from headers only,
order, bounded read-ahead, error propagation."""
import os
import threading
import time

import numpy as np
import pytest
from PIL import Image


def _write(path, h, w, seed):
    rng = np.random.default_rng(seed)
    Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(path)


@pytest.fixture()
def corpus(tmp_path):
    files = []
    sizes = [(40, 56), (64, 64), (40, 56), (64, 64), (64, 64), (17, 300), (40, 56), (64, 64), (64, 64)]
    for i, (h, w) in enumerate(sizes):
        p = str(tmp_path / f"im{i:02d}.png")      # PNG: lossless, so the decoded pixels are known exactly
        _write(p, h, w, i)
        files.append(p)
    return files, sizes


def test_plan_and_batches_match_files(corpus):
    import sgic_amd  # noqa
    from sgic_amd.ingest import ShardLoader, image_size, plan_batches
    files, sizes = corpus
    assert [image_size(f) for f in files] == sizes
    plan = plan_batches(files, sizes, 2)
    assert [(h, w, i) for h, w, i in plan] == [(40, 56, [0, 2]), (40, 56, [6]), (64, 64, [1, 3]), (64, 64, [4, 7]), (64, 64, [8]),
                                                (17, 300, [5])]
    ld = ShardLoader(files, batch_size=2, workers=3, depth=2, pin=False)
    seen = []
    for b in ld:
        assert b.u8.shape == (len(b.indices), b.H, b.W, 3) and b.u8.dtype.is_floating_point is False
        for j, i in enumerate(b.indices):
            assert np.array_equal(b.u8[j].numpy(), np.asarray(Image.open(files[i]).convert("RGB")))
        seen += b.indices
        b.release()
    ld.close()
    assert sorted(seen) == list(range(len(files))) and len(seen) == len(files)


def test_read_ahead_is_bounded(corpus):
    """the producer may run at most `depth` queued batches + the slots the consumer has not released ahead of the
    consumer: host memory does not grow with the shard"""
    import sgic_amd  # noqa
    from sgic_amd import ingest
    files, _ = corpus
    files = files * 6                               # 54 images, batch 1 -> 54 batches
    decoded = []
    real = ingest.decode_rgb_u8

    def counting(path, out):
        decoded.append(path)
        real(path, out)

    ingest.decode_rgb_u8 = counting
    try:
        ld = ingest.ShardLoader(files, batch_size=1, workers=2, depth=2, pin=False)
        it = iter(ld)
        first = next(it)
        time.sleep(0.5)                             # give the producer every chance to run ahead
        assert len(decoded) <= 1 + 2 + 2 + 1, len(decoded)   # consumed + queue depth + spare slots (+ one in flight)
        first.release()
        rest = 0
        for b in it:
            b.release()
            rest += 1
        assert rest == len(files) - 1
        ld.close()
    finally:
        ingest.decode_rgb_u8 = real


def test_decode_error_reaches_the_consumer(corpus, tmp_path):
    import sgic_amd  # noqa
    from sgic_amd.ingest import ShardLoader
    files, _ = corpus
    bad = str(tmp_path / "zz_broken.png")
    good = open(files[1], "rb").read()
    open(bad, "wb").write(good[:len(good) // 2])    # valid header (size readable), truncated pixel data
    ld = ShardLoader(files + [bad], batch_size=4, workers=2, depth=2, pin=False)
    with pytest.raises(Exception):
        for b in ld:
            b.release()
    ld.close()
    assert not any(t.name == "sgic-ingest" and t.is_alive() for t in threading.enumerate())


def test_get_padding_size_is_the_one_mirror():
    """compress.py imports the reference-named helper instead of carrying its own copy"""
    import sgic_amd  # noqa
    import inspect
    from sgic_amd import compress
    from sgic_amd.entropy.compression_model import get_padding_size
    assert compress.get_padding_size is get_padding_size
    assert get_padding_size(859, 1000, p=256) == (0, 24, 0, 165)        # the reference's apple.jpg geometry -> 1024 x 1024
    assert "def get_padding_size" not in inspect.getsource(compress)


def test_duplicate_stems_keep_only_the_sorted_last_file():
    """a.jpg and a.png of different geometry must not race for a.c2df / a.npy -- the earlier one is never encoded"""
    from sgic_amd.compress import unique_stems
    files = sorted(["d/a.jpg", "d/a.png", "d/b.png", "d/c.jpeg", "d/c.jpg", "d/c.png", "d/z.bmp"])
    assert unique_stems(files) == ["d/a.png", "d/b.png", "d/c.png", "d/z.bmp"]
    assert unique_stems([]) == [] and unique_stems(["x/only.png"]) == ["x/only.png"]


def test_codes_to_unit_is_the_reference_expression_for_all_256_codes():
    """search.py:21 of the reference computes (q / 255.0) * 2.0 - 1.0 in fp32; c * (2/255) - 1 is one ulp off for 111 codes"""
    import numpy as np
    from sgic_amd.search import codes_to_unit, unit_rows
    c = np.arange(256, dtype=np.uint8)
    ref = (c.astype(np.float32) / 255.0) * 2.0 - 1.0
    assert ref.dtype == np.float32 and np.array_equal(codes_to_unit(c), unit_rows(ref))
    assert int((ref != c.astype(np.float32) * np.float32(2.0 / 255.0) - np.float32(1.0)).sum()) == 111


def test_get_padding_size_pads_right_and_bottom_to_the_multiple():
    from sgic_amd.entropy.compression_model import get_padding_size
    for h, w, p in [(859, 1000, 256), (256, 256, 256), (1, 1, 64), (257, 512, 256), (64, 65, 64), (1024, 1, 256)]:
        nh, nw = (h + p - 1) // p * p, (w + p - 1) // p * p
        assert get_padding_size(h, w, p) == (0, nw - w, 0, nh - h)
    assert get_padding_size(100, 100) == (0, 28, 0, 28)          # default p = 64 as the reference
