"""GPU JPEG decode (csrc/jpeg.hip) against the installed Pillow -- the decoder behind the reference's
`Image.open(path).convert("RGB")` (compress.py:160).  Integer pipeline: the bar is equality of every pixel."""
import io
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import jpeg_cases  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,data", jpeg_cases.cases(small=False), ids=lambda v: v if isinstance(v, str) else "")
def test_gpu_decode_is_bit_exact_with_pillow(name, data):
    import sgic_amd  # noqa: F401
    from PIL import Image
    from sgic_amd import jpeg as J
    ref = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    got = J.JpegBatch([data]).decode("cuda:0").cpu().numpy()[0]
    assert got.shape == ref.shape
    bad = int((got != ref).sum())
    assert bad == 0, f"{name}: {bad} of {ref.size} samples differ (max {np.abs(got.astype(int) - ref.astype(int)).max()})"


def test_batch_of_mixed_tables_and_samplings():
    """one launch, eight files of one geometry with different quality / sampling / Huffman tables / restart intervals"""
    import sgic_amd  # noqa: F401
    from PIL import Image
    from sgic_amd import jpeg as J
    rng = np.random.default_rng(3)
    datas = []
    for i, kw in enumerate([dict(quality=75), dict(quality=95, subsampling=0), dict(quality=40, subsampling=1), dict(quality=85, optimize=True),
                            dict(quality=60, restart_marker_blocks=5), dict(quality=99), dict(quality=10), dict(quality=80, subsampling=2, optimize=True)]):
        buf = io.BytesIO()
        Image.fromarray(jpeg_cases.natural_like(200, 264, rng)).save(buf, "JPEG", **kw)
        datas.append(buf.getvalue())
    buf = io.BytesIO()
    Image.fromarray(jpeg_cases.natural_like(200, 264, rng, grey=True)).save(buf, "JPEG", quality=77)
    datas.append(buf.getvalue())
    got = J.JpegBatch(datas).decode("cuda:0").cpu().numpy()
    for b, d in enumerate(datas):
        assert np.array_equal(got[b], np.asarray(Image.open(io.BytesIO(d)).convert("RGB"))), b


def test_corrupt_scan_is_reported_not_decoded_silently():
    import sgic_amd  # noqa: F401
    from PIL import Image
    from sgic_amd import jpeg as J
    rng = np.random.default_rng(5)
    buf = io.BytesIO()
    Image.fromarray(jpeg_cases.natural_like(64, 64, rng)).save(buf, "JPEG", quality=90, optimize=True)
    good = buf.getvalue()
    b = J.JpegBatch([good, good])
    b.tabs[4 * J.TAB_BYTES:5 * J.TAB_BYTES] = 0                 # image 1 (a view into the batch blob): an empty DC table -> every code is invalid
    with pytest.raises(RuntimeError):
        b.decode("cuda:0")
    out = b.decode("cuda:0", check=False)
    assert b.last_err.cpu().tolist()[0] == 0 and b.last_err.cpu().tolist()[1] != 0
    assert np.array_equal(out[0].cpu().numpy(), np.asarray(Image.open(io.BytesIO(good)).convert("RGB")))


def test_ingest_takes_the_gpu_path_for_jpeg_and_the_host_path_otherwise(tmp_path):
    """ShardLoader + DeviceIngest: baseline JPEG batches never exist as pixels on the host; progressive JPEG / PNG batches do; the
    tensors the encoder receives are identical either way (= Pillow's pixels)"""
    import sgic_amd  # noqa: F401
    from PIL import Image
    from sgic_amd.ingest import DeviceIngest, ShardLoader
    rng = np.random.default_rng(9)
    files = []
    for i in range(5):
        p = tmp_path / f"a{i}.jpg"
        Image.fromarray(jpeg_cases.natural_like(120, 136, rng)).save(p, "JPEG", quality=80 + i)
        files.append(str(p))
    p = tmp_path / "b0.jpg"
    Image.fromarray(jpeg_cases.natural_like(64, 72, rng)).save(p, "JPEG", progressive=True)      # host path
    files.append(str(p))
    p = tmp_path / "c0.png"
    Image.fromarray(jpeg_cases.natural_like(64, 80, rng)).save(p)
    files.append(str(p))
    ld = ShardLoader(files, batch_size=4, workers=2, depth=2)
    ing = DeviceIngest("cuda:0")
    seen = 0
    for b in ld:
        x, done = ing(b, (0, 0, 0, 0))
        done.synchronize()
        for j, i in enumerate(b.indices):
            ref = torch.from_numpy(np.asarray(Image.open(files[i]).convert("RGB")).copy()).permute(2, 0, 1).float().div(255.0) * 2.0 - 1.0
            assert torch.equal(x[j].cpu(), ref), files[i]
            seen += 1
        assert (b.jpeg is not None) == files[b.indices[0]].endswith(("a0.jpg", "a1.jpg", "a2.jpg", "a3.jpg", "a4.jpg"))
        b.release()
    ld.close()
    assert seen == len(files) and ld.gpu_batches == 2 and ld.host_batches == 2
