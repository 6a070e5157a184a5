"""CPU side of the GPU JPEG decoder: the host parser (sgic_amd/jpeg.py) and the numpy restatement of the decode
(oracle/jpeg_ref.py) against the installed Pillow -- what the reference's Test_Dataset runs (compress.py:160).  Bit-exact."""
import io
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import jpeg_cases  # noqa: E402


@pytest.mark.parametrize("name,data", jpeg_cases.cases(small=True), ids=lambda v: v if isinstance(v, str) else "")
def test_oracle_decode_is_bit_exact_with_pillow(name, data):
    from PIL import Image
    from oracle import jpeg_ref
    ref = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    got = jpeg_ref.decode(data)
    assert got.shape == ref.shape and np.array_equal(got, ref), name


def test_parser_geometry_tables_and_scan_cleaning():
    import sgic_amd  # noqa: F401
    from PIL import Image
    from sgic_amd import jpeg as J
    for name, data in jpeg_cases.cases(small=True):
        im = Image.open(io.BytesIO(data))
        p = J.parse(data)
        assert (p.W, p.H) == im.size and p.ncomp == len(im.getbands()), name
        for k, t in im.quantization.items():          # Pillow hands the tables out in natural order
            assert np.array_equal(p.quant[k], np.asarray(t, dtype=np.uint16)), name
        # the cleaned scan holds no marker: every 0xFF of the file's entropy-coded segment was followed by a stuffed 0x00 or RSTn
        assert len(p.segs) >= 1 and p.segs[0] == 0 and np.all(np.diff(p.segs) > 0)
        if p.restart:
            assert len(p.segs) == -(-p.mcus_x * p.mcus_y // p.restart)
        b = J.JpegBatch([data])
        assert b.params.shape == (1, J.NP) and b.scan.size % J.CHUNK == 0 and b.tabs.size == 4 * J.TAB_BYTES
        assert b.total_blocks == sum(c["bw"] * c["bh"] for c in p.comps)


def test_unsupported_variants_are_refused_not_misdecoded(tmp_path):
    """progressive files (the reference's own sample image is one), CMYK and non-JPEG bytes raise Unsupported -> host decoder"""
    import sgic_amd  # noqa: F401
    from PIL import Image
    from sgic_amd import jpeg as J
    rng = np.random.default_rng(1)
    img = jpeg_cases.natural_like(40, 40, rng)
    for kw in (dict(progressive=True), ):
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, "JPEG", **kw)
        with pytest.raises(J.Unsupported):
            J.parse(buf.getvalue())
    buf = io.BytesIO()
    Image.fromarray(img).convert("CMYK").save(buf, "JPEG")
    with pytest.raises(J.Unsupported):
        J.parse(buf.getvalue())
    with pytest.raises(J.Unsupported):
        J.parse(b"\x89PNG\r\n\x1a\n")


def test_huffman_lookup_table_is_canonical():
    """every code of a DHT decodes to its symbol through the fast / slow lookup the kernel uses"""
    import sgic_amd  # noqa: F401
    from sgic_amd import jpeg as J
    bits = np.array([0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 125], dtype=np.uint8)
    vals = np.arange(int(bits.sum()), dtype=np.uint8)[::-1].copy()
    t = J.huff_table(bits, vals)
    fast, maxcode, valoff, huffval = t[:1024].view(np.uint16), t[1024:1096].view(np.int32), t[1096:1164].view(np.int32), t[1168:]
    code, k = 0, 0
    for l in range(1, 17):
        for _ in range(int(bits[l - 1])):
            if l <= 9:
                e = int(fast[code << (9 - l)])
                assert (e >> 8, e & 255) == (l, int(vals[k]))
            else:
                assert int(fast[code >> (l - 9)]) == 0 and code <= maxcode[l] and int(huffval[valoff[l] + code]) == int(vals[k])
            code += 1
            k += 1
        code <<= 1
