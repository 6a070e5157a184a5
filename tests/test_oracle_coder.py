"""CPU: the plain-C oracle (oracle/sgic_oracle.c) against golden vectors produced by the REAL reference
coder (oracle/gen_golden_coder.py) and the reference's worked example IO/bitstreams/apple.c2df."""
import os
import struct

import numpy as np
import pytest

from oracle import orc


@pytest.fixture(scope="module")
def table(golden_dir):
    t = np.load(os.path.join(golden_dir, "cdf_table.npz"))
    return orc.Table(t["cdf"], t["cdf_length"], t["offset"])


def test_pmf_to_quantized_cdf_kats(golden_dir):
    k = np.load(os.path.join(golden_dir, "pmf_kats.npz"))
    for i in range(int(k["n"])):
        got = orc.pmf_to_quantized_cdf(k[f"pmf_{i}"], 16)
        assert np.array_equal(got, k[f"cdf_{i}"]), i
        assert got[0] == 0 and got[-1] == 65536 and np.all(np.diff(got.astype(np.int64)) > 0)


def test_cdf_table_row0_and_shape(table):
    # row 0 = [0,1,2,65533,65534,65535,65536], lengths 7..103, offsets -2..-50
    assert table.cdf.shape == (256, 103)
    assert table.cdf[0, :7].tolist() == [0, 1, 2, 65533, 65534, 65535, 65536]
    assert table.sizes.min() == 7 and table.sizes.max() == 103
    assert table.offsets.max() == -2 and table.offsets.min() == -50


def test_rans_encode_kats_bit_exact(golden_dir, table):
    k = np.load(os.path.join(golden_dir, "rans_kats.npz"))
    for i in range(int(k["n"])):
        got = orc.rans_encode(k[f"sym_{i}"], k[f"idx_{i}"], table)
        assert got == k[f"stream_{i}"].tobytes(), f"kat {i}"


def test_rans_decode_kats_multicall_cursor(golden_dir, table):
    k = np.load(os.path.join(golden_dir, "rans_kats.npz"))
    for i in range(int(k["n"])):
        sym, idx, cuts = k[f"sym_{i}"], k[f"idx_{i}"], k[f"cuts_{i}"]
        d = orc.Decoder(k[f"stream_{i}"].tobytes(), table)
        got = np.concatenate([d.decode(idx[a:b]) for a, b in zip(cuts[:-1], cuts[1:])])
        assert np.array_equal(got, np.where(idx < 0, 0, sym)), f"kat {i}"


def test_rans_degenerate_inputs_self_consistent(table):
    # the reference crashes on these; the build defines the natural answer
    s = orc.rans_encode(np.zeros(16, np.int16), -np.ones(16, np.int16), table)
    assert s == bytes([0x01, 0x00, 0x00, 0x80, 0x00])
    s = orc.rans_encode(np.zeros(0, np.int16), np.zeros(0, np.int16), table)
    assert s == bytes([0x01, 0x00, 0x00, 0x80, 0x00])
    sym = np.array([-3, 2, 0], np.int16)
    idx = np.array([0, 0, 0], np.int16)
    s = orc.rans_encode(sym, idx, table)
    assert np.array_equal(orc.Decoder(s, table).decode(idx), sym)


def _parse_c2df(data):
    assert data[:4] == b"C2DF"
    off = 6
    hlen, = struct.unpack_from("<I", data, off)
    off += 4 + hlen
    n, = struct.unpack_from("<I", data, off)
    off += 4
    out = {}
    for _ in range(n):
        kl, = struct.unpack_from("<H", data, off)
        off += 2
        key = data[off:off + kl].decode()
        off += kl
        t = data[off]
        off += 1
        if t == 2:
            out[key] = struct.unpack_from("<q", data, off)[0]
            off += 8
        else:
            L, = struct.unpack_from("<I", data, off)
            off += 4
            out[key] = (t, data[off:off + L])
            off += L
    return out


def test_pack12_matches_reference_sample_zstream(golden_dir):
    """z_bit_stream of the reference's apple.c2df == pack12(indices) || 0x40 (torchac, uniform cdf)."""
    d = _parse_c2df(open(os.path.join(golden_dir, "ref_apple.c2df"), "rb").read())
    t, payload = d["z_bit_stream"]
    L, = struct.unpack_from("<I", payload, 0)
    z = payload[4:4 + L]
    n = d["token_length"]
    assert n == 512 and len(z) == 769 and z[-1] == 0x40
    idx = orc.unpack12(z, n)
    assert idx.min() >= 0 and idx.max() < 4096
    assert orc.pack12(idx) == z
    assert orc.torchac_uniform_encode(idx) == z


def test_pack12_equals_torchac_restatement_odd_and_even():
    rng = np.random.default_rng(3)
    for n in (0, 1, 2, 3, 31, 32, 33, 128, 255):
        idx = rng.integers(0, 4096, size=n).astype(np.int16)
        a, b = orc.pack12(idx), orc.torchac_uniform_encode(idx)
        assert a == b, n
        assert np.array_equal(orc.unpack12(a, n), idx)


def test_resize_bicubic_matches_pillow():
    from PIL import Image
    rng = np.random.default_rng(5)
    for (h, w, oh, ow) in [(256, 256, 224, 224), (300, 256, 262, 224), (100, 180, 224, 403), (512, 512, 224, 224)]:
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img, "RGB").resize((ow, oh), Image.BICUBIC))
        got = orc.resize_bicubic_u8(np.ascontiguousarray(img.transpose(2, 0, 1)), oh, ow).transpose(1, 2, 0)
        assert np.array_equal(got, ref), (h, w, oh, ow)


def test_oracle_coder_on_the_wide_reference_fixture(golden_dir):
    """the C restatement of the coder against the 33 streams the REAL reference produced end to end
    (tests/golden/streams_small.npz, oracle/gen_golden_streams.py): encode is byte-identical, decode returns the
    reference's symbols (0 where the index says skip) and consumes the stream"""
    import os
    g = np.load(os.path.join(golden_dir, "streams_small.npz"))
    t = np.load(os.path.join(golden_dir, "cdf_table.npz"))
    tab = orc.Table(t["cdf"], t["cdf_length"], t["offset"])
    for name in g["names"]:
        sym, idx, ref = g[f"{name}.sym"].reshape(-1), g[f"{name}.idx"].reshape(-1), g[f"{name}.stream"].tobytes()
        assert orc.rans_encode(sym, idx, tab) == ref, name
        d = orc.Decoder(ref, tab)
        assert np.array_equal(d.decode(idx), np.where(idx < 0, 0, sym)), name
        assert orc.pack12(g[f"{name}.vq"]).endswith(b"\x40") or len(g[f"{name}.vq"]) % 2 == 1
