"""GPU parity of the HIP entropy-coder kernels (through the C ABI) against the C oracle and the golden
vectors produced by the real reference coder.  Bit-exact: this is integer/byte work."""
import os

import numpy as np
import pytest
import torch

from oracle import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tab(golden_dir):
    t = np.load(os.path.join(golden_dir, "cdf_table.npz"))
    return t["cdf"], t["cdf_length"], t["offset"]


@pytest.fixture(scope="module")
def codec(tab):
    import sgic_amd  # noqa
    from sgic_amd.entropy.MLCodec_rans import RansDecoder, RansEncoder
    enc, dec = RansEncoder(False, 1), RansDecoder(1)
    g = enc.add_cdf(*tab)
    assert dec.add_cdf(*tab) == g
    return enc, dec, g


def test_facade_kats_bit_exact(golden_dir, codec):
    enc, dec, g = codec
    k = np.load(os.path.join(golden_dir, "rans_kats.npz"))
    for i in range(int(k["n"])):
        sym, idx, cuts = k[f"sym_{i}"], k[f"idx_{i}"], k[f"cuts_{i}"]
        enc.reset()
        for a, b in zip(cuts[:-1], cuts[1:]):
            enc.encode_with_indexes(sym[a:b], idx[a:b], g)
        enc.flush()
        got = enc.get_encoded_stream()
        assert got.tobytes() == k[f"stream_{i}"].tobytes(), f"kat {i}"
        dec.set_stream(got)
        out = np.concatenate([dec.decode_stream(idx[a:b], g) for a, b in zip(cuts[:-1], cuts[1:])])
        assert np.array_equal(out, np.where(idx < 0, 0, sym)), f"kat {i} decode"


def test_degenerate_streams(codec, tab):
    enc, dec, g = codec
    enc.reset()
    enc.encode_with_indexes(np.zeros(16, np.int16), -np.ones(16, np.int16), g)
    enc.flush()
    assert enc.get_encoded_stream().tobytes() == bytes([1, 0, 0, 0x80, 0])
    enc.reset()
    enc.flush()
    assert enc.get_encoded_stream().tobytes() == bytes([1, 0, 0, 0x80, 0])
    # tiny high-entropy stream (3 symbols of freq 1): the reference corrupts its heap here
    sym, idx = np.array([-3, 2, 0], np.int16), np.zeros(3, np.int16)
    enc.reset()
    enc.encode_with_indexes(sym, idx, g)
    enc.flush()
    s = enc.get_encoded_stream()
    assert s.tobytes() == orc.rans_encode(sym, idx, orc.Table(*tab))
    dec.set_stream(s)
    assert np.array_equal(dec.decode_stream(idx, g), sym)


def test_truncated_stream_is_detected(codec):
    enc, dec, g = codec
    rng = np.random.default_rng(0)
    idx = rng.integers(0, 256, 512).astype(np.int16)
    sym = rng.integers(-5, 6, 512).astype(np.int16)
    enc.reset()
    enc.encode_with_indexes(sym, idx, g)
    enc.flush()
    s = enc.get_encoded_stream()
    dec.set_stream(s[: len(s) // 2])
    with pytest.raises(RuntimeError):
        dec.decode_stream(idx, g)


def test_batched_encode_decode_vs_oracle_full_size(tab):
    """B=32 images x 4096 symbols (config 2) and B=16 x 16384 (config 5): device-resident batch API,
    every stream byte-identical to the C oracle; decode in 4 cursor-continuing calls."""
    import sgic_amd  # noqa
    from sgic_amd._lib import call
    from sgic_amd.entropy.MLCodec_rans import _Tables
    T = _Tables()
    g = T.add(*tab)
    otab = orc.Table(*tab)
    scale_table = np.exp(np.linspace(np.log(0.11), np.log(64.0), 256))
    dev = torch.device("cuda:0")
    for (B, n) in [(32, 4096), (16, 16384), (3, 1000)]:
        rng = np.random.default_rng(B * 7 + n)
        idx = rng.integers(0, 120, size=(B, n)).astype(np.int16)
        idx[rng.random((B, n)) < 0.4] = -1
        sym = np.rint(rng.standard_normal((B, n)) * scale_table[np.maximum(idx, 0)] * 1.2).astype(np.int16)
        sym[rng.random((B, n)) < 0.002] = 3000  # a few bypass symbols
        d_sym, d_idx = torch.from_numpy(sym).to(dev), torch.from_numpy(idx).to(dev)
        cap = 2 * n + 64
        out = torch.zeros(B, cap, dtype=torch.uint8, device=dev)
        off = torch.zeros(B, dtype=torch.int32, device=dev)
        ln = torch.zeros(B, dtype=torch.int32, device=dev)
        err = torch.zeros(B, dtype=torch.int32, device=dev)
        call("sgic_rans_encode_batch", T.handles[g], d_sym, d_idx, B, n, out, cap, off, ln, err)
        assert int(err.abs().sum()) == 0
        h_out, h_off, h_len = out.cpu().numpy(), off.cpu().numpy(), ln.cpu().numpy()
        for b in range(B):
            got = h_out[b, h_off[b]:h_off[b] + h_len[b]].tobytes()
            assert got == orc.rans_encode(sym[b], idx[b], otab), (B, n, b)
        # decode: 4 calls of n/4 each, continuing the cursor (like the 4 prior steps)
        state = torch.zeros(B, 4, dtype=torch.int32, device=dev)
        call("sgic_rans_decode_init_batch", out, cap, off, ln, B, state)
        dec = torch.zeros(B, n, dtype=torch.int16, device=dev)
        q = n // 4
        for k in range(4):
            a = k * q
            cnt = q if k < 3 else n - a
            call("sgic_rans_decode_batch", T.handles[g], out, cap, off, ln, B, state, d_idx.view(-1)[a:],
                 cnt, n, dec.view(-1)[a:], n)
        assert int(state[:, 2].abs().sum()) == 0
        assert np.array_equal(dec.cpu().numpy(), np.where(idx < 0, 0, sym))


def test_pack12_batch_vs_oracle():
    import sgic_amd  # noqa
    from sgic_amd._lib import call, lib
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(1)
    for (B, n) in [(32, 32), (16, 128), (5, 33), (1, 1)]:
        idx = rng.integers(0, 4096, size=(B, n)).astype(np.int32)
        nb = lib.sgic_pack12_size(n)
        d = torch.from_numpy(idx).to(dev)
        out = torch.zeros(B, nb, dtype=torch.uint8, device=dev)
        call("sgic_pack12_batch", d, B, n, out)
        h = out.cpu().numpy()
        for b in range(B):
            assert h[b].tobytes() == orc.pack12(idx[b].astype(np.int16)), (B, n, b)
        back = torch.zeros(B, n, dtype=torch.int32, device=dev)
        call("sgic_unpack12_batch", out, B, n, back)
        assert np.array_equal(back.cpu().numpy(), idx)


def test_quant_step_vs_oracle_bit_exact():
    """Fused masked quantise + index build (K9): NHWC HIP kernel vs the NCHW C oracle, same fp32 inputs."""
    import sgic_amd  # noqa
    from sgic_amd._lib import call
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(11)
    for (B, H, W) in [(4, 8, 8), (2, 16, 16), (1, 3, 5)]:
        C = 64
        y = (rng.standard_normal((B, C, H, W)) * 3).astype(np.float32)
        y[0, :, 0, 0] = np.float32(0.5) + np.arange(C, dtype=np.float32)  # exact ties -> round-half-even
        sc = np.exp(rng.uniform(np.log(0.05), np.log(70.0), size=(B, C, H, W))).astype(np.float32)
        mu = rng.standard_normal((B, C, H, W)).astype(np.float32)
        mu[0, :, 0, 0] = 0
        nhwc = lambda a: np.ascontiguousarray(a.transpose(0, 2, 3, 1))
        d_y = torch.from_numpy(nhwc(y)).to(dev).view(-1, C)
        sm = torch.from_numpy(np.concatenate([nhwc(sc), nhwc(mu)], axis=-1)).to(dev).view(-1, 2 * C)
        yhat = torch.zeros(B * H * W, 2 * C, device=dev)
        sym = torch.zeros(B, 4, C // 4, H, W, dtype=torch.int16, device=dev)
        idx = torch.zeros_like(sym)
        for k in range(4):
            call("sgic_quant_step", d_y, sm, sm.view(-1)[C:], 2 * C, yhat, 2 * C, B, H, W, C, k, 0.12, sym, idx)
        h_sym, h_idx = sym.cpu().numpy(), idx.cpu().numpy()
        h_yhat = yhat.cpu().numpy()[:, :C].reshape(B, H, W, C).transpose(0, 3, 1, 2)
        for b in range(B):
            ref_hat = np.zeros((C, H, W), np.float32)
            for k in range(4):
                s, i = orc.quant_step(y[b], sc[b], mu[b], k, 0.12, ref_hat)
                assert np.array_equal(h_sym[b, k], s), (B, H, W, b, k)
                assert np.array_equal(h_idx[b, k], i), (B, H, W, b, k)
            assert np.array_equal(h_yhat[b], ref_hat)


def test_reference_named_entropy_api(tab):
    """the L3 mirrors (entropy.entropy_models.EntropyCoder / GaussianEncoder) behave like the reference's:
    encode(x, scales, thr) x4 -> flush -> bytes == C oracle; decode_stream continues the cursor"""
    import sgic_amd  # noqa
    from sgic_amd.entropy.entropy_models import EntropyCoder, GaussianEncoder
    ec = EntropyCoder(False, 1)
    ge = GaussianEncoder("gaussian")
    ge.update(force=True, entropy_coder=ec)
    rng = np.random.default_rng(4)
    scales = [torch.from_numpy(np.exp(rng.uniform(np.log(0.05), np.log(40), size=(1, 16, 8, 8))).astype(np.float32)).cuda() for _ in range(4)]
    xs = [torch.from_numpy(np.rint(rng.standard_normal((1, 16, 8, 8)) * 3).astype(np.float32)).cuda() for _ in range(4)]
    ec.reset()
    for x, s in zip(xs, scales):
        ge.encode(x, s, skip_thres=0.12)
    ec.flush()
    stream = ec.get_encoded_stream()
    # oracle: same index formula in numpy float32 + C coder
    log_min, log_step = np.float32(np.log(0.11)), np.float32((np.log(64.0) - np.log(0.11)) / 255)
    syms, idxs = [], []
    for x, s in zip(xs, scales):
        sc = np.maximum(s.cpu().numpy().reshape(-1), np.float32(1e-5))
        fi = np.clip((np.log(sc.astype(np.float64)).astype(np.float32) - log_min) / log_step, 0, 255).astype(np.int32)
        fi[sc < np.float32(0.12)] = -1
        idxs.append(fi.astype(np.int16))
        syms.append(x.cpu().numpy().reshape(-1).astype(np.int16))
    assert stream == orc.rans_encode(np.concatenate(syms), np.concatenate(idxs), orc.Table(*tab))
    ec.set_stream(stream)
    for x, s, i in zip(xs, scales, idxs):
        got = ge.decode_stream(s, torch.float32, "cpu", skip_thres=0.12)
        assert np.array_equal(got.numpy().reshape(-1), np.where(i < 0, 0, x.cpu().numpy().reshape(-1)))


def test_reference_named_codec_plugin():
    """`target: models.codec_sq_fixbpp.Codec` constructed from the reference's YAML kwargs (config_test.yaml)"""
    import sgic_amd  # noqa
    from sgic_amd.config import SMALL
    from sgic_amd.data import synth_images
    from sgic_amd.entropy.compression_model import get_padding_size
    from sgic_amd.models.codec_sq_fixbpp import Codec
    params = dict(embed_dim=64, feat_dim=256, in_pos_enc=[1, 5], in_pos_dec=[1, 5], n_attn=2, config=SMALL.titok_dict(),
                  vqganconfig=dict(embed_dim=64, n_embed=64, ddconfig=SMALL.vqgan_ddconfig(), lossconfig=None),
                  imglossconfig={}, featlossconfig={}, training_strategy={"learning_rate": 5e-5}, monitor="saved_loss",
                  ckpt_path=None, ignore_keys=['epoch_for_strategy', 'lmbda_idx', 'lmbda_list'], titok_pretrain_path=None,
                  tune_titok=False, no_attn_vqgan=False)
    model = Codec(**params).to("cuda:0").eval()
    model.hybrid_codec.quantize_feat.force_zero_thres = 0.12
    model.hybrid_codec.quantize_feat.update(force=True)
    assert model.cfg == SMALL
    img = synth_images(1, 256, 256, 7)[:, :, :200, :231]
    pl, pr, pt, pb = get_padding_size(200, 231, p=256)
    xp = torch.nn.functional.pad(img, (pl, pr, pt, pb), mode="replicate")
    enc = model.encode_only(xp.cuda())
    assert set(enc) == {"z_bit_stream", "h_bit_stream", "img_shape", "feat_shape", "stack_shape", "token_length", "z_indices_shape"}
    x_hat = model.decode_only(**enc, clip_stream=b"", clip_meta={})
    assert x_hat.shape == (1, 3, 256, 256)


def test_error_paths_enospc_and_bad_index(tab):
    """slot too small -> SGIC_ENOSPC (-3) per image, index >= rows -> SGIC_EINVAL (-1); the facade retries / raises"""
    import sgic_amd  # noqa
    from sgic_amd._lib import call
    from sgic_amd.entropy.MLCodec_rans import RansEncoder, _Tables
    T = _Tables()
    g = T.add(*tab)
    dev = torch.device("cuda:0")
    n = 2048
    rng = np.random.default_rng(9)
    idx = rng.integers(0, 256, size=(2, n)).astype(np.int16)
    sym = rng.integers(-20000, 20000, size=(2, n)).astype(np.int16)      # bypass-heavy: ~10 bytes per symbol
    sym[1] = 0
    idx[1] = 0
    d_sym, d_idx = torch.from_numpy(sym).to(dev), torch.from_numpy(idx).to(dev)
    cap = 2 * n + 64
    out = torch.zeros(2, cap, dtype=torch.uint8, device=dev)
    meta = torch.zeros(3, 2, dtype=torch.int32, device=dev)
    call("sgic_rans_encode_batch", T.handles[g], d_sym, d_idx, 2, n, out, cap, meta[0], meta[1], meta[2])
    m = meta.cpu().numpy()
    assert m[2, 0] == -3 and m[1, 0] == 0          # image 0 did not fit
    assert m[2, 1] == 0 and m[1, 1] > 0            # image 1 is fine
    # the facade retries with the hard upper bound and matches the oracle
    enc = RansEncoder(False, 1)
    enc.add_cdf(*tab)
    enc.encode_with_indexes(sym[0], idx[0], 0)
    enc.flush()
    assert enc.get_encoded_stream().tobytes() == orc.rans_encode(sym[0], idx[0], orc.Table(*tab))
    # bad index
    idx_bad = idx.copy()
    idx_bad[1, 7] = 300
    call("sgic_rans_encode_batch", T.handles[g], d_sym, torch.from_numpy(idx_bad).to(dev), 2, n, out, 16 * n + 64 if False else cap,
         meta[0], meta[1], meta[2])
    assert meta.cpu().numpy()[2, 1] == -1
    enc.reset()
    with pytest.raises(IndexError):
        enc.encode_with_indexes(sym[1], idx_bad[1], 0)
        enc.flush()


def test_torchac_shim_vs_reference_sample_and_oracle(golden_dir):
    """seam B2 (models/codec_sq_fixbpp.py:864,887): sgic_amd.torchac.{encode,decode}_float_cdf with the codec's uniform
    cdf reproduce the z_bit_stream of the reference's own apple.c2df byte for byte, agree with the C restatement of
    torchac's coder (oracle), round-trip, and refuse any other cdf."""
    import sgic_amd  # noqa
    import sgic_amd.torchac as torchac
    from sgic_amd.filemaker import unpack_c2df
    enc, _ = unpack_c2df(os.path.join(golden_dir, "ref_apple.c2df"))
    z, n = enc["z_bit_stream"], int(enc["token_length"])
    cdf = torch.zeros(4097)
    cdf[1:] = torch.cumsum(torch.ones(4096) / 4096, dim=0)                 # Codec.set_torchac (:841-846)
    shaped = cdf.unsqueeze(0).repeat(n, 1)
    idx = torchac.decode_float_cdf(shaped, z)
    assert idx.dtype == torch.int16 and idx.shape == (n,) and np.array_equal(idx.numpy(), orc.unpack12(z, n))
    assert torchac.encode_float_cdf(shaped, idx) == z
    rng = np.random.default_rng(5)
    for m in (1, 2, 31, 32, 33, 128):
        s = torch.from_numpy(rng.integers(0, 4096, size=m).astype(np.int16))
        c = cdf.unsqueeze(0).repeat(m, 1)
        b = torchac.encode_float_cdf(c, s)
        assert b == orc.torchac_uniform_encode(s.numpy()) and torch.equal(torchac.decode_float_cdf(c, b), s)
    with pytest.raises(NotImplementedError):
        torchac.encode_float_cdf(torch.linspace(0, 1, 11).unsqueeze(0), torch.zeros(1, dtype=torch.int16))
    with pytest.raises(NotImplementedError):
        torchac.encode_float_cdf((cdf ** 2).unsqueeze(0), torch.zeros(1, dtype=torch.int16))
    with pytest.raises(ValueError):
        torchac.decode_float_cdf(shaped, z[:-1])


def test_overflowing_image_is_reencoded_alone_not_the_batch(tab):
    """batched path (pipeline.finish / encode_batch -> bottleneck.slice_streams): one bypass-heavy image overflows its
    2n+64 slot (SGIC_ENOSPC); it is re-encoded on its own with the 16n+64 bound and the other images of the batch keep
    their streams -- all equal to the oracle's.  A bad index (EINVAL) still raises."""
    import sgic_amd  # noqa
    from sgic_amd import ops
    from sgic_amd.bottleneck import SGIC_ENOSPC, slice_streams
    from sgic_amd.entropy.MLCodec_rans import _Tables
    T = _Tables()
    gid = T.add(*tab)
    otab = orc.Table(*tab)
    rng = np.random.default_rng(5)
    B, n = 3, 512
    sym = rng.integers(-3, 4, (B, n)).astype(np.int16)
    idx = rng.integers(0, 256, (B, n)).astype(np.int16)
    sym[1] = rng.choice(np.array([-30000, 30000], dtype=np.int16), n)        # every symbol of image 1 takes the bypass path
    ds, di = torch.from_numpy(sym).cuda(), torch.from_numpy(idx).cuda()
    out, meta = ops.rans_encode_batch(T.handles[gid], ds, di, B, n)
    m = meta.cpu().numpy()
    assert m[2].tolist() == [0, SGIC_ENOSPC, 0]
    with pytest.raises(RuntimeError):
        slice_streams(out.cpu().numpy(), m)                                    # without the device data there is nothing to retry from
    got = slice_streams(out.cpu().numpy(), m, retry=(T.handles[gid], ds, di, n))
    for b in range(B):
        assert got[b] == orc.rans_encode(sym[b], idx[b], otab), b
    assert len(got[1]) > 2 * n + 64
    idx[2, 7] = 300                                                            # outside the 256-row table
    out, meta = ops.rans_encode_batch(T.handles[gid], ds, torch.from_numpy(idx).cuda(), B, n)
    with pytest.raises(RuntimeError):
        slice_streams(out.cpu().numpy(), meta.cpu().numpy(), retry=(T.handles[gid], ds, torch.from_numpy(idx).cuda(), n))
