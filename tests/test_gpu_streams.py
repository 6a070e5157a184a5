"""Stream-level parity against the REFERENCE on a wide fixture (tests/golden/streams_small.npz, made by
oracle/gen_golden_streams.py from the imported reference modules + the reference C++ coder): 32 images 256x256 and one
image at the reference's worked geometry (859x1000 -> 1024x1024, 16 tiles), SMALL architecture, B = 1 reference calls.

Encode side: the HIP path recomputes symbols / indexes / h_bit_stream from the same inputs; fp32 summation order differs
from torch-CPU's, so a sigma that sits on a bin edge can flip an index (the reference has the same sensitivity between
BLAS back ends, SURVEY 7 hard part 1).  The test REPORTS streams_identical / total and the flip counts and bounds the
flip rate.  Decode side: a single wrong index while decoding a reference-made stream desynchronises rANS for the rest of
the image, so there the bar is absolute: EVERY reference stream must decode with no error flag, consume exactly its
bytes, and give exactly the reference's symbols and its y_hat.  The decoder's own indexes are compared too and their
flips COUNTED: a flip between two table rows whose entry for the coded symbol coincides is harmless (the symbols still
come out right -- that is what is asserted); any harmful one fails the symbol check."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(golden_dir):
    import sgic_amd  # noqa
    from sgic_amd import weights as W
    from sgic_amd.codec import Codec
    from sgic_amd.config import SMALL
    g = np.load(os.path.join(golden_dir, "streams_small.npz"))
    sd = W.synth_weights(W.encoder_spec(SMALL) + W.codec_misc_spec(SMALL) + W.bottleneck_spec(SMALL), seed=1234)
    codec = Codec(sd, SMALL, "cuda:0")
    codec.hybrid_codec.quantize_feat.force_zero_thres = 0.12
    codec.hybrid_codec.quantize_feat.update(force=True)
    return g, codec, SMALL


def _input(H, W, seed):
    from sgic_amd import ops
    from sgic_amd.data import synth_images
    from sgic_amd.entropy.compression_model import get_padding_size
    Hs, Ws = 256 * ((H + 255) // 256), 256 * ((W + 255) // 256)
    img = synth_images(1, Hs, Ws, seed)[:, :, :H, :W].contiguous().cuda()
    return ops.pad_replicate(img, *get_padding_size(H, W, p=256))


def _groups(g):
    """cases of equal padded geometry run as one batch (the HIP path is batch-invariant)"""
    by = {}
    for name, (H, W, seed) in zip(g["names"], g["geometry"]):
        H, W, seed = int(H), int(W), int(seed)
        by.setdefault((256 * ((H + 255) // 256), 256 * ((W + 255) // 256)), []).append((str(name), H, W, seed))
    return by


def _coded_sigma(taps, B, hh, ww, Q):
    """our sigma at the coded positions, arranged like the reference's scales_w_k (B, 4, Q/4, h, w): at step k and position
    (i, j) the active quarter is p ^ {0,3,2,1}[k], p = 2 (i & 1) + (j & 1) (compression_model.py:277-280)"""
    q4 = Q // 4
    out = np.zeros((B, 4, q4, hh, ww), dtype=np.float32)
    ii, jj = np.meshgrid(np.arange(hh), np.arange(ww), indexing="ij")
    p = 2 * (ii & 1) + (jj & 1)
    for k, t in enumerate(taps):
        s = t.cpu().numpy().reshape(B, hh, ww, Q)
        quarter = p ^ [0, 3, 2, 1][k]
        for c in range(q4):
            out[:, k, c] = np.take_along_axis(s, (quarter * q4 + c)[None, :, :, None], axis=3)[..., 0]
    return out


def test_encode_side_streams_vs_reference(setup, golden_dir):
    g, codec, cfg = setup
    sig_path = os.path.join(golden_dir, "streams_small_sigma.npz")
    gsig = np.load(sig_path) if os.path.exists(sig_path) else None
    total = ident = sym_flips = idx_flips = n_sym = 0
    vq_mismatch = 0
    per_image = {}
    causes = {"same_sigma_different_index": 0, "sigma_differs": 0, "details": []}
    sig_stats = {"positions": 0, "bitwise_equal": 0, "max_ulps": 0}
    for (Hp, Wp), cases in _groups(g).items():
        x = torch.cat([_input(H, W, seed) for _, H, W, seed in cases])
        encs = codec.encode_batch(x)
        r = codec.encode_device(x)
        sym, idx, vq = r["sym"].cpu().numpy(), r["idx"].cpu().numpy(), r["vq"].cpu().numpy().reshape(len(cases), -1)
        taps = []
        bn = codec.bottleneck
        hh, ww = r["feat_hw"]
        bn.quantise(bn.analysis(r["h"], len(cases), hh, ww), len(cases), hh, ww, sigma_taps=taps)
        ours = _coded_sigma(taps, len(cases), hh, ww, bn.Q)
        for b, (name, H, W, seed) in enumerate(cases):
            gs, gi = g[f"{name}.sym"], g[f"{name}.idx"]
            sf, jf = int((sym[b] != gs).sum()), int((idx[b] != gi).sum())
            if gsig is not None:
                # attribution of every index flip: the reference's sigma at the same position, as bits
                ref_sig = gsig[f"{name}.sigma"]
                coded = gi >= 0
                a, bb = ours[b][coded].view(np.int32).astype(np.int64), ref_sig[coded].view(np.int32).astype(np.int64)
                sig_stats["positions"] += int(coded.sum())
                sig_stats["bitwise_equal"] += int((a == bb).sum())
                sig_stats["max_ulps"] = max(sig_stats["max_ulps"], int(np.abs(a - bb).max()) if a.size else 0)
                for pos in zip(*np.nonzero(idx[b] != gi)):
                    so, sr = float(ours[b][pos]), float(ref_sig[pos])
                    same = np.float32(so).view(np.int32) == np.float32(sr).view(np.int32)
                    causes["same_sigma_different_index" if same else "sigma_differs"] += 1
                    causes["details"].append({"image": name, "pos": [int(v) for v in pos], "sigma_ours": so.hex(), "sigma_ref": sr.hex(),
                                              "idx_ours": int(idx[b][pos]), "idx_ref": int(gi[pos])})
            same = encs[b]["h_bit_stream"] == g[f"{name}.stream"].tobytes()
            # identical symbols + indexes must give the reference's bytes; the converse does not hold (an index flip between
            # two table rows whose cdf entry for that symbol coincides leaves the stream unchanged)
            assert same or (sf + jf) > 0, f"{name}: coder disagrees with the reference on identical symbols/indexes"
            vq_mismatch += int((vq[b] != g[f"{name}.vq"].astype(np.int64)).sum())
            total, ident, sym_flips, idx_flips, n_sym = total + 1, ident + int(same), sym_flips + sf, idx_flips + jf, n_sym + gs.size
            per_image[name] = {"sym_flips": sf, "idx_flips": jf, "stream_identical": bool(same)}
            assert sf == 0 and jf <= 2, f"{name}: {sf} symbol / {jf} index flips of {gs.size}"
    report = {"streams_identical": ident, "total": total, "symbol_flips": sym_flips, "index_flips": idx_flips, "symbols": n_sym,
              "vq_index_mismatches": vq_mismatch, "index_flip_causes": causes, "sigma_vs_reference": sig_stats, "per_image": per_image}
    print("\n[stream parity, encode side] " + json.dumps({k: v for k, v in report.items() if k != "per_image"}))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        json.dump(report, open(os.path.join(out, "stream_parity_encode.json"), "w"), indent=1)
    # The bar is the claimed level with one chance event of headroom (round 2 measured 32 / 33 identical, 0 symbol flips, 2 index
    # flips in 196 608, 0 VQ mismatches; the kernels are bitwise independent of tile choice and batch, so a box does not change it):
    assert total == 33 and vq_mismatch == 0          # every TiTok token equals the reference's
    assert sym_flips == 0 and idx_flips <= 6         # <= 3e-5 of the coded positions (a sigma within fp32 noise of a bin edge)
    assert ident >= 31                               # streams byte-identical to the reference's, end to end


def test_every_reference_stream_decodes(setup):
    """decode side: every reference-made stream must come out as the reference's y_hat.  The plain decode is verified by
    the rANS end-of-stream condition (stream_status); where fp32 noise flipped a near-boundary index -- which
    desynchronises the rest of that stream -- the verified retry (_retry_edge_flips) must repair it."""
    g, codec, cfg = setup
    bn = codec.bottleneck
    decoded = idx_flips = repaired = 0
    report = {}
    for (Hp, Wp), cases in _groups(g).items():
        hh, ww, B = Hp // 32, Wp // 32, len(cases)
        streams = [g[f"{name}.stream"].tobytes() for name, *_ in cases]
        cap = max(len(s) for s in streams)
        buf = np.zeros((B, cap), dtype=np.uint8)
        for b, s in enumerate(streams):
            buf[b, :len(s)] = np.frombuffer(s, dtype=np.uint8)
        ln = torch.tensor([len(s) for s in streams], dtype=torch.int32, device="cuda:0")
        y_hat, state, sym, idx = bn.decode_latent(torch.from_numpy(buf).cuda(), None, ln, cap, B, hh, ww)
        status = bn.stream_status(state.cpu().numpy(), [len(s) for s in streams])
        sym, idx = sym.cpu().numpy(), idx.cpu().numpy()
        for b, (name, *_) in enumerate(cases):
            gi, gs, ref = g[f"{name}.idx"], g[f"{name}.sym"].copy(), g[f"{name}.y_hat"]
            gs[gi < 0] = 0                                   # skipped positions decode to 0 (rans.cpp:317-320)
            flips = int((idx[b] != gi).sum())
            tol = 2e-5 * max(1.0, float(np.abs(ref).max()))
            if status[b] == 0:
                # clean end of stream => the symbols are the reference's (index flips, if any, were between rows whose
                # entry for the coded symbol coincides: harmless)
                assert np.array_equal(sym[b], gs), f"{name}: end-of-stream check passed but symbols differ"
                y = y_hat.view(B, hh, ww, -1)[b].permute(2, 0, 1).cpu().numpy()
                idx_flips += flips
            else:
                assert not np.array_equal(sym[b], gs), f"{name}: end-of-stream check failed on a correct decode"
                # where does the genuine flip rank among the near-boundary candidates?
                _, _, _, idx1, marg1, alt1 = bn.decode_latent(torch.from_numpy(buf[b:b + 1, :len(streams[b])].copy()).cuda(), None, ln[b:b + 1],
                                                               len(streams[b]), 1, hh, ww, margins=True)
                m1, i1 = marg1.cpu().numpy()[0], idx1.cpu().numpy()[0]
                for k in range(4):
                    bad = np.nonzero((i1[k] != gi[k]).reshape(-1))[0]
                    if len(bad):
                        mk = m1[k].reshape(-1)
                        print(f"[diag] {name}: first flipped step {k}: positions {bad[:4].tolist()} margins {mk[bad[:4]].tolist()} "
                              f"rank among all margins {[int((m1.reshape(-1) < mk[p]).sum()) for p in bad[:4]]}; alt {alt1.cpu().numpy()[0, k].reshape(-1)[bad[:4]].tolist()} "
                              f"ref {gi[k].reshape(-1)[bad[:4]].tolist()}")
                        break
                yb, tries = bn._retry_edge_flips(streams[b], hh, ww)
                y = yb.view(hh, ww, -1).permute(2, 0, 1).cpu().numpy()
                repaired += 1
                report[name] = {"plain_decode_status": status[b], "index_flips_seen": flips, "verified_attempts": tries}
            assert np.abs(y - ref).max() <= tol, f"{name}: y_hat differs from the reference's"
            decoded += 1
    assert decoded == len(g["names"])
    print(f"\n[stream parity, decode side] {decoded}/{decoded} reference streams reproduce the reference's y_hat; "
          f"{repaired} needed the verified near-boundary retry {report}; {idx_flips} harmless index flips")
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        json.dump({"streams_decoded": decoded, "total": int(len(g["names"])), "repaired_by_retry": report, "harmless_index_flips": idx_flips},
                  open(os.path.join(out, "stream_parity_decode.json"), "w"))
    # a flip is a chance event (a sigma within ~1e-7 relative of a bin edge, where two fp32 implementations disagree): about 2 are
    # expected over the 196 608 coded positions of this fixture (observed: 1 with SGIC_GEMM=f32, 4 with the default split GEMM, at
    # identical error levels against the reference's activations; DESIGN section 4 on why the rates differ); every one must be REPAIRED
    assert decoded == 33 and repaired <= 3 and idx_flips <= 6


def test_decompress_repairs_and_rejects(setup):
    """the product entry (BottleneckHIP.decompress): a foreign-numerics stream is repaired transparently; a corrupted
    stream (bit flipped mid-stream) is REJECTED loudly instead of decoding to garbage like the reference would"""
    g, codec, cfg = setup
    bn = codec.bottleneck
    name = "apple_geometry"
    s = g[f"{name}.stream"].tobytes()
    h = bn.decompress([s], 1, 32, 32)
    assert tuple(h.shape) == (32 * 32, cfg.feat_dim) and bool(torch.isfinite(h).all())
    bad = bytearray(g["s03.stream"].tobytes())
    bad[len(bad) // 2] ^= 0x10
    with pytest.raises(RuntimeError):
        bn.decompress([bytes(bad)], 1, 8, 8)
    ok = bn.decompress([g["s03.stream"].tobytes(), g["s04.stream"].tobytes()], 2, 8, 8)
    assert bn.last_repairs == 0 and tuple(ok.shape) == (2 * 64, cfg.feat_dim)


def test_apple_geometry_container_round_trip(setup, tmp_path):
    """the reference's one worked example geometry end to end through the container: 859x1000 -> 16 tiles -> .c2df ->
    decode_only -> crop; header fields as compress.py:272-279 writes them"""
    g, codec, cfg = setup
    from sgic_amd.filemaker import pack_c2df, unpack_c2df
    i = list(g["names"]).index("apple_geometry")
    H, W, seed = (int(v) for v in g["geometry"][i])
    x = _input(H, W, seed)
    assert tuple(x.shape) == (1, 3, 1024, 1024)
    enc = codec.encode_only(x)
    assert enc["stack_shape"] == (4, 4) and enc["token_length"] == 32 * 16 and len(enc["z_bit_stream"]) == 769   # apple.c2df: 769 B
    blob = pack_c2df(dict(enc, clip_stream=b"", clip_meta={}), {"version": 2, "image_hw": [H, W], "padding": [0, 24, 0, 165]})
    e2, hdr = unpack_c2df(blob)
    assert e2["h_bit_stream"] == enc["h_bit_stream"] and hdr["padding"] == [0, 24, 0, 165]
