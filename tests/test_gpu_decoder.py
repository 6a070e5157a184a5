"""GPU parity of the decode path (D1-D5) against golden outputs of the REAL reference modules
(tests/golden/dec_small_*.npz, oracle/gen_golden_dec.py) and round-trip properties.
Tolerance 2e-5 * max|ref| (measured 1e-6 ... 2.5e-6) for float tensors (the VQGAN stack amplifies fp32 ordering noise a little more than
the encoder: 30+ GroupNorm/conv layers), pixels additionally as PSNR vs the reference reconstruction."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope="module")
def codec():
    import sgic_amd  # noqa
    from sgic_amd import weights as W
    from sgic_amd.codec import Codec
    from sgic_amd.config import SMALL
    sd = W.synth_weights(W.full_spec(SMALL), seed=1234)
    c = Codec(sd, SMALL, "cuda:0")
    c.hybrid_codec.quantize_feat.force_zero_thres = 0.12
    c.hybrid_codec.quantize_feat.update(force=True)
    return c


def _rel(a, b):
    return float((a - b).abs().max() / max(1e-6, float(b.abs().max())))


def _tm16_to_nchw(t, B, Hf, Wf):
    C = t.shape[1]
    nH, nW = Hf // 16, Wf // 16
    return t.reshape(B, nH, nW, 16, 16, C).permute(0, 5, 1, 3, 2, 4).reshape(B, C, Hf, Wf)


def _enc_results(g, B, H, W, cfg):
    nH, nW, T = H // 256, W // 256, cfg.num_latent_tokens
    from oracle import orc
    res = []
    for b in range(B):
        idx = g["vq_idx"][b * nH * nW * T:(b + 1) * nH * nW * T].astype(np.int16)
        res.append(dict(z_bit_stream=orc.pack12(idx), h_bit_stream=g[f"stream_{b}"].tobytes(), img_shape=(H, W),
                        feat_shape=(1, cfg.feat_dim, H // 32, W // 32), stack_shape=(nH, nW), token_length=nH * nW * T,
                        z_indices_shape=(nH * nW, cfg.token_size, 1, T)))
    return res


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_decode_vs_reference_golden(case, codec, golden_dir):
    cfg = codec.cfg
    g = np.load(os.path.join(golden_dir, f"nn_small_{case}.npz"))
    d = np.load(os.path.join(golden_dir, f"dec_small_{case}.npz"))
    B, H, W = int(g["B"]), int(g["H"]), int(g["W"])
    Hf, Wf = H // 16, W // 16
    taps = {}
    x_hat = codec.decode_batch(_enc_results(g, B, H, W, cfg), taps=taps).cpu()
    # z_hat (N,12,1,32), h_hat (B,F,hh,ww): exact-ish (integer decode + a few layers)
    N, T = d["z_hat"].shape[0], cfg.num_latent_tokens
    z = taps["z_rows"].cpu().reshape(N, T, -1).permute(0, 2, 1).reshape(N, -1, 1, T)
    h = taps["h_hat"].cpu().reshape(B, H // 32, W // 32, -1).permute(0, 3, 1, 2)
    e = dict(z_hat=_rel(z, torch.from_numpy(d["z_hat"])), h_hat=_rel(h, torch.from_numpy(d["h_hat"])))
    titok = _tm16_to_nchw(taps["titok"].cpu(), B, Hf, Wf)
    feat = _tm16_to_nchw(taps["feat"].cpu(), B, Hf, Wf)
    latent = _tm16_to_nchw(taps["latent"].cpu(), B, Hf, Wf)
    if case == "a":
        e.update(titok=_rel(titok, torch.from_numpy(d["titok"])), feat=_rel(feat, torch.from_numpy(d["feat"])),
                 logits=_rel(_tm16_to_nchw(taps["logits"].cpu(), B, Hf, Wf), torch.from_numpy(d["logits"])),
                 latent=_rel(latent, torch.from_numpy(d["latent"])))
        x_ref, x_cmp = torch.from_numpy(d["x_hat"]), x_hat
    else:
        e.update(titok=_rel(titok[:, :, ::2, ::2], torch.from_numpy(d["titok_s2"])),
                 feat=_rel(feat[:, :, ::2, ::2], torch.from_numpy(d["feat_s2"])),
                 latent=_rel(latent[:, :, ::2, ::2], torch.from_numpy(d["latent_s2"])))
        x_ref, x_cmp = torch.from_numpy(d["x_hat_s4"]), x_hat[:, :, ::4, ::4]
    e["x_hat"] = float((x_cmp - x_ref).abs().max())
    mse = float(((x_cmp - x_ref) ** 2).mean())
    psnr = 10 * np.log10(4.0 / max(mse, 1e-20))
    print(f"case {case}: rel err {{{', '.join(f'{k} {v:.1e}' for k, v in e.items())}}}  PSNR vs reference recon {psnr:.1f} dB")
    assert all(v < TOL for k, v in e.items() if k != "x_hat"), e
    assert e["x_hat"] < 2e-4 and psnr > 100.0        # measured: max |dx| 4e-5, PSNR 110-112 dB vs the reference's reconstruction
    assert x_hat.shape == (B, 3, H, W) and float(x_hat.abs().max()) <= 1.0


def test_entropy_roundtrip_encode_then_decode_is_lossless(codec):
    """encoder-side y_hat == decoder-side y_hat bitwise (the reference's own self-check,
    models/sq_bottleneck.py:202-216): compress a batch, decompress its streams, compare the latents."""
    from sgic_amd.data import synth_images
    from sgic_amd import ops
    x = synth_images(4, 256, 256, 77).cuda()
    _, h, _ = codec.encoder.forward(x)
    b = codec.bottleneck
    y = b.analysis(h, 4, 8, 8)
    sym, idx, ctx, paramsB = b.quantise(y, 4, 8, 8)
    n = sym[0].numel()
    out, meta = ops.rans_encode_batch(b.tables.handles[b.group], sym, idx, 4, n)
    y_hat_enc = ops.colop(ctx[:, 0:b.Q], paramsB[:, 0:b.Q], 2)
    m = meta.clone()
    y_hat_dec, state, sym_d, idx_d = b.decode_latent(out, m[0].contiguous(), m[1].contiguous(), out.shape[1], 4, 8, 8)
    assert int(state[:, 2].abs().sum()) == 0
    assert torch.equal(idx_d, idx)
    assert torch.equal(sym_d, torch.where(idx < 0, torch.zeros_like(sym), sym))
    assert torch.equal(y_hat_dec, y_hat_enc)


def test_full_roundtrip_compress_decompress(codec):
    """encode_batch -> .c2df bytes -> unpack -> decode_batch: container + both coders + generator end to end"""
    from sgic_amd.data import synth_images
    from sgic_amd.filemaker import pack_c2df, unpack_c2df
    x = synth_images(2, 256, 512, 5).cuda()
    encs = codec.encode_batch(x)
    blobs = [pack_c2df(dict(e, clip_stream=b"", clip_meta={}), {"version": 2}) for e in encs]
    back = [unpack_c2df(b)[0] for b in blobs]
    x1 = codec.decode_batch(back)
    x2 = codec.decode_batch(encs)
    assert x1.shape == (2, 3, 256, 512) and torch.equal(x1, x2)
    one = codec.decode_only(**{k: v for k, v in back[1].items()})
    assert torch.equal(one[0], x1[1])   # batch-invariant: B=1 decode == row of the B=2 decode
    assert torch.isfinite(x1).all()


@pytest.mark.parametrize("B,H,W,Cin,Cout,res", [(2, 24, 40, 128, 3, False),      # thin VALU kernel (taming conv_out 128 -> 3)
                                                  (2, 24, 40, 64, 128, True),      # implicit-GEMM path with residual
                                                  (1, 16, 16, 128, 4, False)])     # Cout = 4 stays on the GEMM path
@pytest.mark.parametrize("precision", ["f32", "split3"])
def test_conv3x3_vs_torch(B, H, W, Cin, Cout, res, precision):
    """sgic_conv3x3_f32 / sgic_conv3x3_split3_f32 (taming Conv2d 3x3 stride 1 pad 1, model.py:38-137,531-537) vs F.conv2d in fp64.
    Tolerance: 2e-5 * max|ref| (fp32 accumulation over K = 9*Cin <= 1152 terms)."""
    import sgic_amd  # noqa
    from sgic_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)
    b = torch.randn(Cout, generator=g)
    r = torch.randn(B, Cout, H, W, generator=g) if res else None
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1)
    if res:
        ref = ref + r.double()
    halo = torch.zeros(B, H + 2, W + 2, Cin, device="cuda:0")
    halo[:, 1:-1, 1:-1] = x.permute(0, 2, 3, 1).cuda()
    wk = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().cuda()            # (ky, kx, cin) order
    ld = 4 if Cout == 3 else Cout
    out = torch.zeros(B * H * W, ld, device="cuda:0")
    rr = r.permute(0, 2, 3, 1).reshape(B * H * W, Cout).contiguous().cuda() if res else None
    ops.conv3x3(halo, wk, b.cuda(), B, H, W, Cin, Cout, residual=rr, out=out[:, :Cout], precision=precision)
    got = out[:, :Cout].reshape(B, H, W, Cout).permute(0, 3, 1, 2).cpu().double()
    assert float((got - ref).abs().max()) < 2e-5 * float(ref.abs().max())
    if precision == "split3" and Cout != 3:      # every tile mode of the split kernel: bitwise the same convolution
        for tile in (1, 2, 3, 5, 10, 11, 14, 15, 16, 17, 30):
            o2 = torch.zeros_like(out)
            ops.conv3x3(halo, wk, b.cuda(), B, H, W, Cin, Cout, residual=rr, out=o2[:, :Cout], precision=precision, tile=tile)
            assert torch.equal(o2, out), tile
    if ld > Cout:
        assert float(out[:, Cout:].abs().max()) == 0.0      # the padding column of the output buffer is untouched
