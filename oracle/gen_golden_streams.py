"""Stream-level golden set from the REAL reference modules (imported read-only under stubs, see ref_harness.py; build
container only): 32 SMALL-architecture 256x256 images + one image at the reference's own worked geometry (859 x 1000 ->
replicate-padded to 1024 x 1024 = 16 tiles, README / IO/images/apple.jpg), each pushed through the reference's
compress.py loop body (compress.py:252-263 -> codec_sq_fixbpp.py:849-878 -> sq_bottleneck.py:159-182) at B = 1.

Per image the committed fixture holds DATA only: the reference's VQ indices, the four-step symbols / indexes, the
h_bit_stream bytes the reference C++ coder produced, and the reference's y_hat (what its decoder reconstructs from that
stream, compression_model.py:377-418).  The inputs are regenerated from the seeds (sgic_amd.data.synth_images).

Run:  make -C oracle && python oracle/gen_golden_streams.py      -> tests/golden/streams_small.npz
      python oracle/gen_golden_streams.py --large                -> tests/golden/streams_large.npz: two 256x256 images through
      the PRODUCTION architecture (TiTok ViT-L, 24 layers, 768-wide detail branch) of the real reference, with z and h
      kept as well, so the large model is pinned against the reference itself and not only against oracle/torch_ref.py
      python oracle/gen_golden_streams.py --sigma                -> tests/golden/streams_small_sigma.npz: the reference's sigma
      (scales_w_k, compression_model.py:317-353, as fp32 bit patterns) at every coded position of the same 33 images and
      nothing else; with it tests/test_gpu_streams.py attributes every index flip to its cause (same sigma, different index =
      the index arithmetic, entropy_models.py:355-362; different sigma = the arithmetic upstream).  Does not rewrite the main file.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_harness  # noqa: E402

wrap = ref_harness.setup()
import torch  # noqa: E402

import sgic_amd  # noqa: E402,F401
from sgic_amd import weights as W  # noqa: E402
from sgic_amd.config import LARGE, SMALL  # noqa: E402
from sgic_amd.data import synth_images  # noqa: E402

from entropy.compression_model import get_padding_size  # noqa: E402  (reference code)
from models.codec_sq_fixbpp import Hybrid_Codec  # noqa: E402  (reference code)

BIG = "--large" in sys.argv
SIGMA = "--sigma" in sys.argv and not BIG
sig_out = {}
cfg = LARGE if BIG else SMALL
torch.manual_seed(0)
torch.set_num_threads(8)
hc = Hybrid_Codec(wrap(cfg.titok_dict()), list(cfg.in_pos_enc), list(cfg.in_pos_dec), cfg.feat_dim, cfg.embed_dim, cfg.n_attn).eval()
sd = W.synth_weights(W.encoder_spec(cfg) + W.codec_misc_spec(cfg) + W.bottleneck_spec(cfg), seed=1234)
missing, unexpected = hc.load_state_dict({n[len("hybrid_codec."):]: t for n, t in sd.items()}, strict=False)
assert not unexpected and all(m.startswith("decoder.") for m in missing)
bn = hc.quantize_feat
bn.force_zero_thres = cfg.force_zero_thres
bn.update(force=True)

# (name, H, W, seed): 32 single-tile images, then the 16-tile worked geometry
cases = [(f"s{i:02d}", 256, 256, 200 + i) for i in range(32)] + [("apple_geometry", 859, 1000, 300)]
if BIG:
    cases = [("L0", 256, 256, 400), ("L1", 256, 256, 401)]
out = {"names": np.array([c[0] for c in cases]), "geometry": np.array([c[1:] for c in cases], dtype=np.int32)}
for name, H, Wd, seed in cases:
    Hs, Ws = 256 * ((H + 255) // 256), 256 * ((Wd + 255) // 256)
    img = synth_images(1, Hs, Ws, seed)[:, :, :H, :Wd].contiguous()                    # the "file": an H x W image in [-1,1]
    pl, pr, pt, pb = get_padding_size(H, Wd, p=256)
    x = torch.nn.functional.pad(img, (pl, pr, pt, pb), mode="replicate")               # compress.py:258-261
    x01 = x * 0.5 + 0.5                                                                # encode_only: codec_sq_fixbpp.py:855
    z, h, stack = hc.encoder(pixel_values=x01, latent_tokens=hc.latent_tokens)
    _, zres = hc.quantize(z)
    vq = zres["min_encoding_indices"].flatten()
    q_enc, q_dec, q_prior = bn.get_qp(0, h.shape)
    y = bn.encode(h, q_enc)
    r = bn.compress_four_part_prior(y, bn.y_prior_fusion(q_prior), bn.y_spatial_prior_adaptor_1, bn.y_spatial_prior_adaptor_2,
                                    bn.y_spatial_prior_adaptor_3, bn.y_spatial_prior,
                                    y_spatial_prior_reduction=bn.y_spatial_prior_reduction)
    sym = torch.stack([t.clamp(-30000, 30000).to(torch.int16)[0] for t in r[0:4]])
    idx = torch.stack([bn.gaussian_encoder.build_indexes(s, bn.force_zero_thres).to(torch.int16)[0] for s in r[4:8]])
    if SIGMA:
        sig_out[f"{name}.sigma"] = torch.stack([t[0] for t in r[4:8]]).numpy().astype(np.float32)
        print(f"{name}: sigma {sig_out[f'{name}.sigma'].shape}", flush=True)
        continue
    stream = bn.compress(h, 0)                                                         # the reference's own coder (oracle/_ref)
    y_hat_dec = None
    # the reference's decoder on its own stream: y_hat before the synthesis transform
    bn.entropy_coder.reset()
    bn.entropy_coder.set_stream(stream)
    y_hat_dec = bn.decompress_four_part_prior(bn.y_prior_fusion(q_prior), bn.y_spatial_prior_adaptor_1, bn.y_spatial_prior_adaptor_2,
                                              bn.y_spatial_prior_adaptor_3, bn.y_spatial_prior, bn.y_spatial_prior_reduction)
    assert torch.equal(y_hat_dec, r[8]), "reference encoder-side and decoder-side y_hat differ"
    h_hat = bn.decode(y_hat_dec, q_dec)
    out[f"{name}.vq"] = vq.numpy().astype(np.int16)
    out[f"{name}.sym"] = sym.numpy()
    out[f"{name}.idx"] = idx.numpy()
    out[f"{name}.stream"] = np.frombuffer(stream, dtype=np.uint8).copy()
    out[f"{name}.y_hat"] = y_hat_dec.numpy()[0]
    out[f"{name}.h_hat_absmax"] = np.float32(h_hat.abs().max())
    if BIG:
        out[f"{name}.z"], out[f"{name}.h"], out[f"{name}.y"] = z.numpy(), h.numpy(), y.numpy()
    print(f"{name}: {H}x{Wd} -> {tuple(x.shape[2:])} tiles {stack} stream {len(stream)} B coded {int((idx >= 0).sum())}/{idx.numel()} "
          f"|sym|max {int(sym.abs().max())}", flush=True)
if SIGMA:
    dst = os.path.join(ROOT, "tests", "golden", "streams_small_sigma.npz")
    np.savez_compressed(dst, **sig_out)
    print("wrote", dst, os.path.getsize(dst), "bytes")
    sys.exit(0)
dst = os.path.join(ROOT, "tests", "golden", "streams_large.npz" if BIG else "streams_small.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, os.path.getsize(dst), "bytes")
