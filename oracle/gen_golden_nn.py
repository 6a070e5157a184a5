"""Generate floating-point golden vectors from the REAL reference modules (imported read-only from
/root/reference under stubs, see ref_harness.py) with this repo's deterministic synthetic weights,
and check oracle/torch_ref.py (the restatement) against them on the spot.  Build-container only.

Committed output: tests/golden/nn_small_<case>.npz  (SMALL config: TiTok 'small' 8 layers width 512,
detail width 256, same topology as the production 'large' config) holding the reference's
z / h / VQ indices / 4-step symbols+indexes / h_bit_stream for seeded inputs.

Run:  make -C oracle && python oracle/gen_golden_nn.py
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
from sgic_amd.config import SMALL  # noqa: E402
from sgic_amd.data import synth_images  # noqa: E402
import torch_ref as TR  # noqa: E402

from models.codec_sq_fixbpp import Hybrid_Codec  # noqa: E402  (reference code)

OUT = os.path.join(ROOT, "tests", "golden")
cfg = SMALL
torch.manual_seed(0)
hc = Hybrid_Codec(wrap(cfg.titok_dict()), list(cfg.in_pos_enc), list(cfg.in_pos_dec), cfg.feat_dim, cfg.embed_dim,
                  cfg.n_attn).eval()

spec = W.encoder_spec(cfg) + W.codec_misc_spec(cfg) + W.bottleneck_spec(cfg)
sd = W.synth_weights(spec, seed=1234)
ref_sd = hc.state_dict()
# every key of our inventory exists in the reference with the same shape ...
for name, shape, _ in spec:
    k = name[len("hybrid_codec."):]
    assert k in ref_sd, f"missing in reference: {k}"
    assert tuple(ref_sd[k].shape) == tuple(shape), (k, tuple(ref_sd[k].shape), shape)
# ... and covers every encoder / quantize / bottleneck / latent key of the reference
ours = {n[len("hybrid_codec."):] for n, _, _ in spec}
for k in ref_sd:
    if k.startswith(("encoder.", "quantize.", "quantize_feat.", "latent_tokens")):
        assert k in ours, f"reference key not in inventory: {k}"
missing, unexpected = hc.load_state_dict({n[len("hybrid_codec."):]: t for n, t in sd.items()}, strict=False)
assert not unexpected, unexpected
assert all(m.startswith("decoder.") for m in missing), [m for m in missing if not m.startswith("decoder.")][:5]
print("inventory ok:", len(spec), "tensors")

bn = hc.quantize_feat
bn.force_zero_thres = cfg.force_zero_thres
bn.update(force=True)

for case, (B, H, Wimg, seed) in {"a": (1, 256, 256, 7), "b": (1, 512, 512, 8), "c": (2, 256, 512, 9)}.items():
    x = synth_images(B, H, Wimg, seed)  # [-1,1]
    x01 = x * 0.5 + 0.5
    z, h, stack = hc.encoder(pixel_values=x01, latent_tokens=hc.latent_tokens)
    zq, zres = hc.quantize(z)
    idx = zres["min_encoding_indices"].flatten()
    # bottleneck write path, image by image like compress.py (B=1 calls) ...
    streams, syms, inds = [], [], []
    for b in range(B):
        hb = h[b:b + 1]
        q_enc, q_dec, q_prior = bn.get_qp(0, hb.shape)
        y = bn.encode(hb, q_enc)
        params = bn.y_prior_fusion(q_prior)
        r = bn.compress_four_part_prior(y, params, bn.y_spatial_prior_adaptor_1, bn.y_spatial_prior_adaptor_2,
                                        bn.y_spatial_prior_adaptor_3, bn.y_spatial_prior,
                                        y_spatial_prior_reduction=bn.y_spatial_prior_reduction)
        yq, sc = r[0:4], r[4:8]
        syms.append(torch.stack([t.clamp(-30000, 30000).to(torch.int16)[0] for t in yq]))
        inds.append(torch.stack([bn.gaussian_encoder.build_indexes(s, bn.force_zero_thres).to(torch.int16)[0] for s in sc]))
        streams.append(np.frombuffer(bn.compress(hb, 0), dtype=np.uint8))
    syms, inds = torch.stack(syms), torch.stack(inds)
    n_coded = int((inds >= 0).sum())
    print(f"case {case}: z {tuple(z.shape)} h {tuple(h.shape)} stack {stack} coded {n_coded}/{inds.numel()} "
          f"|sym|max {int(syms.abs().max())} stream bytes {[len(s) for s in streams]}")

    # ---- check the restatement against the reference right here ----
    z2, h2, stack2 = TR.encoder_forward(x01, sd, cfg)
    assert tuple(stack2) == tuple(stack)
    ez, eh = (z2 - z).abs().max().item(), (h2 - h).abs().max().item()
    print(f"   torch_ref vs reference: max|dz|={ez:.2e} (|z|max {z.abs().max():.2f})  max|dh|={eh:.2e} (|h|max {h.abs().max():.2f})")
    assert ez < 2e-4 * max(1, z.abs().max().item()) and eh < 2e-4 * max(1, h.abs().max().item())
    idx2 = TR.vq_indices(z, sd)
    assert torch.equal(idx2, idx)
    y_ref = torch.cat([bn.encode(h[b:b + 1], bn.get_qp(0, h[b:b + 1].shape)[0]) for b in range(B)])
    y2 = TR.bottleneck_analysis(h, sd)
    assert (y2 - y_ref).abs().max().item() < 1e-4 * max(1, y_ref.abs().max().item())
    # per image (B=1 calls like compress.py: CPU conv kernels are not batch-invariant)
    r2 = [TR.four_part_prior_write(y_ref[b:b + 1], sd, cfg.force_zero_thres) for b in range(B)]
    s2, i2 = torch.cat([r[0] for r in r2]), torch.cat([r[1] for r in r2])
    mism = int((s2 != syms).sum() + (i2 != inds).sum())
    print(f"   torch_ref 4-step symbols/indexes mismatches vs reference (same y): {mism}")
    assert mism == 0

    kw = {f"stream_{b}": s for b, s in enumerate(streams)}
    np.savez_compressed(os.path.join(OUT, f"nn_small_{case}.npz"), B=B, H=H, W=Wimg, seed=seed, z=z.numpy(), h=h.numpy(),
                        vq_idx=idx.numpy().astype(np.int32), y=y_ref.numpy(), sym=syms.numpy(), idx=inds.numpy(), **kw)
print("done")
