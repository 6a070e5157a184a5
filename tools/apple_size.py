"""The reference's worked example geometry (IO/images/apple.jpg: 859x1000 -> padded 1024x1024, 16 tiles, 512 tokens,
feat 32x32) through compress + decompress with the production architecture (synthetic weights)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sgic_amd
from sgic_amd import weights as W
from sgic_amd.codec import Codec, ClipCodec
from sgic_amd.config import LARGE, CLIP_B32
from sgic_amd.data import synth_images
from sgic_amd.filemaker import pack_c2df, unpack_c2df
sd = W.synth_weights(W.full_spec(LARGE), seed=1234)
codec = Codec(sd, LARGE, "cuda:0")
codec.hybrid_codec.quantize_feat.update(force=True)
clipc = ClipCodec(W.synth_weights(W.clip_spec(CLIP_B32), seed=4321), CLIP_B32, "cuda:0")
img = synth_images(1, 1024, 1024, 3)[:, :, :1000, :859].contiguous()
xp = torch.nn.functional.pad(img, (0, 165, 0, 24), mode="replicate").cuda().contiguous()
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    enc = codec.encode_batch(xp)[0]
    unit, q = clipc.batch_to_codes(img.cuda())
    torch.cuda.synchronize(); t1 = time.perf_counter()
    x_hat = codec.decode_batch([enc])
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"iter {it}: compress {(t1-t0)*1e3:.0f} ms, decompress {(t2-t1)*1e3:.0f} ms", flush=True)
enc["clip_stream"] = clipc.compress_codes(q[0].cpu().numpy()); enc["clip_meta"] = clipc.meta(512)
blob = pack_c2df(enc, {"version": 2, "model_id": clipc.model_name, "embed_dim": 512, "quant_type": "u8_symmetric_-1_1",
                       "image_hw": [1000, 859], "padding": [0, 165, 0, 24]})
e2, h2 = unpack_c2df(blob)
print("c2df bytes", len(blob), "z", len(enc["z_bit_stream"]), "h", len(enc["h_bit_stream"]), "clip", len(enc["clip_stream"]),
      "token_length", enc["token_length"], "stack", enc["stack_shape"], "feat", tuple(enc["feat_shape"]), "x_hat", tuple(x_hat.shape),
      "finite", bool(torch.isfinite(x_hat).all()))
