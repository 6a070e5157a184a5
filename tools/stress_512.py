"""config 5: batch=16 512x512 compress + decompress at full size on one MI355X (LDS tile / HBM stress)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sgic_amd
from sgic_amd import weights as W
from sgic_amd.codec import Codec, ClipCodec
from sgic_amd.config import LARGE, CLIP_B32
from sgic_amd.data import synth_images
B, S = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 512
sd = W.synth_weights(W.full_spec(LARGE), seed=1234)
codec = Codec(sd, LARGE, "cuda:0")
codec.hybrid_codec.quantize_feat.update(force=True)
clipc = ClipCodec(W.synth_weights(W.clip_spec(CLIP_B32), seed=4321), CLIP_B32, "cuda:0")
x = synth_images(B, S, S, 11).cuda()
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    encs = codec.encode_batch(x)
    unit, q = clipc.batch_to_codes(x)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    x_hat = codec.decode_batch(encs)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"iter {it}: compress {B/(t1-t0):.1f} img/s ({(t1-t0)*1e3:.0f} ms)  decompress {B/(t2-t1):.1f} img/s ({(t2-t1)*1e3:.0f} ms)  "
          f"bytes/img {sum(len(e['z_bit_stream'])+len(e['h_bit_stream']) for e in encs)/B:.0f}  mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
one = codec.decode_batch([encs[3]])
print("batch-invariant decode:", bool(torch.equal(one[0], x_hat[3])), " x_hat finite:", bool(torch.isfinite(x_hat).all()))
enc1 = codec.encode_batch(x[5:6].contiguous())[0]
print("batch-invariant encode:", enc1["h_bit_stream"] == encs[5]["h_bit_stream"] and enc1["z_bit_stream"] == encs[5]["z_bit_stream"])
