"""Single-image latency (the reference's compress.py / decompress.py call encode_only / decode_only per image):
eager launches vs a hipGraph replay of the same device work.  usage: python tools/graph_probe_b1.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import sgic_amd  # noqa
from sgic_amd import weights as W
from sgic_amd.codec import Codec
from sgic_amd.config import LARGE
from sgic_amd.data import synth_images

dev = torch.device("cuda:0")
cfg = LARGE
sd = W.synth_weights(W.encoder_spec(cfg) + W.codec_misc_spec(cfg) + W.bottleneck_spec(cfg), seed=1234)
codec = Codec(sd, cfg, dev)
codec.hybrid_codec.quantize_feat.force_zero_thres = 0.12
codec.hybrid_codec.quantize_feat.update(force=True)
x = synth_images(1, 256, 256, seed=1000).to(dev)
for _ in range(2):
    codec.encode_device(x)
torch.cuda.synchronize()


def timeit(fn, n=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print("B=1 encode_device eager  ms", timeit(lambda: codec.encode_device(x)), flush=True)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    codec.encode_device(x)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    r = codec.encode_device(x)
torch.cuda.synchronize()
print("B=1 encode_device graph  ms", timeit(g.replay), flush=True)
r0 = codec.encode_device(x)
g.replay()
torch.cuda.synchronize()
print("graph == eager:", torch.equal(r["hs"], r0["hs"]), torch.equal(r["zs"], r0["zs"]), torch.equal(r["hmeta"], r0["hmeta"]))
