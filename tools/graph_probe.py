"""How much of the encoder's wall time is launch gaps?  eager vs hipGraph replay of HybridEncoderHIP.forward."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import sgic_amd  # noqa
from sgic_amd import ops
from sgic_amd import weights as W
from sgic_amd.codec import Codec
from sgic_amd.config import LARGE
from sgic_amd.data import synth_images

dev = torch.device("cuda:0")
cfg = LARGE
sd = W.synth_weights(W.encoder_spec(cfg) + W.codec_misc_spec(cfg) + W.bottleneck_spec(cfg), seed=1234)
codec = Codec(sd, cfg, dev)
x = synth_images(32, 256, 256, seed=1000).to(dev)
enc = codec.encoder
for _ in range(2):
    z, h, _ = enc.forward(x)     # autotune + lazily built tables
torch.cuda.synchronize()


def timeit(fn, n=5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print("eager   ms/forward", timeit(lambda: enc.forward(x)), flush=True)
ops.profile_begin()
print("eager+events ms/forward", timeit(lambda: enc.forward(x)), flush=True)
ops.profile_end()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    enc.forward(x)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    zg, hg, _ = enc.forward(x)
torch.cuda.synchronize()
print("graph   ms/forward", timeit(g.replay), flush=True)
g.replay()
torch.cuda.synchronize()
print("graph output == eager output:", torch.equal(zg, z), torch.equal(hg, h))
