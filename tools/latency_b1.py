"""Single-image latency of the reference-shaped API (encode_only / decode_only, one 256x256 image per call),
production architecture, synthetic weights.  usage: python tools/latency_b1.py"""
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
sd = W.synth_weights(W.full_spec(LARGE), seed=1234)
codec = Codec(sd, LARGE, dev)
codec.hybrid_codec.quantize_feat.force_zero_thres = 0.12
codec.hybrid_codec.quantize_feat.update(force=True)
x = synth_images(1, 256, 256, seed=5).to(dev)
for _ in range(3):
    enc = codec.encode_only(x)
    codec.decode_only(**enc)
torch.cuda.synchronize()


def timeit(fn, n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print(f"encode_only (device fp32 image -> host byte strings): {timeit(lambda: codec.encode_only(x)):.2f} ms / image")
print(f"decode_only (host byte strings -> device fp32 image): {timeit(lambda: codec.decode_only(**enc)):.2f} ms / image")
