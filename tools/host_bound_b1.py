"""Is a single-image encode host-bound?  Times the Python enqueue of `Codec.encode_device` (B = 1, production architecture)
against enqueue + GPU drain, and profiles the host side.  usage: python tools/host_bound_b1.py
Round 2: 549 launches per encode; host enqueue 7.1 ms (13 us per launch), enqueue + GPU drain 13.9 ms -> the GPU side is the limit."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, sgic_amd
from sgic_amd import weights as W
from sgic_amd.codec import Codec
from sgic_amd.config import LARGE
from sgic_amd.data import synth_images
dev = torch.device("cuda:0")
sd = W.synth_weights(W.encoder_spec(LARGE) + W.codec_misc_spec(LARGE) + W.bottleneck_spec(LARGE), seed=1234)
codec = Codec(sd, LARGE, dev)
codec.hybrid_codec.quantize_feat.force_zero_thres = 0.12
codec.hybrid_codec.quantize_feat.update(force=True)
x = synth_images(1, 256, 256, seed=5).to(dev)
for _ in range(5):
    codec.encode_device(x)
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = codec.encode_device(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
print(f"B=1 encode_device: host enqueue {sorted(enq)[5]:.2f} ms, enqueue+GPU drain {sorted(tot)[5]:.2f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): codec.encode_device(x)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
