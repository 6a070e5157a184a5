"""Per-kernel GPU time of ONE single-image encode_only (production architecture) from a rocprofv3 kernel trace.
usage (GPU box):  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_b1e -- python3 tools/trace_b1.py
                  python tools/trace_b1.py --summarise gpurun_out/kt_b1e"""
import csv
import glob
import os
import sys

if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    # the last 10 encodes are identical launch sequences: take the trailing 10 x n kernels
    names = [r["Kernel_Name"] for r in rows]
    n = None
    for cand in range(200, 1500):          # period of the launch sequence
        if names[-cand:] == names[-2 * cand:-cand]:
            n = cand
            break
    assert n, "no periodic tail found"
    tail = rows[-10 * n:]
    agg = {}
    for r in tail:
        k = r["Kernel_Name"].split("(")[0][:60]
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    span = (int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])) / 1e3 / 10
    busy = sum(a[1] for a in agg.values()) / 10
    print(f"{n} kernels per encode; GPU busy {busy / 1e3:.2f} ms of {span / 1e3:.2f} ms per encode")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:60s} calls/encode {a[0] / 10:6.1f}  avg {a[1] / a[0]:7.1f} us  total {a[1] / 10 / 1e3:6.2f} ms")
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import sgic_amd  # noqa
from sgic_amd import weights as W
from sgic_amd.codec import Codec
from sgic_amd.config import LARGE
from sgic_amd.data import synth_images

sd = W.synth_weights(W.encoder_spec(LARGE) + W.codec_misc_spec(LARGE) + W.bottleneck_spec(LARGE), seed=1234)
codec = Codec(sd, LARGE, "cuda:0")
codec.hybrid_codec.quantize_feat.force_zero_thres = 0.12
codec.hybrid_codec.quantize_feat.update(force=True)
x = synth_images(1, 256, 256, seed=5).cuda()
for _ in range(25):
    codec.encode_only(x)
torch.cuda.synchronize()
