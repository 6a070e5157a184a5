#!/usr/bin/env python3
"""Does a one-wave-per-image kernel on a side stream run UNDER the big split GEMMs of the main stream?  Times N back-to-back
(9248 x 4096 x 1024) GEMMs alone, the JPEG batch decode alone, and both together (decode enqueued on a side stream first)."""
import io
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgic_amd  # noqa: E402,F401
from PIL import Image  # noqa: E402
from sgic_amd import jpeg as J, ops  # noqa: E402
from sgic_amd.data import synth_images  # noqa: E402

dev = torch.device("cuda:0")
noise = ((synth_images(32, 256, 256, 3) * 0.5 + 0.5) * 255).round().byte().permute(0, 2, 3, 1).numpy()
datas = []
for i in range(32):
    buf = io.BytesIO()
    Image.fromarray(noise[i]).save(buf, "JPEG", quality=90)
    datas.append(buf.getvalue())
jb = J.JpegBatch(datas, alloc=lambda n: torch.empty(n, dtype=torch.uint8).pin_memory())
side = torch.cuda.Stream(device=dev)
M, N, K = 9248, 4096, 1024
a = torch.randn(M, K, device=dev)
w = torch.randn(N, K, device=dev) * 0.03
ap = ops.split3_planes(a)
out = torch.empty(M, N, device=dev)


def gemms(n, tile):
    for _ in range(n):
        ops.gemm(ap, w, out=out, tile=tile, precision="split3")


def timed(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for tile in (10, 4):
    gemms(3, tile)
    jb.decode(dev)
    t_g = timed(lambda: gemms(60, tile))

    def dec_side():
        with torch.cuda.stream(side):
            jb.decode(dev, check=False)
        torch.cuda.current_stream().wait_stream(side)
    t_d = timed(dec_side)

    def both():
        with torch.cuda.stream(side):
            jb.decode(dev, check=False)
        gemms(60, tile)
        torch.cuda.current_stream().wait_stream(side)
    t_b = timed(both)
    print(f"tile mode {tile}: 60 GEMMs alone {t_g:.2f} ms | decode alone {t_d:.2f} ms | together {t_b:.2f} ms "
          f"(perfect overlap {max(t_g, t_d):.2f}, serial {t_g + t_d:.2f})", flush=True)
