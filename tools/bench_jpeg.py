#!/usr/bin/env python3
"""GPU JPEG decode (csrc/jpeg.hip) vs Pillow on the host: a batch of 32 files 256x256, noise (the CLI benchmark's worst case for an
entropy decoder) and natural-like content.  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import io
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import jpeg_cases  # noqa: E402
import sgic_amd  # noqa: E402,F401
from PIL import Image  # noqa: E402
from sgic_amd import jpeg as J  # noqa: E402
from sgic_amd.data import synth_images  # noqa: E402

rng = np.random.default_rng(0)
noise = ((synth_images(32, 256, 256, 3) * 0.5 + 0.5) * 255).round().byte().permute(0, 2, 3, 1).numpy()
sets = {"noise_q90": [(noise[i], dict(quality=90)) for i in range(32)],
        "natural_q90": [(jpeg_cases.natural_like(256, 256, rng), dict(quality=90)) for i in range(32)],
        "natural_q75_1024": [(jpeg_cases.natural_like(1024, 1024, rng), dict(quality=75)) for i in range(8)]}
for name, items in sets.items():
    datas = []
    for img, kw in items:
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, "JPEG", **kw)
        datas.append(buf.getvalue())
    t0 = time.perf_counter()
    for d in datas:
        np.asarray(Image.open(io.BytesIO(d)).convert("RGB"))
    t_pil = (time.perf_counter() - t0) / len(datas)
    t0 = time.perf_counter()
    b = J.JpegBatch(datas)
    t_parse = (time.perf_counter() - t0) / len(datas)
    out = b.decode("cuda:0")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        out = b.decode("cuda:0", check=False)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{name}: {len(datas)} files, {sum(map(len, datas)) / len(datas) / 1024:.1f} KiB each | Pillow {t_pil * 1e3:.2f} ms/file on one core | "
          f"host parse {t_parse * 1e3:.2f} ms/file | GPU batch decode {ms:.2f} ms ({ms / len(datas):.3f} ms/file, one wave per file)", flush=True)
