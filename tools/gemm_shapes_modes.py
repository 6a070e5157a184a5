"""Time every tile mode on the model's dominant GEMM shapes.  usage: python tools/gemm_shapes_modes.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import sgic_amd  # noqa
from sgic_amd import ops

dev = torch.device("cuda:0")
ops.AUTOTUNE = False
MODES = tuple(int(x) for x in os.environ.get("MODES", "1,2,3,4,5,7,9,10,11,12,13,14,15").split(","))
SHAPES = [(9248, 1024, 4096, 1), (9248, 4096, 1024, 0), (9248, 3072, 1024, 0), (17440, 3072, 768, 0), (17440, 768, 3072, 1),
          (8192, 3072, 768, 0), (8192, 768, 3072, 1), (17440, 2304, 768, 0), (9248, 1024, 1024, 1), (8192, 2304, 768, 0),
          (8192, 768, 768, 1), (17440, 768, 768, 1), (1600, 768, 3072, 1), (1600, 3072, 768, 0), (1600, 2304, 768, 0), (1600, 768, 768, 1), (2048, 768, 768, 0), (2048, 128, 256, 1),
          (2048, 512, 128, 0), (289, 1024, 1024, 1), (289, 1024, 4096, 1), (289, 3072, 1024, 0), (289, 4096, 1024, 0), (545, 768, 3072, 1),
          (2048, 768, 1536, 1), (64, 64, 512, 0)]
if os.environ.get("SHAPES") == "small":
    SHAPES = [s for s in SHAPES if s[0] <= 2048]
for (M, N, K, res) in SHAPES:
    a = torch.rand(M, K, device=dev) * 2 - 1
    w = torch.rand(N, K, device=dev) * 2 - 1
    r = torch.rand(M, N, device=dev) if res else None
    out = torch.empty(M, N, device=dev)
    line = []
    for mode in MODES:
        for _ in range(3):
            ops.gemm(a, w, residual=r, out=out, tile=mode)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gemm(a, w, residual=r, out=out, tile=mode)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e3
        line.append((2 * M * N * K / t / 1e6, mode))
    best = max(line)
    print(f"({M},{N},{K},res={res}) " + " ".join(f"m{m}={tf:.0f}" for tf, m in line) + f" | best m{best[1]} {best[0]:.1f} TF = {2 * M * N * K / best[0] / 1e6:.1f} us", flush=True)
