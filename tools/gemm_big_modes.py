#!/usr/bin/env python3
"""Kernel durations (dispatch timestamps) of the large compress GEMM shapes for every tile mode of the split GEMM."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgic_amd  # noqa: E402,F401
from sgic_amd import ops  # noqa: E402

SHAPES = [(9248, 1024, 4096, 1, 0), (9248, 4096, 1024, 0, 1), (9248, 3072, 1024, 0, 0), (17440, 768, 3072, 1, 0), (17440, 3072, 768, 0, 1),
          (8192, 768, 3072, 1, 0), (9248, 1024, 1024, 1, 0), (8192, 2304, 768, 0, 0), (17440, 2304, 768, 0, 0)]
MODES = (1, 2, 6, 8, 9, 10, 11, 12, 13)


def timed(fn, reps=8):
    fn()
    ops.profile_begin(4 * reps)
    for _ in range(reps):
        fn()
    recs = ops.profile_end()
    ms = sorted(r[1] for r in recs)
    return ms[len(ms) // 2] * 1e3


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    print(f"{'shape':>26} | " + "  ".join(f"m{m:<5}" for m in MODES) + " | best TFLOP/s")
    for (M, N, K, res, act) in SHAPES:
        a = torch.randn(M, K, device=dev, generator=g)
        w = torch.randn(N, K, device=dev, generator=g) * 0.03
        b = torch.randn(N, device=dev, generator=g)
        r = torch.randn(M, N, device=dev, generator=g) if res else None
        ap = ops.Planes(M, K, dev)
        ops.split3(a, out=ap.t.view(3, M, K))
        row = [timed(lambda: ops.gemm(ap, w, b, residual=r, act=act, tile=mode, precision="split3")) for mode in MODES]
        print(f"{str((M, N, K, res, act)):>26} | " + "  ".join(f"{t:6.1f}" for t in row) + f" | {2.0 * M * N * K / min(row) / 1e6:.1f}")


if __name__ == "__main__":
    main()
