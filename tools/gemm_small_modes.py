#!/usr/bin/env python3
"""Kernel durations (dispatch timestamps, no host time) of the single-image GEMM shapes: fp32 MFMA modes 13 / 15 vs the bf16x3
split kernel modes 3 / 4 (+ its activation split pass when the operand arrives as fp32)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgic_amd  # noqa: E402,F401
from sgic_amd import ops  # noqa: E402

SHAPES = [(289, 1024, 1024, 1, 0), (289, 3072, 1024, 0, 0), (289, 4096, 1024, 0, 1), (289, 1024, 4096, 1, 0), (545, 768, 768, 1, 0),
          (545, 2304, 768, 0, 0), (545, 3072, 768, 0, 1), (545, 768, 3072, 1, 0), (256, 768, 768, 1, 0), (50, 768, 768, 1, 0),
          (64, 768, 3072, 1, 0), (64, 64, 512, 0, 0)]


def timed(fn, reps=20):
    fn()
    ops.profile_begin(4 * reps)
    for _ in range(reps):
        fn()
    recs = ops.profile_end()
    per = len(recs) // reps
    ms = sorted(sum(r[1] for r in recs[i * per:(i + 1) * per]) for i in range(reps))
    return ms[len(ms) // 2] * 1e3


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    print(f"{'shape':>24} | f32 m13  f32 m15 | s3 m2   s3 m3   s3 m4  (planes operand)")
    for (M, N, K, res, act) in SHAPES:
        a = torch.randn(M, K, device=dev, generator=g)
        w = torch.randn(N, K, device=dev, generator=g) * 0.03
        b = torch.randn(N, device=dev, generator=g)
        r = torch.randn(M, N, device=dev, generator=g) if res else None
        ap = ops.split3_planes(a)
        row = []
        for mode in (13, 15):
            row.append(timed(lambda: ops.gemm(a, w, b, residual=r, act=act, tile=mode, precision="f32")))
        for mode in (2, 3, 4):
            row.append(timed(lambda: ops.gemm(ap, w, b, residual=r, act=act, tile=mode, precision="split3")))
        print(f"{str((M, N, K, res, act)):>24} | " + "  ".join(f"{t:6.1f}" for t in row))


if __name__ == "__main__":
    main()
