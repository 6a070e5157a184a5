#!/usr/bin/env python3
"""Kernel durations (dispatch timestamps) of the generative decoder's 3x3 convolutions for the tile modes of the split GEMM."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgic_amd  # noqa: E402,F401
from sgic_amd import ops  # noqa: E402

SHAPES = [(32, 256, 256, 128, 128, 1), (32, 128, 128, 256, 256, 0), (32, 64, 64, 256, 256, 1), (32, 128, 128, 256, 128, 0), (32, 16, 16, 512, 512, 1)]
MODES = (1, 2, 5, 10, 11, 14, 15)


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    print(f"{'B,H,W,Cin,Cout,res':>28} | " + "  ".join(f"m{m:<6}" for m in MODES) + " | best TFLOP/s")
    for (B, H, W, Cin, Cout, res) in SHAPES:
        halo = torch.zeros(B, H + 2, W + 2, Cin, device=dev)
        halo[:, 1:-1, 1:-1] = torch.randn(B, H, W, Cin, device=dev, generator=g)
        w = torch.randn(Cout, 9 * Cin, device=dev, generator=g) * 0.02
        b = torch.randn(Cout, device=dev, generator=g)
        r = torch.randn(B * H * W, Cout, device=dev, generator=g) if res else None
        hp = ops.halo_planes_buffer(dev, B, H, W, Cin)
        ops.split3(halo.view(-1, Cin), out=hp.t.view(3, -1, Cin))
        row = []
        for mode in MODES:
            fn = lambda: ops.conv3x3(hp, w, b, B, H, W, Cin, Cout, residual=r, tile=mode, precision="split3")
            fn()
            ops.profile_begin(16)
            for _ in range(4):
                fn()
            recs = ops.profile_end()
            row.append(sorted(x[1] for x in recs)[len(recs) // 2] * 1e3)
        print(f"{str((B, H, W, Cin, Cout, res)):>28} | " + "  ".join(f"{t:7.1f}" for t in row) + f" | {2.0 * B * H * W * Cout * 9 * Cin / min(row) / 1e6:.1f}")


if __name__ == "__main__":
    main()
