"""Per-tile fixed cost of the GEMM kernel: time vs K at fixed (M, N) for each tile mode -> slope (main loop) and
intercept (prologue + epilogue + dispatch).  usage: python tools/gemm_ksweep.py [M N]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import sgic_amd  # noqa
from sgic_amd import ops

dev = torch.device("cuda:0")
M, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (9216, 4096)
ops.AUTOTUNE = False
Ks = (32, 256, 512, 768, 1024, 2048, 4096)
for mode in tuple(int(x) for x in os.environ.get("MODES", "1,3,2,4,5,7,11").split(",")):
    ts = []
    for K in Ks:
        a = torch.rand(M, K, device=dev) * 2 - 1
        w = torch.rand(N, K, device=dev) * 2 - 1
        out = torch.empty(M, N, device=dev)
        res = torch.rand(M, N, device=dev) if os.environ.get('RES') else None
        for _ in range(3):
            ops.gemm(a, w, residual=res, out=out, tile=mode)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gemm(a, w, residual=res, out=out, tile=mode)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    sl, ic = np.polyfit(np.array(Ks[2:], dtype=np.float64), np.array(ts[2:]), 1)
    print(f"mode {mode} M={M} N={N}: " + " ".join(f"K{K}={t:.0f}us({2*M*N*K/t/1e6:.0f}TF)" for K, t in zip(Ks, ts))
          + f" | slope {2*M*N/sl/1e6:.1f} TF, intercept {ic:.1f} us = {ic/(sl*32):.1f} k-iterations", flush=True)
