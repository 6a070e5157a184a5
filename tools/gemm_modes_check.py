"""Every GEMM tile mode must give bitwise identical results (fixed k order).  usage: python tools/gemm_modes_check.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import sgic_amd  # noqa
from sgic_amd import ops
from sgic_amd._lib import lib

dev = torch.device("cuda:0")
ops.AUTOTUNE = False
torch.manual_seed(0)
ok = True
for (M, N, K, res, act) in [(9248, 1024, 1024, True, 0), (9248, 4096, 1024, False, 1), (17440, 768, 768, True, 0), (1600, 768, 3072, True, 0),
                            (300, 200, 64, False, 0), (8192, 3072, 768, False, 1), (70000, 128, 128, True, 2), (129, 132, 32, False, 0)]:
    a = torch.rand(M, K, device=dev) * 2 - 1
    w = torch.rand(N, K, device=dev) * 2 - 1
    b = torch.rand(N, device=dev)
    r = torch.rand(M, N, device=dev) if res else None
    outs = {}
    for mode in (1, 2, 3, 4, 5, 7, 9, 10, 11, 12, 13, 14):
        lib.sgic_gemm_set_tile(mode)
        outs[mode] = ops.gemm(a, w, b, act=act, residual=r).clone()
    lib.sgic_gemm_set_tile(0)
    torch.cuda.synchronize()
    bad = [m for m in outs if not torch.equal(outs[m], outs[1])]
    ref = torch.nn.functional.linear(a.double(), w.double(), b.double())
    print(M, N, K, "mismatching modes:", bad, "max err vs fp64 (pre-act):" if act == 0 and not res else "",
          float((outs[1].double() - ref).abs().max()) if act == 0 and not res else "", flush=True)
    ok &= not bad
print("ALL MODES IDENTICAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
