"""Micro-benchmark + check of sgic_gemm_f32 on the GEMM shapes of the compress path (GPU box only)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sgic_amd
from sgic_amd import ops

torch.manual_seed(0)
dev = torch.device("cuda:0")
shapes = [(9248, 3072, 1024), (9248, 1024, 1024), (9248, 4096, 1024), (9248, 1024, 4096),
          (17440, 2304, 768), (17440, 3072, 768), (17440, 768, 3072), (8192, 2304, 768), (2048, 768, 768), (2048, 128, 128),
          (4096, 4096, 4096)]
for (M, N, K) in shapes:
    a = torch.rand(M, K, device=dev) * 2 - 1
    w = torch.rand(N, K, device=dev) * 2 - 1
    b = torch.rand(N, device=dev)
    r = torch.rand(M, N, device=dev)
    out = ops.gemm(a, w, b, r, ops.ACT_GELU)
    ref = torch.nn.functional.gelu(a.double() @ w.double().t() + b.double()) + r.double()
    err = (out.double() - ref).abs().max().item()
    for _ in range(3):
        ops.gemm(a, w, b, r, ops.ACT_GELU, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 20
    e0.record()
    for _ in range(it):
        ops.gemm(a, w, b, r, ops.ACT_GELU, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    print(f"M={M} N={N} K={K}: {ms*1e3:.1f} us  {2*M*N*K/ms/1e9:.1f} TFLOP/s  maxerr={err:.2e}", flush=True)
