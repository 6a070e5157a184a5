"""Micro-benchmark + check of sgic_gemm_f32 on the GEMM shapes of the compress path (GPU box only)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sgic_amd
from sgic_amd import ops

torch.manual_seed(0)
dev = torch.device("cuda:0")
# (M, N, K, use_residual, act): the epilogues the model actually uses
shapes = [(9248, 3072, 1024, 0, 0), (9248, 1024, 1024, 1, 0), (9248, 4096, 1024, 0, 1), (9248, 1024, 4096, 1, 0),
          (17440, 2304, 768, 0, 0), (17440, 768, 768, 1, 0), (17440, 3072, 768, 0, 1), (17440, 768, 3072, 1, 0),
          (8192, 2304, 768, 0, 0), (8192, 1536, 768, 0, 1), (8192, 768, 1536, 1, 0), (2048, 768, 768, 0, 0), (2048, 128, 128, 0, 0),
          (4096, 4096, 4096, 0, 0)]
tot_t = tot_f = 0.0
for (M, N, K, use_r, act) in shapes:
    a = torch.rand(M, K, device=dev) * 2 - 1
    w = torch.rand(N, K, device=dev) * 2 - 1
    b = torch.rand(N, device=dev)
    r = torch.rand(M, N, device=dev)
    rr = r if use_r else None
    out = ops.gemm(a, w, b, rr, act)
    ref = a.double() @ w.double().t() + b.double()
    ref = torch.nn.functional.gelu(ref) if act else ref
    ref = ref + r.double() if use_r else ref
    err = (out.double() - ref).abs().max().item()
    for _ in range(3):
        ops.gemm(a, w, b, rr, act, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 20
    e0.record()
    for _ in range(it):
        ops.gemm(a, w, b, rr, act, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    tot_t += ms; tot_f += 2*M*N*K
    print(f"M={M} N={N} K={K} res={use_r} act={act}: {ms*1e3:.1f} us  {2*M*N*K/ms/1e9:.1f} TFLOP/s  maxerr={err:.2e}", flush=True)
print(f"aggregate {tot_f/tot_t/1e9:.1f} TFLOP/s")
