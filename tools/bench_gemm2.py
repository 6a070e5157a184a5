import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sgic_amd
from sgic_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (M, N, K) in [(4096, 2048, 1024), (8192, 2048, 1024), (6144, 2048, 1024), (4096, 2048, 4096), (4096, 4096, 1024), (2048, 2048, 1024),
                  (4096, 1024, 1024), (8192, 1024, 1024), (9248, 1024, 1024), (9216, 1024, 1024), (8192, 1024, 4096)]:
    a = torch.rand(M, K, device=dev) * 2 - 1
    w = torch.rand(N, K, device=dev) * 2 - 1
    out = torch.empty(M, N, device=dev)
    for _ in range(3):
        ops.gemm(a, w, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.gemm(a, w, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    nb128 = ((M + 127) // 128) * ((N + 127) // 128)
    print(f"M={M} N={N} K={K}: blocks128={nb128} ({nb128/256:.2f}/CU) {ms*1e3:.1f} us  {2*M*N*K/ms/1e9:.1f} TFLOP/s", flush=True)
