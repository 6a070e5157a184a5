"""Run one GEMM shape a few times (for rocprofv3 --pmc / --kernel-trace)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sgic_amd
from sgic_amd import ops
M, N, K = [int(v) for v in sys.argv[1:4]]
it = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device("cuda:0")
torch.manual_seed(0)
a = torch.rand(M, K, device=dev) * 2 - 1
w = torch.rand(N, K, device=dev) * 2 - 1
out = torch.empty(M, N, device=dev)
for _ in range(it):
    ops.gemm(a, w, out=out)
torch.cuda.synchronize()
