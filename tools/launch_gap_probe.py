"""GPU-side cost of one dependent kernel launch: a chain of tiny LayerNorm launches, eager and as a hipGraph."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import sgic_amd  # noqa
from sgic_amd import ops

dev = torch.device("cuda:0")
x = torch.randn(64, 256, device=dev)
w = torch.ones(256, device=dev)
b = torch.zeros(256, device=dev)
y = torch.empty_like(x)
N = 500


def chain():
    for _ in range(N):
        ops.layernorm(x, w, b, out=y)


chain()
torch.cuda.synchronize()
t0 = time.perf_counter()
chain()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"eager: host enqueue {t_host / N * 1e6:.1f} us/launch, end-to-end {t_all / N * 1e6:.1f} us/launch")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    chain()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    chain()
torch.cuda.synchronize()
g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
print(f"graph: {(time.perf_counter() - t0) / 5 / N * 1e6:.2f} us per dependent tiny kernel")
