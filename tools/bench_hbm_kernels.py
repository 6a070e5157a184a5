"""Effective HBM bandwidth of the memory-bound kernels at the shapes the model uses (algorithmic bytes / time)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sgic_amd
from sgic_amd import ops
dev = torch.device("cuda:0")


def timeit(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3


def rep(name, bytes_, t):
    print(f"{name:58s} {t*1e6:8.1f} us  {bytes_/t/1e12:5.2f} TB/s  ({bytes_/1e6:.1f} MB algorithmic)", flush=True)


for (M, C) in [(9248, 1024), (17440, 768), (8192, 768), (9248, 4096)]:
    if C > 2048:
        continue
    x = torch.randn(M, C, device=dev); g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev); y = torch.empty_like(x)
    rep(f"layernorm M={M} C={C}", 2 * M * C * 4, timeit(lambda: ops.layernorm(x, g, b, out=y)))
B, H, W, C = 32, 16, 16, 768
x = torch.randn(B * H * W, C, device=dev); w = torch.randn(25, C, device=dev); bb = torch.zeros(C, device=dev); ls = torch.ones(C, device=dev)
y = torch.empty_like(x)
rep("dwconv5x5 NHWC TM16 (32,16,16,768)", 2 * x.numel() * 4, timeit(lambda: ops.dwconv(x, w, bb, ls, B, H, W, 5, tile16=True, out=y)))
for (H, C) in [(256, 128), (128, 256), (64, 256), (16, 512)]:
    x = torch.randn(32 * H * H, C, device=dev); g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    n = x.numel() * 4
    rep(f"groupnorm+swish -> halo (32,{H},{H},{C}) [2 reads + 1 write]", 3 * n, timeit(lambda: ops.groupnorm(x, g, b, 32, H, H, swish=True, halo=True), it=5))
    hb = ops.halo_buffer(dev, 32, H, H, C)
    wt = torch.randn(C, 9 * C, device=dev); bo = torch.zeros(C, device=dev)
    t = timeit(lambda: ops.conv3x3(hb, wt, bo, 32, H, H, C, C), it=5)
    print(f"conv3x3 implicit GEMM (32,{H},{H},{C}->{C}) {t*1e6:9.1f} us  {2*32*H*H*9*C*C/t/1e12:6.1f} TFLOP/s", flush=True)
Bq, Hq = 32, 8
y = torch.randn(Bq * Hq * Hq, 64, device=dev); sm = torch.rand(Bq * Hq * Hq, 128, device=dev) + 0.05
yh = torch.zeros(Bq * Hq * Hq, 128, device=dev); sym = torch.zeros(Bq, 4, 16, Hq, Hq, dtype=torch.int16, device=dev); idx = torch.zeros_like(sym)
rep("quant_step (B=32, 8x8x64) [launch-bound]", Bq * Hq * Hq * 16 * (3 * 4 + 4 + 4), timeit(lambda: ops.quant_step(y, sm, sm[:, 64:], 128, yh, 128, Bq, Hq, Hq, 64, 1, 0.12, sym, idx)))
x = torch.rand(32, 3, 256, 256, device=dev) * 2 - 1
rep("im2col_patch 16x16 (+x*0.5+0.5)", 2 * x.numel() * 4, timeit(lambda: ops.im2col_patch(x, 16, 0.5, 0.5, tile16=True)))

# VQ nearest-code search (1024 tokens x 4096 codes x 12 dims at the bench batch; 32 tokens for a single image)
cbk = torch.randn(4096, 12, device=dev)
for ntok in (1024, 32):
    zt = torch.randn(ntok, 12, device=dev)
    rep(f"vq_argmin {ntok} tokens x 4096 codes", ntok * 4096 * 12 * 4, timeit(lambda: ops.vq_argmin(zt, cbk, l2norm=True)))
