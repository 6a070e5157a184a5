"""Single-image attention launches (nseq = 1): how the time depends on L (ragged query rows L % 32, key tail L % 32) and the attn_mode.
usage: python tools/bench_attn_b1.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sgic_amd  # noqa
from sgic_amd import ops
dev = torch.device("cuda:0")
ops.AUTOTUNE = False
torch.manual_seed(0)
for (L, nseq, heads) in [(288, 1, 16), (289, 1, 16), (290, 1, 16), (320, 1, 16), (544, 1, 12), (545, 1, 12), (256, 1, 12), (256, 4, 12), (50, 1, 12), (64, 1, 12), (77, 1, 8)]:
    D = heads * 64
    qkv = torch.randn(nseq * L, 3 * D, device=dev)
    out = torch.empty(nseq * L, D, device=dev)
    res = []
    for mw in (1, 2, 5):
        for _ in range(3):
            ops.attention(qkv[:, :D], qkv[:, D:2*D], qkv[:, 2*D:], out, L, nseq, heads, mode=mw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.attention(qkv[:, :D], qkv[:, D:2*D], qkv[:, 2*D:], out, L, nseq, heads, mode=mw)
        e1.record(); torch.cuda.synchronize()
        res.append(f"mode{mw}: {e0.elapsed_time(e1) / 20 * 1e3:6.1f} us")
    print(f"L={L} nseq={nseq} heads={heads}: " + "  ".join(res), flush=True)
