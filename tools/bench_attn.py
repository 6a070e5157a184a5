import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sgic_amd
from sgic_amd import ops
dev = torch.device("cuda:0")
ops.AUTOTUNE = False
torch.manual_seed(0)
for (L, nseq, heads, bias) in [(289, 32, 16, False), (545, 32, 12, False), (256, 32, 12, True), (50, 32, 12, False)]:
    D = heads * 64
    qkv = torch.randn(nseq * L, 3 * D, device=dev)
    out = torch.empty(nseq * L, D, device=dev)
    b = torch.randn(1, L, L, device=dev) if bias else None
    fl = 4.0 * L * L * 64 * heads * nseq
    res = []
    for mw in tuple(int(x) for x in os.environ.get("MODES", "1,2,3,4,5,6,7").split(",")):
        for _ in range(3):
            ops.attention(qkv[:, :D], qkv[:, D:2*D], qkv[:, 2*D:], out, L, nseq, heads, bias=b, mode=mw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.attention(qkv[:, :D], qkv[:, D:2*D], qkv[:, 2*D:], out, L, nseq, heads, bias=b, mode=mw)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        res.append(f"mode{mw}: {ms*1e3:6.1f}us {fl/ms/1e9:5.1f}TF")
    print(f"L={L} heads={heads}: " + "  ".join(res), flush=True)
