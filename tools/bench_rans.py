"""rANS coder kernels at the bench geometry: 32 streams x 4096 symbols (4 prior steps x 1024).  usage: python tools/bench_rans.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import sgic_amd  # noqa
from sgic_amd import ops
from sgic_amd._lib import call
from sgic_amd.bottleneck import gaussian_cdf_table
from sgic_amd.entropy.MLCodec_rans import _Tables

dev = torch.device("cuda:0")
cdf, ln, off = gaussian_cdf_table()
tab = _Tables()
g = tab.add(cdf, ln, off)
B, n = 32, 4096
rng = np.random.default_rng(0)
idx = rng.integers(0, 256, size=(B, n)).astype(np.int16)
idx[rng.random((B, n)) < 0.3] = -1
scale = np.exp(np.linspace(np.log(0.11), np.log(64), 256))[np.clip(idx, 0, 255)]
sym = np.clip(np.rint(rng.standard_normal((B, n)) * scale), -30000, 30000).astype(np.int16)
d_sym, d_idx = torch.from_numpy(sym).to(dev), torch.from_numpy(idx).to(dev)


def timed(fn, reps=10):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


out, meta = ops.rans_encode_batch(tab.handles[g], d_sym, d_idx, B, n)
print(f"encode: {timed(lambda: ops.rans_encode_batch(tab.handles[g], d_sym, d_idx, B, n)):.0f} us for {B} x {n} symbols, "
      f"stream bytes {meta[1].min().item()}..{meta[1].max().item()}")
state = torch.zeros(B, 4, dtype=torch.int32, device=dev)
dec = torch.zeros(B, n, dtype=torch.int16, device=dev)
cap = out.shape[1]


def decode_all():
    call("sgic_rans_decode_init_batch", ops._p(out), cap, ops._p(meta[0]), ops._p(meta[1]), B, ops._p(state))
    for k in range(4):   # the codec decodes in 4 dependent launches of 1024 symbols
        call("sgic_rans_decode_batch", tab.handles[g], ops._p(out), cap, ops._p(meta[0]), ops._p(meta[1]), B, ops._p(state),
             ops._p(d_idx[:, k * 1024:]), 1024, n, ops._p(dec[:, k * 1024:]), n)


t = timed(decode_all)
ref = np.where(idx < 0, 0, sym)
ok = np.array_equal(dec.cpu().numpy(), ref) and int(state[:, 2].abs().sum()) == 0
print(f"decode: {t:.0f} us for init + 4 launches of {B} x 1024 symbols ({t / 4:.0f} us per launch); round trip exact: {ok}")
