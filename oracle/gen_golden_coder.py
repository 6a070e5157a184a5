"""Generate the integer/byte golden vectors in tests/golden/ from the REAL reference coder
(oracle/_ref, compiled from /root/reference/src/cpp) and the reference's own Python
(entropy/entropy_models.py GaussianEncoder.update).  Build-container only; committed output:

  tests/golden/cdf_table.npz     256x103 int32 Gaussian CDF table + lengths + offsets (A9)
  tests/golden/pmf_kats.npz      pmf_to_quantized_cdf KATs incl. the steal path (N6)
  tests/golden/rans_kats.npz     (symbols, indexes) -> stream bytes, multi-call decode (N1-N3,N5)
  tests/golden/apple.c2df ...    copies of the reference's worked-example DATA files (IO/*)

Run:  make -C oracle && python oracle/gen_golden_coder.py
"""
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

ref_harness.setup()
import torch  # noqa: E402
from entropy.entropy_models import EntropyCoder, GaussianEncoder  # noqa: E402  (reference code)
from entropy.MLCodec_rans import RansDecoder, RansEncoder  # noqa: E402  (reference C++)
from entropy.MLCodec_CXX import pmf_to_quantized_cdf  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)

# ---- A9: CDF table through the reference's own update() ----
ec = EntropyCoder(False, 1)
ge = GaussianEncoder(distribution="gaussian")
ge.update(force=True, entropy_coder=ec)
cdf, cdf_len, offset = [np.ascontiguousarray(np.asarray(t)).astype(np.int32) for t in ge.get_cdf_info()]
np.savez_compressed(os.path.join(OUT, "cdf_table.npz"), cdf=cdf, cdf_length=cdf_len, offset=offset,
                    scale_table=ge.scale_table.numpy())
print("cdf", cdf.shape, cdf_len.min(), cdf_len.max(), offset.min(), offset.max())

# ---- N6: pmf KATs ----
rng = np.random.default_rng(20251031)
pmfs, outs = [], []
for case in range(24):
    n = int(rng.integers(3, 120))
    p = rng.random(n).astype(np.float32) ** (1 + case % 5)
    if case % 3 == 0:  # zero-probability bins -> steal path
        p[rng.integers(0, n, size=max(1, n // 4))] = 0.0
    if case % 4 == 1:  # very peaky
        p[:] = 1e-9
        p[n // 2] = 1.0
    p = (p / p.sum()).astype(np.float32)
    c = np.array(pmf_to_quantized_cdf(p.tolist(), 16), dtype=np.uint32)
    pmfs.append(p)
    outs.append(c)
np.savez_compressed(os.path.join(OUT, "pmf_kats.npz"), n=len(pmfs),
                    **{f"pmf_{i}": p for i, p in enumerate(pmfs)}, **{f"cdf_{i}": c for i, c in enumerate(outs)})

# ---- N1-N3,N5: rANS KATs ----
scale_table = ge.scale_table.numpy()


def n_entries(sym, idx):
    tot = 0
    for s, i in zip(sym.tolist(), idx.tolist()):
        if i < 0:
            continue
        mx = int(cdf_len[i]) - 2
        v = s - int(offset[i])
        tot += 1
        raw = None
        if v < 0:
            raw = -2 * v - 1
        elif v >= mx:
            raw = 2 * (v - mx)
        if raw is not None:
            nb = 0
            while (raw >> (2 * nb)) != 0:
                nb += 1
            tot += nb // 3 + 1 + nb
    return tot


def make_case(kind, n, parts):
    if kind == "typical":
        idx = rng.integers(0, 256, size=n).astype(np.int16)
        idx[rng.random(n) < 0.15] = -1
        sg = scale_table[np.maximum(idx, 0)]
        sym = np.rint(rng.standard_normal(n) * sg * 1.3).astype(np.int16)
    elif kind == "lowscale":  # what the codec mostly sees: sigma near 0.11..1
        idx = rng.integers(0, 90, size=n).astype(np.int16)
        idx[rng.random(n) < 0.5] = -1
        sym = np.rint(rng.standard_normal(n) * 1.5).astype(np.int16)
    elif kind == "bypass":  # large magnitudes, n_bypass up to 8, both signs
        idx = rng.integers(0, 256, size=n).astype(np.int16)
        mag = (2.0 ** rng.uniform(0, 14.87, size=n)).astype(np.int64)
        sym = (mag * rng.choice([-1, 1], size=n)).clip(-30000, 30000).astype(np.int16)
    elif kind == "edges":  # exactly at offset / max_value boundaries
        idx = rng.integers(0, 256, size=n).astype(np.int16)
        mx = cdf_len[idx] - 2
        choice = rng.integers(0, 6, size=n)
        v = np.select([choice == 0, choice == 1, choice == 2, choice == 3, choice == 4],
                      [0, mx - 1, mx, mx + 1, -1], default=-2)
        sym = (v + offset[idx]).astype(np.int16)
    else:
        raise ValueError(kind)
    # split into `parts` encode_with_indexes calls (like the 4 steps of compress())
    cuts = [0] + sorted(rng.integers(0, n + 1, size=parts - 1).tolist()) + [n]
    return sym, idx, np.array(cuts, dtype=np.int64)


kats = {}
specs = [("typical", 4096, 4), ("typical", 1024, 4), ("lowscale", 4096, 4), ("lowscale", 16384, 4),
         ("bypass", 512, 2), ("bypass", 2048, 4), ("edges", 1024, 3), ("typical", 97, 1), ("edges", 64, 2)]
for ci, (kind, n, parts) in enumerate(specs):
    while True:
        sym, idx, cuts = make_case(kind, n, parts)
        enc = RansEncoder(False, 1)
        dec = RansDecoder(1)
        g = enc.add_cdf(cdf, cdf_len, offset)
        dec.add_cdf(cdf, cdf_len, offset)
        enc.reset()
        for a, b in zip(cuts[:-1], cuts[1:]):
            enc.encode_with_indexes(np.ascontiguousarray(sym[a:b]), np.ascontiguousarray(idx[a:b]), g)
        ne = n_entries(sym, idx)
        if ne < 16:
            continue
        enc.flush()
        stream = np.array(enc.get_encoded_stream(), dtype=np.uint8)
        # stay inside the reference's own (buggy) scratch bound: bytes <= entries
        if len(stream) - 1 > ne:
            print("  regenerate: stream", len(stream), "entries", ne)
            continue
        break
    dec.set_stream(stream)
    got = [dec.decode_stream(np.ascontiguousarray(idx[a:b]), g) for a, b in zip(cuts[:-1], cuts[1:])]
    got = np.concatenate(got).astype(np.int16)
    exp = np.where(idx < 0, 0, sym).astype(np.int16)
    assert np.array_equal(got, exp), f"reference round trip failed for case {ci}"
    kats[f"sym_{ci}"] = sym
    kats[f"idx_{ci}"] = idx
    kats[f"cuts_{ci}"] = cuts
    kats[f"stream_{ci}"] = stream
    print(f"kat {ci} {kind} n={n} entries={ne} bytes={len(stream)}")
np.savez_compressed(os.path.join(OUT, "rans_kats.npz"), n=len(specs), **kats)

# ---- reference worked-example data files ----
for rel in ("IO/bitstreams/apple.c2df", "IO/clip_vecs/apple.npy", "IO/faiss/index.faiss", "IO/faiss/ids.txt"):
    shutil.copyfile(os.path.join(ref_harness.REF, rel), os.path.join(OUT, "ref_" + os.path.basename(rel)))
print("done")
