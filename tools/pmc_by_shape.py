"""Join the rocprofv3 --pmc passes of `bench.py --steps 1 --warmup 0 --no-secondary --no-cpu-baseline` with the
launch-ordered GEMM shape list bench.py writes (SGIC_BENCH_SHAPES) -> per-shape HBM-side counter bytes per launch
(FETCH_SIZE x2-corrected per MI355X_MICROARCH.md + WRITE_SIZE), next to the algorithmic bytes, plus the whole-step summary.
usage: python tools/pmc_by_shape.py gpurun_out/pmcC gpurun_out/shapes.json profiles/round3_pmc_gemm_by_shape.json profiles/round3_pmc_gemm_summary.json"""
import csv
import glob
import json
import os
import sys

root, shapes_json, out_shape, out_sum = sys.argv[1:5]
STEP_FIRST_KERNEL = sys.argv[5] if len(sys.argv) > 5 else "im2col_patch"   # first kernel of a timed step (decompress: pass its first kernel)
shapes = json.load(open(shapes_json))
n = len(shapes)


def is_s3(key):
    return len(key) > 5 and key[5] == "s3"


def dispatches(rec):
    """kernel dispatches behind one profile record: the split GEMM's modes 6 / 7 are two launches when the grid has whole rounds
    of the 256 CUs plus a tail (csrc/gemm_split.hip: s3_dispatch)"""
    key, mode = rec[0], (rec[3] if len(rec) > 3 else None)
    if is_s3(key) and mode == 31:
        tiles_n, tiles_m = (key[1] + 255) // 256, (key[0] + 255) // 256
        m_full = (tiles_m * tiles_n // 256) * 256 // tiles_n
        return 2 if (m_full > 0 and m_full * 256 < key[0]) else 1
    if not is_s3(key) or mode not in (6, 7, 8, 9, 12, 13, 26, 27):
        return 1
    M, N = key[0], key[1]
    tn = 256 if mode in (6, 8, 12, 26, 27) else 128
    tiles_n, tiles_m = (N + tn - 1) // tn, (M + 127) // 128
    m_full = (tiles_m * tiles_n // 256) * 256 // tiles_n
    return 2 if (m_full > 0 and m_full * 128 < M) else 1


GEMM_KERNELS = ("gemm_f32_kernel", "gemm_lat16_kernel", "conv3x3_thin", "gemm_split3_")
s3_share = sum(1 for r in shapes if is_s3(r[0])) / max(1, n)


def tail_signature(rec):
    """(kernel name fragment, grid size in threads) of the SECOND launch of a split-launch record, or None.  The second launch covers the
    rows beyond the whole rounds with 32x32 tiles (modes 8 / 9 / 12 / 13: 4 waves) or 64x128 tiles (6 / 7: 8 waves)"""
    key, mode = rec[0], (rec[3] if len(rec) > 3 else None)
    if is_s3(key) and mode == 31 and not (len(key) > 6 and key[6] == "conv"):   # 256x256 whole rounds + 128x256 tiles for the rest
        tiles_n, tiles_m = (key[1] + 255) // 256, (key[0] + 255) // 256
        m_full = (tiles_m * tiles_n // 256) * 256 // tiles_n
        if not (m_full > 0 and m_full * 256 < key[0]):
            return None
        return "gemm_split3_dma_kernel<2, 4, 4, 4, false", ((key[0] - m_full * 256 + 127) // 128) * tiles_n * 512
    if not is_s3(key) or mode not in (6, 7, 8, 9, 12, 13, 26, 27) or (len(key) > 6 and key[6] == "conv"):
        return None
    M, N = key[0], key[1]
    tn = 256 if mode in (6, 8, 12, 26, 27) else 128
    tiles_n, tiles_m = (N + tn - 1) // tn, (M + 127) // 128
    m_full = (tiles_m * tiles_n // 256) * 256 // tiles_n
    if not (m_full > 0 and m_full * 128 < M):
        return None
    rows = M - m_full * 128
    if mode in (26, 27):   # the ring kernel's 80x64 tiles
        return "gemm_split3_ring_kernel<1, 4, 5, 1,", ((rows + 79) // 80) * ((N + 63) // 64) * 256
    if mode in (6, 7):
        return "gemm_split3_kernel<2, 4, 2, 2,", ((rows + 63) // 64) * ((N + 127) // 128) * 512
    return "gemm_split3_kernel<2, 2, 1, 1,", ((rows + 31) // 32) * ((N + 31) // 32) * 256


def load(tag, counter):
    """counter value per profile record.  The timed step is located in the dispatch stream by its first kernel (the encoder's
    im2col_patch: a step launches it twice, encoder then CLIP tower), and every record takes one GEMM-class dispatch plus -- for a
    split-launch mode -- the following dispatch when its kernel AND grid are the expected tail launch (round 3: counting dispatches
    from the mode alone drifted by a few launches per step)."""
    rows = []
    for f in sorted(glob.glob(f"{root}_{tag}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1:]:   # the latest run only
        rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    starts = [i for i, r in enumerate(rows) if STEP_FIRST_KERNEL in r["Kernel_Name"]]
    start = starts[-2] if STEP_FIRST_KERNEL == "im2col_patch" and len(starts) >= 2 else (starts[-1] if starts else 0)
    g = [r for r in rows[start:] if any(k in r["Kernel_Name"] for k in GEMM_KERNELS)]
    out, p = [], 0
    for rec in shapes:
        assert p < len(g), (tag, "ran out of dispatches at record", len(out))
        v = float(g[p]["Counter_Value"])
        p += 1
        sig = tail_signature(rec)
        if sig is not None and p < len(g) and sig[0] in g[p]["Kernel_Name"] and int(g[p]["Grid_Size"]) == sig[1]:
            v += float(g[p]["Counter_Value"])
            p += 1
        out.append(v)
    if p != len(g):
        print(f"[pmc_by_shape] {tag}: {len(g) - p} GEMM-class dispatches of the step left unassigned (of {len(g)})", file=sys.stderr)
    return out


fetch, write = load("FETCH_SIZE", "FETCH_SIZE"), load("WRITE_SIZE", "WRITE_SIZE")
busy, act = load("MFMA", "SQ_VALU_MFMA_BUSY_CYCLES"), load("MFMA", "GRBM_GUI_ACTIVE")
assert len(fetch) == len(write) == len(busy) == n, (len(fetch), len(write), len(busy), n)
agg = {}
for i, rec in enumerate(shapes):
    key, fl, ms = rec[0], rec[1], rec[2]
    k = tuple(key[:3]) if key[0] != "batched" else tuple(key[2:5])
    a = agg.setdefault(k, dict(calls=0, fetch=0.0, write=0.0, busy=0.0, act=0.0, ms=0.0, flops=0.0, res=bool(key[3]) if key[0] != "batched" else False,
                               s3=is_s3(key), conv=len(key) > 6 and key[6] == "conv"))
    a["calls"] += 1
    a["fetch"] += fetch[i]
    a["write"] += write[i]
    a["busy"] += busy[i]
    a["act"] += act[i]
    a["ms"] += ms
    a["flops"] += fl
rows = []
for (M, N, K), a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
    # the split kernel reads its operands as three bf16 planes (6 B per element)
    # (an implicit-GEMM convolution reads its input once: M x Cin, not the M x 9 Cin im2col matrix)
    algo = (6.0 if a["s3"] else 4.0) * (M * (K // 9 if a["conv"] else K) + N * K) + 4.0 * M * N * (2 if a["res"] else 1)
    cnt = (2 * a["fetch"] + a["write"]) * 1024 / a["calls"]
    rows.append({"shape": [M, N, K], "kernel": "split3" if a["s3"] else "f32", "calls": a["calls"], "algorithmic_MB": round(algo / 1e6, 1), "counter_MB": round(cnt / 1e6, 1),
                 "counter_over_algorithmic": round(cnt / algo, 2), "fetch_MB_x2": round(2 * a["fetch"] * 1024 / a["calls"] / 1e6, 1),
                 "write_MB": round(a["write"] * 1024 / a["calls"] / 1e6, 1),
                 "mfma_busy_fraction": round(a["busy"] / (a["act"] / 8 * 1024), 4) if a["act"] else None})
json.dump({"command": "tools/pmc_collect.sh (three rocprofv3 --pmc passes over bench.py --steps 1 --warmup 0 --no-secondary --no-cpu-baseline) "
                      "+ tools/pmc_by_shape.py", "note": "FETCH_SIZE doubled (gfx950 reports half of a wide coalesced stream); Infinity-Cache "
                      "hits are counted; the Infinity Cache (256 MiB) holds the weights + activations of neighbouring launches, so counter bytes "
                      "above the algorithmic bytes are L2 misses served on-die, not HBM re-reads", "shapes": rows}, open(out_shape, "w"), indent=1)
fs, ws, bs, ga = sum(fetch), sum(write), sum(busy), sum(act)
summ = {"command": "rocprofv3 --pmc <counter set> --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-secondary --no-cpu-baseline "
                   "(three separate passes: FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE)",
        "kernel": "gemm_split3_kernel" if s3_share > 0.5 else "gemm_f32_kernel", "what": f"the {n} GEMM-class launches of the timed step", "launches": n,
        "hbm_fetch_MB_per_launch_x2_corrected": round(2 * fs * 1024 / n / 1e6, 2), "hbm_write_MB_per_launch": round(ws * 1024 / n / 1e6, 2),
        "algorithmic_MB_per_launch": round(sum(r["algorithmic_MB"] * r["calls"] for r in rows) / n, 2),
        "SQ_VALU_MFMA_BUSY_CYCLES": bs, "GRBM_GUI_ACTIVE": ga, "mfma_busy_fraction": round(bs / (ga / 8 * 1024), 4) if ga else None}
json.dump(summ, open(out_sum, "w"), indent=1)
print(json.dumps(summ, indent=1))
for r in rows[:12]:
    print(r)
