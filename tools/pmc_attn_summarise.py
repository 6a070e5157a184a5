"""Summarise tools/attn_pmc.sh (two rocprofv3 --pmc passes over tools/bench_attn.py) into profiles/<name>.json:
per (shape, attn_mode) the MFMA-busy fraction of SIMD cycles and the wave-time split.
usage: python tools/pmc_attn_summarise.py gpurun_out/pmcA1 gpurun_out/pmcA2 profiles/round2_pmc_attn_summary.json [MODES]"""
import csv
import glob
import json
import sys

a1, a2, out = sys.argv[1:4]
modes = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "1,2,3,4,5,6").split(",")]
SHAPES = ["L=289, 16 heads, 32 sequences", "L=545, 12 heads, 32 sequences", "L=256 Swin window (+bias), 12 heads, 32 sequences",
          "L=50, 12 heads, 32 sequences (CLIP)"]
PER = 13   # launches per (shape, mode): 3 warm-up + 10 timed


def load(root):
    by = {}
    for f in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "attn_f32_kernel" not in r["Kernel_Name"]:
                continue
            by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    return [by[k] for k in sorted(by)]


d1, d2 = load(a1), load(a2)
res = {"command": "tools/attn_pmc.sh: rocprofv3 --pmc <SQ counters> -- python3 tools/bench_attn.py (two passes), counters summed over the "
                  "13 launches of each (shape, attn_mode)", "kernel": "attn_f32_kernel<NBUF, UP>",
       "modes": "odd = single K/V buffer (2 barriers per tile), even = double-buffered (1 barrier); 3,4 / 5,6 add the start-up stagger",
       "configs": {}}
i = 0
for sh in SHAPES:
    for m in modes:
        g1, g2 = d1[i:i + PER], d2[i:i + PER]
        i += PER
        if len(g1) < PER:
            continue
        s = lambda g, k: sum(x.get(k, 0.0) for x in g)
        wc = s(g1, "SQ_WAVE_CYCLES")
        ga = s(g1, "GRBM_GUI_ACTIVE")
        e = {"mfma_busy_fraction": round(s(g1, "SQ_VALU_MFMA_BUSY_CYCLES") / (ga / 8 * 1024), 4) if ga else None,
             "wave_time_parked_waitcnt_or_barrier": round(s(g1, "SQ_WAIT_ANY") / wc, 3) if wc else None,
             "wave_time_issue_stalled": round(s(g1, "SQ_WAIT_INST_ANY") / wc, 3) if wc else None,
             "wave_time_issuing": round(s(g1, "SQ_ACTIVE_INST_ANY") / wc, 3) if wc else None}
        if g2:
            idx = s(g2, "SQ_LDS_IDX_ACTIVE")
            e["lds_bank_conflict_fraction"] = round(s(g2, "SQ_LDS_BANK_CONFLICT") / idx, 4) if idx else 0.0
        res["configs"][f"{sh}, attn_mode {m}"] = e
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
