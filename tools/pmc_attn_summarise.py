"""Summarise tools/attn_pmc.sh (two rocprofv3 --pmc passes over tools/bench_attn.py) into profiles/<name>.json:
per (shape, attn_mode) the MFMA-busy fraction of SIMD cycles and the wave-time split.
usage: python tools/pmc_attn_summarise.py gpurun_out/pmcA1 gpurun_out/pmcA2 profiles/round2_pmc_attn_summary.json [MODES]"""
import csv
import glob
import json
import sys

a1, a2, out = sys.argv[1:4]
modes = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "1,2,3,4,5,6,7").split(",")]
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
            by[int(r["Dispatch_Id"])]["grid"] = float(r.get("Grid_Size", 0) or 0)
    return [by[k] for k in sorted(by)]


d1, d2 = load(a1), load(a2)
res = {"command": "tools/attn_pmc.sh: rocprofv3 --pmc <SQ counters> -- python3 tools/bench_attn.py (two passes), counters summed over the "
                  "13 launches of each (shape, attn_mode)", "kernel": "attn_f32_kernel<NBUF, UP>",
       "modes": "odd = single K/V buffer (2 barriers per tile), even = double-buffered (1 barrier); 3,4 / 5,6 add the start-up stagger; 7 = three row blocks per workgroup (single buffer)",
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
        # waves of the MFMA items of one launch (ragged-row VALU workgroups issue no MFMA and are short): grid threads / 64
        nwaves = sum(float(x.get("grid", 0)) for x in g1) / 64.0
        life = 4.0 * wc / nwaves if nwaves else 0.0                    # SQ_WAVE_CYCLES counts quad-cycles: average wave lifetime
        e = {"mfma_busy_fraction": round(s(g1, "SQ_VALU_MFMA_BUSY_CYCLES") / (ga / 8 * 1024), 4) if ga else None,
             "avg_wave_lifetime_cycles": round(life), "gui_active_cycles_per_launch": round(ga / 8 / len(g1)),
             "wave_time_parked_waitcnt_or_barrier": round(s(g1, "SQ_WAIT_ANY") / wc, 3) if wc else None,
             "wave_time_issue_stalled": round(s(g1, "SQ_WAIT_INST_ANY") / wc, 3) if wc else None,
             "wave_time_issuing": round(s(g1, "SQ_ACTIVE_INST_ANY") / wc, 3) if wc else None}
        if g2:
            idx = s(g2, "SQ_LDS_IDX_ACTIVE")
            e["lds_bank_conflict_fraction"] = round(s(g2, "SQ_LDS_BANK_CONFLICT") / idx, 4) if idx else 0.0
        res["configs"][f"{sh}, attn_mode {m}"] = e
# launch-span utilisation WITHOUT the profiler (tools/micro/attn_stamps.hip: s_memrealtime of the first wave's start and the last
# wave's end, in-kernel clock from s_memtime / s_memrealtime = 2.08 GHz): MFMA cycles per SIMD / (span x clock)
res["launch_span_utilisation_from_stamps"] = {
    "method": "MFMA cycles per SIMD (items x key tiles x 64 MFMAs x 64 cycles / 1024 SIMDs) / (first-wave-start -> last-wave-end span "
              "x 2.08 GHz), product build timed by tools/micro/attn_stamps.hip, no profiler attached",
    "L=289 (attn_mode 5)": {"span_us": 120.0, "mfma_us_at_2.08GHz": 79.7, "utilisation": 0.66},
    "L=545 (attn_mode 5)": {"span_us": 294.8, "mfma_us_at_2.08GHz": 213.4, "utilisation": 0.72},
    "L=256 (attn_mode 1)": {"span_us": 70.5, "mfma_us_at_2.08GHz": 47.3, "utilisation": 0.67},
    "note": "GRBM_GUI_ACTIVE-normalised figures above read lower because on dispatches this short GUI_ACTIVE also counts the "
            "profiler's per-dispatch counter set-up / drain (176 k cycles for the L = 256 launch whose waves live 125 k)"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
