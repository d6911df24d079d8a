#!/usr/bin/env python3
"""Condenses rocprofv3 outputs under gpurun_out/ into ONE committed json under profiles/ with, per kernel of interest:
average launch duration (from an un-profiled --kernel-trace --stats run), PMC counters per launch (from separate --pmc
passes, one counter group per pass as MI355X_MICROARCH.md prescribes) and the derived figures the roofline discussion
needs: HBM-side bytes per launch (FETCH_SIZE, raw and with the gfx950 x2 correction for wide coalesced reads, + WRITE_SIZE),
achieved GB/s against the 8 TB/s HBM peak, L2 hit rate, LDS bank-conflict share, MFMA busy share.

usage: tools/summarize_pmc.py <out.json> <stats_run> <pmc_run> [<pmc_run> ...] -- <kernel-substring> [...]"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
sep = args.index("--")
out_path, stats_run, pmc_runs, kernels = args[0], args[1], args[2:sep], args[sep + 1:]


def find(run, suffix):
    base = os.path.join(ROOT, "gpurun_out", run)
    for dp, _, fs in os.walk(base):
        for f in fs:
            if f.endswith(suffix):
                return os.path.join(dp, f)
    return None


stats = {}
sp = find(stats_run, "kernel_stats.csv")
if sp:
    for r in csv.DictReader(open(sp)):
        stats[r["Name"]] = r
counters = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))   # kernel -> run -> ctr
launches = collections.defaultdict(lambda: collections.defaultdict(set))
for run in pmc_runs:
    cp = find(run, "counter_collection.csv")
    if not cp:
        continue
    for r in csv.DictReader(open(cp)):
        for k in kernels:
            if k in r["Kernel_Name"]:
                counters[k][run][r["Counter_Name"]] += float(r["Counter_Value"])
                launches[k][run].add(r["Dispatch_Id"])
res = {"stats_run": stats_run, "pmc_runs": pmc_runs, "hbm_peak_GBps": 8000.0,
       "note": "FETCH_SIZE/WRITE_SIZE are KB at the L2's memory side (Infinity-Cache hits included); x2 applies to wide coalesced "
               "16 B/lane reads only (MI355X_MICROARCH.md §HBM), so both figures are given; durations come from the un-profiled run",
       "kernels": {}}
for k in kernels:
    e = {}
    match = [r for n, r in stats.items() if k in n]
    if match:
        calls = sum(int(r["Calls"]) for r in match)
        tot = sum(float(r["TotalDurationNs"]) for r in match)
        e["calls"], e["avg_us"] = calls, tot / max(calls, 1) / 1e3
    per = {}
    for run, cs in counters[k].items():
        n = max(len(launches[k][run]), 1)
        for c, v in cs.items():
            per[c] = v / n
        e.setdefault("launches_profiled", {})[run] = n
    e["per_launch"] = per
    dur = e.get("avg_us")
    if "FETCH_SIZE" in per:
        e["fetch_bytes_raw"] = per["FETCH_SIZE"] * 1024
        e["fetch_bytes_x2"] = 2 * per["FETCH_SIZE"] * 1024
    if "WRITE_SIZE" in per:
        e["write_bytes"] = per["WRITE_SIZE"] * 1024
    if dur and "fetch_bytes_raw" in e:
        w = e.get("write_bytes", 0.0)
        e["hbm_side_GBps_raw"] = (e["fetch_bytes_raw"] + w) / (dur * 1e-6) / 1e9
        e["hbm_side_GBps_x2"] = (e["fetch_bytes_x2"] + w) / (dur * 1e-6) / 1e9
        e["frac_of_hbm_peak_x2"] = e["hbm_side_GBps_x2"] / 8000.0
    hit, miss = per.get("TCC_HIT_sum"), per.get("TCC_MISS_sum")
    if hit is not None and miss is not None and hit + miss > 0:
        e["l2_hit_rate"] = hit / (hit + miss)
        e["l2_requests"] = hit + miss
    if per.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict_share"] = per.get("SQ_LDS_BANK_CONFLICT", 0.0) / per["SQ_LDS_IDX_ACTIVE"]
    if per.get("SQ_WAVE_CYCLES"):
        wc = per["SQ_WAVE_CYCLES"]
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if c in per:
                e[c + "_frac"] = per[c] / wc
        if "SQ_VALU_MFMA_BUSY_CYCLES" in per:
            # per WAVE-cycle: saturates at 1 / (waves resident per SIMD) — right for the one-wave-per-SIMD MLP kernels,
            # misleading for kernels that keep 2-3 waves per SIMD (k_wgrad, k_mlp_fwd_n); the two figures below are per SIMD-time
            e["mfma_busy_over_wave_cycles"] = per["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * wc)
            if dur:
                e["mfma_busy_over_simd_time_at_2p4GHz"] = per["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (dur * 2400.0)
            if per.get("GRBM_GUI_ACTIVE"):
                # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (k_mlp_fwd_l, one wave per SIMD for the whole launch, calibrates it:
                # 0.86 this way against 0.88 per wave-cycle); while a second stream runs other kernels it counts their cycles too
                e["mfma_busy_over_kernel_cycles"] = per["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * per["GRBM_GUI_ACTIVE"] / 8.0)
        if per.get("SQ_INSTS_VALU_MFMA_MOPS_F16") or per.get("SQ_INSTS_MFMA"):
            pass
    res["kernels"][k] = e
os.makedirs(os.path.dirname(os.path.join(ROOT, out_path)), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, out_path), "w"), indent=1)
print(json.dumps({k: {a: b for a, b in v.items() if a != "per_launch"} for k, v in res["kernels"].items()}, indent=1))
