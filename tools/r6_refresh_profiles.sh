#!/bin/bash
# Condenses the gpurun_out/r6_* runs of `tools/r6_measure.sh bench strong pmc pmct` into the committed profiles/round6_* files.
set -eu
cd "$(dirname "$0")/.."
O=gpurun_out
P=profiles
short_stats() {   # <stats run> <out csv>
python - "$1" "$2" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(f"gpurun_out/{sys.argv[1]}/p_kernel_stats.csv")))
with open(sys.argv[2], "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r["Name"].split("(")[0][:80], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
PY
}
tail -1 $O/r6_bench_default.json > $P/round6_bench_line.json
short_stats r6_stats_render $P/round6_kernel_stats.csv
python tools/summarize_pmc.py $P/round6_kernels_pmc.json r6_stats_render r6_pmc_fetch_render r6_pmc_write_render r6_pmc_sq_render -- \
    k_mlp_fwd_a k_search k_features k_classify k_composite k_importance
for mode in fp32 split; do
  short_stats r6_stats_trans_$mode $P/round6_transition_${mode}_kernel_stats.csv
  python tools/summarize_pmc.py $P/round6_transition_${mode}_pmc.json r6_stats_trans_$mode r6_pmc_fetch_trans_$mode r6_pmc_write_trans_$mode r6_pmc_sq_trans_$mode -- \
      "k_cconv_gf<" "k_cconv_gf_epi(" k_cconv_gf_epi_g3 k_trans_stage1 k_trans_front k_cconv3_gather > /dev/null
done
cp $P/round6_transition_fp32_pmc.json $P/round6_transition_pmc.json
cp $P/round6_transition_fp32_kernel_stats.csv $P/round6_transition_kernel_stats.csv
if [ -d $O/r6_stats_train ]; then
  short_stats r6_stats_train $P/round6_train_kernel_stats.csv
  python tools/summarize_pmc.py $P/round6_train_pmc.json r6_stats_train r6_pmc_fetch_train r6_pmc_write_train r6_pmc_sq_train -- \
      k_mlp_fwd_n k_mlp_bwd_n k_wgrad3 k_wgrad_reduce k_search k_composite_bwd_w k_gemm_f32 > /dev/null
fi
if [ -d $O/r6_stats_e2e ]; then short_stats r6_stats_e2e $P/round6_e2e_kernel_stats.csv; fi
[ -s $O/r6_mfma_peak.txt ] && cp $O/r6_mfma_peak.txt $P/round6_mfma_peak.txt
python - <<'PY'
import json
d = json.load(open("profiles/round6_kernels_pmc.json"))
k = d["kernels"]["k_mlp_fwd_a"]
out = {"kernel": "k_mlp_fwd_a", "kernel_source_sha1": __import__("bench").kernel_source_sha1("k_mlp_fwd_a"), "launches_profiled": min(k["launches_profiled"].values()),
       "source": "profiles/round6_kernels_pmc.json (separate rocprofv3 --pmc passes of `bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 3`)",
       "FETCH_SIZE_KB_per_launch_raw": k["per_launch"]["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch_raw": k["per_launch"]["WRITE_SIZE"],
       "gfx950_fetch_correction": "x2 (MI355X_MICROARCH.md: FETCH_SIZE reports 1/2 of wide coalesced 16 B/lane reads)",
       "hbm_bytes_per_launch": k["fetch_bytes_x2"] + k["write_bytes"], "mfma_busy_over_wave_cycles": k["mfma_busy_over_wave_cycles"],
       "SQ_WAIT_ANY_frac": k["SQ_WAIT_ANY_frac"], "avg_launch_us_unprofiled_run": k["avg_us"]}
json.dump(out, open("profiles/round6_mlp_pmc.json", "w"), indent=1)
for mode in ("fp32", "split"):
    t = json.load(open(f"profiles/round6_transition_{mode}_pmc.json"))["kernels"]
    steps = t["k_trans_stage1"]["calls"]
    tot = sum((e.get("fetch_bytes_x2", 0) + e.get("write_bytes", 0)) * e.get("calls", 0) / steps for e in t.values())
    us = {k: round(e.get("avg_us", 0) * e.get("calls", 0) / steps, 1) for k, e in t.items()}
    print(mode, "HBM-side MB / step (fetch x2 + write):", round(tot / 1e6, 1), " kernel us / step:", us, " sum", round(sum(us.values()), 1))
PY
for img in 400 800; do for n in 2 8; do [ -s $O/r6_strong_${img}_w${n}.json ] && cp $O/r6_strong_${img}_w${n}.json $P/round6_strong_${img}_w${n}.json; done; done
ls $P | grep round6
