#!/bin/bash
# Condenses the gpurun_out/r2_* runs of `tools/r2_measure.sh bench strong pmc pmct` into the committed profiles/round2_* files.
# Runs in the build container after the gpurun call has merged its outputs.
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
tail -1 $O/r2_bench_default.json > $P/round2_bench_line.json
short_stats r2_stats_render $P/round2_kernel_stats.csv
short_stats r2_stats_trans $P/round2_transition_kernel_stats.csv
python tools/summarize_pmc.py $P/round2_kernels_pmc.json r2_stats_render r2_pmc_fetch_render r2_pmc_write_render r2_pmc_sq_render -- \
    k_mlp_fwd_l k_search k_features k_classify k_composite k_importance
python tools/summarize_pmc.py $P/round2_transition_pmc.json r2_stats_trans r2_pmc_fetch_trans r2_pmc_write_trans r2_pmc_sq_trans -- \
    k_cconv_gemm k_cconv_gather k_trans_prepare k_trans_search k_trans_conv0
if [ -d $O/r2_stats_train ]; then
  short_stats r2_stats_train $P/round2_train_kernel_stats.csv
  python tools/summarize_pmc.py $P/round2_train_pmc.json r2_stats_train r2_pmc_fetch_train r2_pmc_write_train r2_pmc_sq_train -- \
      k_mlp_fwd_n k_mlp_bwd "k_wgrad(" k_search k_composite_bwd_w
fi
python - <<'PY'
import json
d = json.load(open("profiles/round2_kernels_pmc.json"))
k = d["kernels"]["k_mlp_fwd_l"]
out = {"kernel": "k_mlp_fwd_l", "launches_profiled": min(k["launches_profiled"].values()),
       "source": "profiles/round2_kernels_pmc.json (separate rocprofv3 --pmc passes of `bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1`)",
       "FETCH_SIZE_KB_per_launch_raw": k["per_launch"]["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch_raw": k["per_launch"]["WRITE_SIZE"],
       "gfx950_fetch_correction": "x2 (MI355X_MICROARCH.md: FETCH_SIZE reports 1/2 of wide coalesced 16 B/lane reads)",
       "hbm_bytes_per_launch": k["fetch_bytes_x2"] + k["write_bytes"], "mfma_busy_over_wave_cycles": k["mfma_busy_over_wave_cycles"],
       "SQ_WAIT_ANY_frac": k["SQ_WAIT_ANY_frac"], "avg_launch_us_unprofiled_run": k["avg_us"]}
json.dump(out, open("profiles/round2_mlp_pmc.json", "w"), indent=1)
PY
for img in 400 800; do for n in 1 2 4 8; do [ -s $O/r2_strong_${img}_w${n}.json ] && cp $O/r2_strong_${img}_w${n}.json $P/round2_strong_${img}_w${n}.json; done; done
ls -la $P | grep round2
