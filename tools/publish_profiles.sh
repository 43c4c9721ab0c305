#!/bin/bash
# Copy what tools/collect_profiles.sh <tag> left under gpurun_out/prof_<tag>/ into the tracked profiles/<tag>_* files
# (gpurun_out/ is scratch; the judge reads profiles/). Run HERE after the gpurun call has merged its output back.
set -eu
R=${1:-r05}
P=gpurun_out/prof_$R
mkdir -p profiles/${R}_pmc
cp $P/bench.json profiles/${R}_bench.json
cp $P/kernel_stats.csv profiles/${R}_bench_kernel_stats.csv
cp $P/bench_under_rocprof.json profiles/${R}_bench_under_rocprof.json
cp $P/serial_kernel_stats.csv profiles/${R}_serial_kernel_stats.csv
cp $P/pmc_traffic.json profiles/${R}_pmc_traffic.json
cp $P/pmc_mfma_util.json profiles/${R}_pmc_mfma_util.json
cp $P/fetch_size_counter_collection.csv $P/write_size_counter_collection.csv $P/mfma_busy_counter_collection.csv profiles/${R}_pmc/
cp $P/bf16_fetch_size_counter_collection.csv $P/bf16_write_size_counter_collection.csv profiles/${R}_pmc/
cp $P/bf16_all_4096_kernel_stats.csv profiles/${R}_bf16_all_4096_kernel_stats.csv
cp $P/bf16_all_4096_kernel_times.json profiles/${R}_bf16_all_4096_kernel_times.json
cp $P/bf16_all_4096_pmc_traffic.json profiles/${R}_bf16_all_4096_pmc_traffic.json
cp $P/config3_batch4096.json profiles/${R}_config3_batch4096.json
cp $P/config3_batch512.json profiles/${R}_config3_batch512.json
cp $P/config4_1gpu.jsonl profiles/${R}_config4_1gpu.jsonl
for f in bf16x3_serial_kernel_stats.csv bf16x3_pmc_traffic.json bf16x3_pmc_mfma_util.json wave_states.txt; do
  [ -f $P/$f ] && cp $P/$f profiles/${R}_$f
done
[ -f gpurun_out/stress_parity.json ] && cp gpurun_out/stress_parity.json profiles/${R}_stress_parity.json
[ -f gpurun_out/stress_bf16_tolerance.json ] && cp gpurun_out/stress_bf16_tolerance.json profiles/${R}_stress_bf16_tolerance.json
python3 - "$R" <<'PY'
import json, sys
r = sys.argv[1]
t = json.load(open("profiles/%s_pmc_traffic.json" % r))
print("published profiles/%s_*: collected at commit %s, sources %s" % (r, t.get("commit"), t.get("sources_sha16")))
PY
