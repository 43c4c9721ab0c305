#!/bin/bash
# Collect the per-round rocprofv3 evidence on the GPU box (run from the repo root under gpurun):
#   tools/collect_profiles.sh r03      -> gpurun_out/prof_r03/{bench.json, kernel_stats.csv, fetch/, write/, mfma/}
# Counter passes are separate runs with --pmc only (no trace domains), as the pool requires.
set -u
R=${1:-r06}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline --no-host-path --no-fast-mode --no-configs2 --no-standalone-pass > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv 2>/dev/null
# stand-alone kernel durations (every launch on one stream): the figures bench.py's roofline leads with must agree with this CSV
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/strace -- python3 tools/probe_engine.py fp32 512 threestep reps=40 > /dev/null 2> $OUT/strace.err
cp $(ls $OUT/strace/*/*kernel_stats.csv | head -1) $OUT/serial_kernel_stats.csv 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 4 --warmup 1 --windows 1 --no-cpu-baseline --no-host-path --no-fast-mode --no-configs2 --no-profile-pass > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 4 --warmup 1 --windows 1 --no-cpu-baseline --no-host-path --no-fast-mode --no-configs2 --no-profile-pass > /dev/null 2> $OUT/write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -- python3 tools/probe_engine.py fp32 512 threestep > /dev/null 2> $OUT/mfma.err
python3 tools/pmc_traffic.py $OUT/fetch $OUT/write $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt 2>&1
python3 tools/pmc_mfma_util.py $OUT/mfma $OUT/pmc_mfma_util.json "python3 tools/probe_engine.py fp32 512 threestep" > $OUT/pmc_mfma_util.txt 2>&1
cp $(ls $OUT/fetch/*/*counter_collection.csv | head -1) $OUT/fetch_size_counter_collection.csv 2>/dev/null
cp $(ls $OUT/write/*/*counter_collection.csv | head -1) $OUT/write_size_counter_collection.csv 2>/dev/null
cp $(ls $OUT/mfma/*/*counter_collection.csv | head -1) $OUT/mfma_busy_counter_collection.csv 2>/dev/null
# BASELINE configs[2] (bf16 modes, batch 4096 and 512): throughput + per-kernel times, and the conv path's HBM-side bytes
python3 tools/config3.py 4096 $OUT/config3_batch4096.json > /dev/null 2> $OUT/config3.err
python3 tools/config3.py 512 $OUT/config3_batch512.json > /dev/null 2>> $OUT/config3.err
python3 tools/kernel_time.py bf16_all 4096 10 7 > $OUT/bf16_all_4096_kernel_times.json 2>> $OUT/config3.err
# >= 20 WARM serial forwards (round 4 kept two cold ones: 203 us per fused launch against 189 warm), so that conv_path_hbm.frac is
# reproducible from this file
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/btrace -- python3 tools/probe_engine.py bf16_all 4096 reps=24 > /dev/null 2> $OUT/btrace.err
cp $(ls $OUT/btrace/*/*kernel_stats.csv | head -1) $OUT/bf16_all_4096_kernel_stats.csv 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/bfetch -- python3 tools/probe_engine.py bf16_all 4096 > /dev/null 2> $OUT/bfetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/bwrite -- python3 tools/probe_engine.py bf16_all 4096 > /dev/null 2> $OUT/bwrite.err
python3 tools/pmc_traffic.py $OUT/bfetch $OUT/bwrite $OUT/bf16_all_4096_pmc_traffic.json > $OUT/bf16_all_4096_pmc_traffic.txt 2>&1
cp $(ls $OUT/bfetch/*/*counter_collection.csv | head -1) $OUT/bf16_fetch_size_counter_collection.csv 2>/dev/null
cp $(ls $OUT/bwrite/*/*counter_collection.csv | head -1) $OUT/bf16_write_size_counter_collection.csv 2>/dev/null
# DS_PRECISION_BF16X3 (round 5): stand-alone durations of the split-operand engine, its HBM-side bytes and MFMA-pipe busy share
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/xtrace -- python3 tools/probe_engine.py bf16x3 512 threestep reps=40 > /dev/null 2> $OUT/xtrace.err
cp $(ls $OUT/xtrace/*/*kernel_stats.csv | head -1) $OUT/bf16x3_serial_kernel_stats.csv 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/xfetch -- python3 tools/probe_engine.py bf16x3 512 threestep > /dev/null 2> $OUT/xfetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/xwrite -- python3 tools/probe_engine.py bf16x3 512 threestep > /dev/null 2> $OUT/xwrite.err
python3 tools/pmc_traffic.py $OUT/xfetch $OUT/xwrite $OUT/bf16x3_pmc_traffic.json > $OUT/bf16x3_pmc_traffic.txt 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/xmfma -- python3 tools/probe_engine.py bf16x3 512 threestep > /dev/null 2> $OUT/xmfma.err
python3 tools/pmc_mfma_util.py $OUT/xmfma $OUT/bf16x3_pmc_mfma_util.json "python3 tools/probe_engine.py bf16x3 512 threestep" > $OUT/bf16x3_pmc_mfma_util.txt 2>&1
rm -rf $OUT/xtrace $OUT/xfetch $OUT/xwrite $OUT/xmfma
cat $OUT/bf16x3_pmc_traffic.txt $OUT/bf16x3_pmc_mfma_util.txt
# wave states of the kernels that changed this round (and the unchanged ones beside them)
bash tools/wave_states.sh > $OUT/wave_states.txt 2>&1
# BASELINE configs[3] at its stated size on this build, one GPU: 10 M sites sustained, fp32 / 512 and bf16_all / 4096
python3 tools/config4.py --sites 10000000 --batch 512 --precision fp32 --telemetry > $OUT/config4_1gpu.jsonl 2> $OUT/config4.err
python3 tools/config4.py --sites 10000000 --batch 4096 --precision bf16_all --telemetry >> $OUT/config4_1gpu.jsonl 2>> $OUT/config4.err
rm -rf $OUT/strace $OUT/trace $OUT/fetch $OUT/write $OUT/mfma $OUT/btrace $OUT/bfetch $OUT/bwrite
cat $OUT/bf16_all_4096_pmc_traffic.txt | head -12
ls -la $OUT
cat $OUT/pmc_mfma_util.txt
cat $OUT/pmc_traffic.txt | head -20
head -c 600 $OUT/bench.json
