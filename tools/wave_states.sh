#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/ws; mkdir -p $O

# usage: tools/wave_states.sh ["<probe_engine args>" ...]   (default: the bf16_all / 4096, fp32 / 512 and bf16x3 / 512 engines)
if [ $# -eq 0 ]; then set -- "bf16_all 4096" "fp32 512 threestep" "bf16x3 512 threestep"; fi
for cfg in "$@"; do
  tag=$(echo $cfg | tr ' ' '_')
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $O/${tag}_p1 -- python3 tools/probe_engine.py $cfg > /dev/null 2> $O/${tag}_p1.err
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $O/${tag}_p2 -- python3 tools/probe_engine.py $cfg > /dev/null 2> $O/${tag}_p2.err
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_ADDR_CONFLICT --output-format csv -d $O/${tag}_p3 -- python3 tools/probe_engine.py $cfg > /dev/null 2> $O/${tag}_p3.err
  python3 tools/pmc_wave_states.py $O/${tag}_p1 $O/${tag}_p2 $O/${tag}_p3 > $O/${tag}.txt 2>&1
  cat $O/${tag}.txt
  # (profiler logs go to stderr, not into the summary the caller redirects to profiles/<tag>_wave_states.txt)
  tail -n 3 $O/${tag}_p2.err $O/${tag}_p3.err 1>&2
  rm -rf $O/${tag}_p1 $O/${tag}_p2 $O/${tag}_p3
done

