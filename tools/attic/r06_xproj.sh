#!/bin/bash
# round 6: the BiLSTM cells' start-up, same box: scalar-path initial values (library default) against lstm_acc_init per tile
# (build/variants/lib_noscalar.so), lstm_xproj_kernel off / first step only (default) / all steps
mkdir -p gpurun_out/r06
for rep in 1 2; do
for lib in head noscalar; do
  if [ $lib = head ]; then unset DS_HIP_LIBRARY; else export DS_HIP_LIBRARY=$PWD/build/variants/lib_$lib.so; fi
  for xp in true; do
    echo "== $lib xproj=$xp"
    python3 tools/kernel_time.py bf16x3 512 10 5 lstm lstm_xproj=$xp 2>/dev/null | tail -1
    python3 tools/step_time.py bf16x3 512 100 5 lstm_xproj=$xp fold_fc=false 2>/dev/null | tail -1
    python3 tools/step_time.py bf16x3 512 100 5 lstm_xproj=$xp 2>/dev/null | tail -1
  done
done
done
