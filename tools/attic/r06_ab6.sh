for r in 1 2 3; do for n in "$@"; do
  export DS_HIP_LIBRARY=$PWD/build/variants/lib_$n.so
  k=$(python3 tools/kernel_time.py bf16x3 512 10 5 lstm_cell 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(list(d['kernels'].values())[0]['median_us_per_step'])")
  a=$(python3 tools/step_time.py bf16x3 512 100 5 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['sites_per_s_median'])")
  echo "$n $r chain_us $k folded $a"
done; done
