python -m pytest tests/test_gpu_parity.py -q -x -k "tilings" 2>&1 | tail -2
for args in "" "--lstm-tiling lds22" "" "--lstm-tiling lds22"; do
  python bench.py --no-cpu-baseline --no-configs2 --no-host-path --no-split --no-profile-pass --windows 3 $args 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$args', d['value'], d['windows']['sites_per_s'], 'folded', d['fast_mode_folded']['value'])"
done
python tools/kernel_time.py fp32 512 10 5 "lstm" lstm_tiling=lds22 2>/dev/null
