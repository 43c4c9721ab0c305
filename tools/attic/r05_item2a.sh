# item 2 probes: the fp32 headline (three-step, 512 sites) with the BiLSTM tile variants and slot counts
for args in "" "--lstm-tiling lds2" "--lstm-tiling narrow" "--slots 4" "--slots 6" "--slots 12" "--slots 16"; do
  python bench.py --no-cpu-baseline --no-configs2 --no-host-path --no-fast-mode --no-split --no-profile-pass --windows 3 $args 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$args', d['value'], d['windows']['sites_per_s'])"
done
