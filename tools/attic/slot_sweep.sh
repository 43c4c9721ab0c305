#!/bin/bash
# pipelined fp32 step (reference graph) vs forwards in flight and hardware queues
for s in 4 6 8 10 12 16; do echo -n "slots=$s "; python3 tools/step_time.py fp32 512 400 5 fold_fc=false slots=$s 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['sites_per_s_median'], d['ms_per_step'])"; done
for q in 2 3 6 8; do echo -n "GPU_MAX_HW_QUEUES=$q slots=8 "; GPU_MAX_HW_QUEUES=$q python3 tools/step_time.py fp32 512 400 5 fold_fc=false slots=8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['sites_per_s_median'], d['ms_per_step'])"; done
for q in 8; do for s in 12 16; do echo -n "GPU_MAX_HW_QUEUES=$q slots=$s "; GPU_MAX_HW_QUEUES=$q python3 tools/step_time.py fp32 512 400 5 fold_fc=false slots=$s 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['sites_per_s_median'], d['ms_per_step'])"; done; done
