#!/bin/bash
# Per-stage times of the whole replan under a knob (development aid):
#   tools/pipe_stages.sh [ENV=VALUE ...]   e.g. tools/pipe_stages.sh MP_RL_THREADS=128
for kv in "$@"; do export "$kv"; done
timeout -k 10 150 python3 bench.py --workload pipeline --steps 10 --warmup 2 --reps 3 --cpu-sample 0 --single-plan 0 --streams 1 2>/dev/null |
python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('one stream %.4f ms; ' % d['ms_per_step'] + ', '.join('%s %.3f' % (k.split(' (')[0][:22], v) for k, v in d['stage_ms'].items()))
"
