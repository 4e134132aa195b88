#!/bin/bash
# A/B of environment switches on the headline step:  bash tools/env_sweep.sh "A=1" "B=2 C=3" ...   (each argument: one configuration)
# prints steps/s and ms/step of `bench.py --headline-only` per configuration (same box, back to back; "base" first and last)
run() { env $1 python bench.py --headline-only --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-60s %.2f steps/s  %.3f ms' % (sys.argv[1], d['value'], d['ms_per_step']))" "$1"; }
run "DWG_BASE=1"
for c in "$@"; do run "$c"; done
run "DWG_BASE=2"
