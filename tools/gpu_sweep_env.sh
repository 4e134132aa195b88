#!/bin/bash
# A/B of library experiment switches on the headline step, same box: tools/gpu_sweep_env.sh "<bench args>" "VAR=a VAR2=b" "VAR=c" ...
# (one bench.py --headline-only run per environment setting; prints steps/s and the per-label kernel times that moved)
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/sweep; mkdir -p $O
ARGS="$1"; shift
i=0
for envs in "" "$@"; do
  i=$((i+1))
  env $envs timeout 300 python bench.py --headline-only --no-cpu-baseline $ARGS > $O/run_$i.log 2>&1
  python - <<PY
import json
l=[x for x in open("$O/run_$i.log") if x.startswith('{"metric"')]
if l:
    d=json.loads(l[-1]); r=d["roofline"]; k=d["kernel_ms_per_step"]
    print("[%s] %.2f steps/s  mfma %.2f ms  sum %.2f  " % ("$envs" or "default", d["value"], r["mfma_all"]["ms_per_step"], sum(k.values())),
          {a: round(b,2) for a,b in sorted(k.items(), key=lambda x:-x[1])[:12]})
else:
    print("[$envs] FAILED", open("$O/run_$i.log").read()[-1500:])
PY
done
