#!/bin/bash
# true kernel durations (rocprofv3 --kernel-trace --stats) of one tools/gemm_probe.py case under environment settings:
#   bash tools/probe_rocprof.sh <PROBE_ONLY pattern> "ENV=1" "ENV=2 OTHER=3" ...
cd /tmp && export TMPDIR=/tmp
pat="$1"; shift
for e in "$@"; do
  d=/tmp/pr_$RANDOM
  env $e PROBE_ONLY="$pat" rocprofv3 --kernel-trace --stats -d $d -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/gemm_probe.py f32x > /dev/null 2>&1
  echo "== $e"
  python - "$d" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "gemm" in n or "conv" in n or "splitk" in n:
            print("   %-60s x%-4s avg %8.1f us  min %8.1f" % (n[:60].replace("(anonymous namespace)::", ""), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
