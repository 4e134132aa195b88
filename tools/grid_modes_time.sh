#!/bin/bash
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/grid_modes
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for m in slabs owner copies device; do
  DWG_GRID_XCD_MODE=$m timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$m -o g -- python $REPO/tools/pmc_grid.py > $OUT/$m.log 2>&1
  echo "mode=$m $(tail -1 $OUT/$m.log | cut -c1-60)"
  python - "$OUT/$m/g_kernel_stats.csv" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'grid' in r['Name'] or 'k_gs' in r['Name'] or 'xcd' in r['Name']:
        print("   %-46s calls %3s avg %8.1f us" % (r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:46], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
find $OUT -name "*trace.csv" -delete
