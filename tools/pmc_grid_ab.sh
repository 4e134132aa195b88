#!/bin/bash
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/pmc_grid_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  DWG_GRID_PREFETCH=$v timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t$v -o g -- python $REPO/tools/pmc_grid.py > $OUT/t$v.log 2>&1
  echo "prefetch=$v"; grep "grid_bwd" $OUT/t$v/g_kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,150-
  DWG_GRID_PREFETCH=$v timeout 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $OUT/c$v -o g -- python $REPO/tools/pmc_grid.py > $OUT/c$v.log 2>&1
done
cd $REPO
python - <<'PY'
import csv, collections
for v in (0, 1):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    try:
        for r in csv.DictReader(open("gpurun_out/pmc_grid_ab/c%d/g_counter_collection.csv" % v)):
            if "k_grid_bwd_owner" in r["Kernel_Name"]:
                agg["owner"][r["Counter_Name"]].append(float(r["Counter_Value"]))
    except Exception as e:
        print(v, "ERR", e); continue
    print("prefetch", v, {a: "%.3g" % (sum(b) / len(b)) for a, b in agg["owner"].items()})
PY
find $OUT -name "*.csv" -size +4M -delete
