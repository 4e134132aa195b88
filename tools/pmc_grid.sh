#!/bin/bash
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/pmc_grid
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*\|SQ_[A-Z0-9_]*" | sort -u > $OUT/counters.txt
grep -c . $OUT/counters.txt
grep "ATOMIC\|TCC_EA0_WRREQ\|TCC_REQ\|TCC_HIT\|TCC_MISS\|TCC_TAG_STALL\|TCC_BUSY" $OUT/counters.txt | head -40
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/sq -o g -- python $REPO/tools/pmc_grid.py > $OUT/sq.log 2>&1
timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $OUT/tcc -o g -- python $REPO/tools/pmc_grid.py > $OUT/tcc.log 2>&1
cd $REPO
python - <<'PY'
import csv, glob, collections
for d in ("sq", "tcc"):
    fs = glob.glob("gpurun_out/pmc_grid/%s/**/*counter_collection.csv" % d, recursive=True)
    if not fs: print(d, "no counters", open("gpurun_out/pmc_grid/%s.log" % d).read()[-600:]); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][-40:]
        if "grid" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        print(d, k, {a: "%.3g" % b for a, b in v.items()})
PY
find $OUT -name "*.csv" -size +4M -delete
