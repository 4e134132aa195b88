#!/bin/bash
# SQ counters of the fused f32x attention kernels at the step's shapes (tools/bench_attn.py): where the waves' time goes.
#   bash tools/pmc_attn.sh        -> gpurun_out/pmc_attn/{v1,v2}_{a,b}.txt
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/pmc_attn; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for V in 0 1; do
  DWG_ATTN_V2=$V timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/a$V -o a -- python $REPO/tools/bench_attn.py > $OUT/a$V.log 2>&1
  DWG_ATTN_V2=$V timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/b$V -o b -- python $REPO/tools/bench_attn.py > $OUT/b$V.log 2>&1
done
cd $REPO
python3 - <<'PY'
import csv, glob, collections, re
for tag in ("a0", "a1", "b0", "b1"):
    fs = glob.glob("gpurun_out/pmc_attn/%s/**/*counter_collection.csv" % tag, recursive=True)
    if not fs: print(tag, "no counters"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
    for r in csv.DictReader(open(fs[0])):
        k = re.sub(r"\(anonymous namespace\)::|^void |\(.*$", "", r["Kernel_Name"])
        if "flash" not in k: continue
        k = k + " grid %s" % r.get("Grid_Size", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen: seen.add((k, r["Dispatch_Id"])); n[k] += 1
    for k, c in agg.items():
        print(tag, "%-50s x%-3d" % (k[:50], n[k]), "  ".join("%s=%.3g" % (a, b / n[k]) for a, b in sorted(c.items())))
PY
