#!/bin/bash
# SQ counters of the per-Gaussian MLP chain alone (tools/bench_mlp.py):  bash tools/pmc_mlp.sh
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/pmc_mlp
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $OUT/sq_counters.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/s$i -o m -- python $REPO/tools/bench_mlp.py 300000 > $OUT/s$i.log 2>&1
  f=$(find $OUT/s$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "k_mlp_chain<" not in k: continue
    k = k[k.index("k_mlp_chain<"):].split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    print(k, {c: round(v / n[(k, c)]) for c, v in d.items()})
PY
done
