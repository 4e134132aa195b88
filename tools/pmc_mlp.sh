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
out = {k: {c: round(v / n[(k, c)]) for c, v in d.items()} for k, d in acc.items()}
for k, d in out.items():
    print(k, d)
import json, os
dst = os.path.join(os.path.dirname(sys.argv[1]), "..", "summary.jsonl")
open(dst, "a").write(json.dumps(out) + "\n")
PY
done
python - <<PY
import json, os
rows = [json.loads(l) for l in open("$OUT/summary.jsonl")]
merged = {}
for r in rows:
    for k, d in r.items():
        merged.setdefault(k, {}).update(d)
os.makedirs("$REPO/gpurun_out/profiles_r05", exist_ok=True)
json.dump({"source": "rocprofv3 --pmc (four passes) over python tools/bench_mlp.py 300000: per-launch averages of the per-Gaussian MLP chain "
                     "(k_mlp_chain<1, false>: static network, <2, false>: deformation network, inference; <.., true>: hidden activations kept)",
           "kernels": merged}, open("$REPO/gpurun_out/profiles_r05/r05_pmc_mlp.json", "w"), indent=1)
print("wrote r05_pmc_mlp.json", list(merged))
PY
