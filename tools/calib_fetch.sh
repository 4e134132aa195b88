#!/bin/bash
# FETCH_SIZE against known byte counts (tools/calib_fetch.hip):  bash tools/calib_fetch.sh r05  ->  profiles/r05_fetch_calibration.json
R=${1:-r05}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/calib_fetch; mkdir -p $OUT $REPO/profiles
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $REPO/tools/calib_fetch.hip -o /tmp/calib_fetch || exit 1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc -o cal -- /tmp/calib_fetch > $OUT/run.log 2>&1
cd $REPO
python - "$OUT" "profiles/${R}_fetch_calibration.json" <<'PY'
import csv, glob, json, sys, collections
out, dst = sys.argv[1], sys.argv[2]
useful = json.loads([l for l in open(out + "/run.log") if l.startswith("{")][-1])
f = glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "FETCH_SIZE":
        agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]) * 1024)
res = {}
for k, v in agg.items():
    rep = sorted(v)[len(v) // 2]
    u = useful.get(k + "_useful")
    res[k] = {"fetch_size_bytes_reported": rep, "useful_bytes": u, "reported_over_useful": rep / u if u else None}
    if k == "gather16":
        res[k]["reported_over_64B_sectors"] = rep / useful["gather16_sectors64"]
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE over tools/calib_fetch.hip (1 GiB buffer, one pass per launch, median of 3 launches)", "kernels": res},
          open(dst, "w"), indent=1)
print(json.dumps(res, indent=1))
PY
