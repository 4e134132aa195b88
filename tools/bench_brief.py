"""One or more bench.py configurations, each summarised on two lines (value, rasterizer roofline entries, top kernel times):
    python tools/bench_brief.py c5 c1 c2 c3 [-- extra bench.py flags]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    extra = args[args.index("--") + 1:]; args = args[:args.index("--")]
for c in args or ["c3"]:
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", c, "--headline-only", "--no-cpu-baseline"] + extra,
                       capture_output=True, text=True, timeout=900)
    lines = [x for x in r.stdout.splitlines() if x.startswith('{"metric"')]
    if not lines:
        print(c, "FAILED", (r.stdout + r.stderr)[-1500:]); continue
    d = json.loads(lines[-1]); rf = d.get("roofline", {})
    print(c, round(d["value"], 2), d["unit"][:28], "ms/step", round(d["ms_per_step"], 3),
          {k: (round(v["ms"], 3), round(v["frac_of_hbm_peak"], 4)) for k, v in rf.items() if k.startswith("raster")}, flush=True)
    k = d.get("kernel_ms_per_step", {})
    print("   ", {a: round(b, 3) for a, b in sorted(k.items(), key=lambda x: -x[1]) if a.startswith("raster")}, flush=True)
