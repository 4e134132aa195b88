"""Per-kernel statistics of the STEADY STATE of a traced run, from rocprofv3's kernel trace (round 5; verdict round 4, weak point 11).

rocprofv3 --stats summarises the whole process: for bench.py that includes the one-time set-up -- ~380 weight tensors packed into the f32x
format (torch clamp / half / copy launches), two hipBLASLt and a few rocPRIM launches with Calls = 1 -- which inflated "torch / rocclr" to 8 %
of the kernel time of the round-4 summary.  This tool keeps the rows of <run>_kernel_trace.csv from the FIRST launch of `anchor` on (default
k_preprocess: the rasterizer's first kernel = the first training step; everything before it is model construction) and writes a summary
with rocprofv3's own columns.

    python tools/steady_stats.py <dir with *_kernel_trace.csv> <out.csv> [anchor substring]"""
import collections
import csv
import glob
import math
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    anchor = sys.argv[3] if len(sys.argv) > 3 else "k_preprocess"
    f = sorted(glob.glob(src + "/**/*kernel_trace.csv", recursive=True))[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = next((int(r["Start_Timestamp"]) for r in rows if anchor in r["Kernel_Name"]), None)
    if t0 is None:
        raise SystemExit("steady_stats: no launch of %r in %s" % (anchor, f))
    agg = collections.defaultdict(list)
    dropped = 0
    for r in rows:
        if int(r["Start_Timestamp"]) < t0:
            dropped += 1
            continue
        agg[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = sum(sum(v) for v in agg.values())
    with open(dst, "w", newline="") as out:
        w = csv.writer(out, quoting=csv.QUOTE_ALL)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            n, s = len(v), sum(v)
            mean = s / n
            sd = math.sqrt(sum((x - mean) ** 2 for x in v) / (n - 1)) if n > 1 else 0.0
            w.writerow([k, n, s, "%.6f" % mean, "%.6f" % (100.0 * s / total), min(v), max(v), "%.6f" % sd])
    print("steady_stats: %d launches kept, %d set-up launches before the first %s dropped -> %s" % (sum(len(v) for v in agg.values()), dropped, anchor, dst))


if __name__ == "__main__":
    main()
