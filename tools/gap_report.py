"""Where the GPU idles inside an SDS step: reads a rocprofv3 --kernel-trace CSV, takes the union of busy intervals over all queues and
charges every idle gap to the kernel that ENDS it (the launch the GPU was waiting for).  Steps are delimited by the fused Adam launch.

    python tools/gap_report.py <kernel_trace.csv> [steps_to_average=5] [out.json]"""
import collections
import csv
import json
import re
import sys


def short(n):
    n = re.sub(r"\(.*$", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"at::native::", "", n)
    return n[:70]


def main():
    path = sys.argv[1]
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    adam = [i for i, r in enumerate(rows) if "k_adam" in r[2]]
    # one Adam launch per optimizer group per step may exist: a "step end" is the last Adam launch of a burst
    ends = [i for j, i in enumerate(adam) if j + 1 == len(adam) or rows[adam[j + 1]][0] - rows[i][1] > 2_000_000]
    if len(ends) < nsteps + 1:
        print("only %d step delimiters found" % len(ends)); nsteps = max(1, len(ends) - 1)
    lo, hi = ends[-nsteps - 1] + 1, ends[-1] + 1
    win = rows[lo:hi]
    t0, t1 = win[0][0], max(r[1] for r in win)
    busy_end = win[0][0]
    idle_by = collections.Counter(); idle_n = collections.Counter(); busy = 0
    prev_name = "(step start)"
    pair = collections.Counter()
    for s, e, n in win:
        if s > busy_end:
            g = s - busy_end
            idle_by[short(n)] += g; idle_n[short(n)] += 1
            pair[(prev_name, short(n))] += g
            busy += e - s
            busy_end = e
        else:
            if e > busy_end:
                busy += e - busy_end
                busy_end = e
        if e >= busy_end:
            prev_name = short(n)
    wall = (t1 - t0) / nsteps / 1e6
    print("steps %d  wall %.3f ms/step  busy %.3f  idle %.3f  launches/step %.0f" % (nsteps, wall, busy / nsteps / 1e6, wall - busy / nsteps / 1e6, len(win) / nsteps))
    out = {"steps": nsteps, "wall_ms": wall, "busy_ms": busy / nsteps / 1e6, "launches_per_step": len(win) / nsteps, "idle_before": {}, "pairs": []}
    print("idle charged to the kernel that ends the gap (ms/step, gaps/step, avg us):")
    for n, g in idle_by.most_common(30):
        print("  %-72s %.3f  %5.1f  %6.1f" % (n, g / nsteps / 1e6, idle_n[n] / nsteps, g / idle_n[n] / 1e3))
        out["idle_before"][n] = {"ms_per_step": g / nsteps / 1e6, "gaps_per_step": idle_n[n] / nsteps}
    print("largest (previous -> next) idle pairs:")
    for (a, b), g in pair.most_common(25):
        print("  %-50s -> %-50s %.3f" % (a[:50], b[:50], g / nsteps / 1e6))
        out["pairs"].append([a, b, g / nsteps / 1e6])
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1)


main()
