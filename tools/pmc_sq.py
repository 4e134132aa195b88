"""Where the waves' time goes, per kernel, from ONE rocprofv3 PMC pass of SQ counters (8 slots on gfx950; MI355X_MICROARCH.md "rocprofv3 PMC
slots"): SQ_WAVE_CYCLES = SQ_WAIT_ANY (parked: s_waitcnt / barrier) + SQ_WAIT_INST_ANY (issue stall) + SQ_ACTIVE_INST_ANY (issuing), all in
quad-cycles; plus the instruction mix.  Used to state what actually bounds the rasterizer / small-M GEMM kernels.

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS \\
              --kernel-trace --output-format csv -d gpurun_out/pmc_sq -o sds -- python bench.py --headline-only --no-cpu-baseline --eager --steps 2 --warmup 1
    python tools/pmc_sq.py gpurun_out/pmc_sq/.../sds_counter_collection.csv profiles/r03_pmc_sq.json [commit]"""
import collections, csv, json, re, sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from pmc_traffic import sources_sha, short  # noqa: E402

NAMES = ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS")


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(sys.argv[1])):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r.get("Dispatch_Id"))
        if key not in seen:
            seen.add(key); launches[k] += 1
    out = {}
    for k, c in agg.items():
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if wc <= 0:
            continue
        out[k] = {"launches": launches[k], "wave_quad_cycles_per_launch": wc / launches[k],
                  "parked_frac": c.get("SQ_WAIT_ANY", 0.0) / wc, "issue_stall_frac": c.get("SQ_WAIT_INST_ANY", 0.0) / wc,
                  "issuing_frac": c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, "issuing_valu_frac": c.get("SQ_ACTIVE_INST_VALU", 0.0) / wc,
                  "valu_insts_per_launch": c.get("SQ_INSTS_VALU", 0.0) / launches[k], "salu_insts_per_launch": c.get("SQ_INSTS_SALU", 0.0) / launches[k],
                  "lds_insts_per_launch": c.get("SQ_INSTS_LDS", 0.0) / launches[k]}
    json.dump({"sources_sha": sources_sha(), "commit": sys.argv[3] if len(sys.argv) > 3 else None,
               "source": "rocprofv3 --pmc " + " ".join(NAMES) + " over python bench.py --headline-only --eager --steps 2 --warmup 1",
               "units": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves; the three fractions are disjoint and sum to ~1",
               "kernels": out}, open(sys.argv[2], "w"), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["wave_quad_cycles_per_launch"] * kv[1]["launches"])[:16]:
        print("%-34s %5d launches  parked %.2f  issue-stall %.2f  issuing %.2f (VALU %.2f)" % (k, v["launches"], v["parked_frac"], v["issue_stall_frac"],
                                                                                                   v["issuing_frac"], v["issuing_valu_frac"]))


if __name__ == "__main__":
    main()
