"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected separately: the TCC block has 4
slots, FETCH_SIZE takes 3 and WRITE_SIZE 2 -- MI355X_MICROARCH.md, rocprofv3 PMC slots).

    cd /tmp && export TMPDIR=/tmp && cd <repo>
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o sds -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --eager
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o sds -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --eager
    python tools/pmc_traffic.py gpurun_out/pmc_fetch/sds_counter_collection.csv gpurun_out/pmc_write/sds_counter_collection.csv profiles/r01_pmc_traffic.json

Units / corrections (same guide, HBM section): counters are in KiB (x1024 -> bytes); on gfx950 FETCH_SIZE tallies the 128-byte
requests of coalesced reads at 64 bytes, i.e. reports HALF the bytes of such streams -> the read side is doubled; WRITE_SIZE is used
as reported (uncalibrated).  Infinity-Cache hits are counted, not excluded.

Round 5 -- ONE number per kernel.  tools/calib_fetch.hip / profiles/r05_fetch_calibration.json measure FETCH_SIZE on known byte counts
(1 GiB buffer, beyond the Infinity Cache): coalesced streams at 16 AND at 4 bytes per lane report 0.500 of their bytes; a random 16-byte
gather reports 0.996 x (64 bytes per lane), i.e. its 64-byte requests at FACE VALUE; random 48-byte rows 1.66 x their useful bytes (1.5
sectors per row, partly shared).  So the factor is 2 for kernels whose reads are coalesced streams and 1 for kernels whose reads are
gathers; GATHER_KERNELS lists the latter (the rasterizer's record / row gathers and binary-search probes) and every kernel's entry says
which factor it got."""
import collections, csv, hashlib, json, os, re, sys


def sources_sha():
    """Same hash bench.py computes at run time (bench.sources_sha): the profile is stamped with the kernel sources it was taken on, and
    bench.py marks `traffic_stale` when the sources it runs differ."""
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dreamwaltz-g_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


# kernels whose HBM reads are dominated by gathers (<= 16 bytes per lane at unrelated addresses): FETCH_SIZE at face value
GATHER_KERNELS = ("k_render_fwd", "k_render_bwd", "k_gather_partials", "k_rank_merge", "k_grid_fwd", "k_grid_bwd")


def short(name):
    m = re.search(r"(k_[A-Za-z0-9_]+(?:<[^>]*>)?)", name)
    s = m.group(1) if m else name[:60]
    return s.replace(" ", "")


def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1; a[1] += float(r["Counter_Value"])
    return agg


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        nf, f = fetch.get(k, [0, 0.0]); nw, w = write.get(k, [0, 0.0])
        n = max(nf, nw)
        if n == 0:
            continue
        fb, wb = f * 1024 / max(nf, 1), w * 1024 / max(nw, 1)
        factor = 1 if k.split("<")[0] in GATHER_KERNELS else 2
        out[k] = {"launches": n, "fetch_bytes_per_launch_raw": fb, "fetch_factor": factor, "fetch_bytes_per_launch_corrected": factor * fb,
                  "write_bytes_per_launch": wb, "hbm_bytes_per_launch": factor * fb + wb}
    json.dump({"sources_sha": sources_sha(), "commit": sys.argv[4] if len(sys.argv) > 4 else None, "plan_dtype": sys.argv[5] if len(sys.argv) > 5 else "f32x", "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over python bench.py --steps 2 --warmup 1 --eager",
               "corrections": "KiB->bytes x1024; FETCH_SIZE x 2 for kernels that read coalesced streams, x 1 for gather kernels (fetch_factor per kernel; "
                              "calibrated on known byte counts: profiles/r05_fetch_calibration.json, tools/calib_fetch.hip); WRITE_SIZE as reported",
               "kernels": out}, open(sys.argv[3], "w"), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:14]:
        print("%-34s %5d launches  fetch(corr) %9.2f MB  write %9.2f MB" % (k, v["launches"], v["fetch_bytes_per_launch_corrected"] / 1e6, v["write_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
