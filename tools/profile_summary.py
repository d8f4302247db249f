"""Summarise rocprofv3 output directories produced by tools/profile.sh."""
import csv, glob, os, sys, collections
out = sys.argv[1]

def find(sub, pat):
    r = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return r[0] if r else None

f = find("trace", "*kernel_stats.csv")
if f:
    print("== kernel stats (rocprofv3 --kernel-trace --stats):", f)
    for row in csv.DictReader(open(f)):
        print("  %-90s calls %7s total_ms %10.3f avg_us %9.2f pct %6s" % (
            row.get("Name", "")[:90], row.get("Calls"), float(row.get("TotalDurationNs", 0)) / 1e6,
            float(row.get("AverageNs", 0)) / 1e3, row.get("Percentage")))
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    f = find(sub, "*counter_collection.csv")
    if not f:
        print("no counter csv for", sub); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:60]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        n[(k, row["Counter_Name"])] += 1
    print("== counters", sub)
    for k, d in agg.items():
        print("  ", k)
        for c, v in d.items():
            print("      %-28s sum %.4g  per-dispatch %.4g  (n=%d)" % (c, v, v / n[(k, c)], n[(k, c)]))

# HBM traffic of the conv kernel per launch -> profiles/pmc_traffic.json (read by bench.py)
import json
def conv_sum(sub, counter):
    f = find(sub, "*counter_collection.csv")
    tot, n = 0.0, 0
    if f:
        for row in csv.DictReader(open(f)):
            if any(k in row["Kernel_Name"] for k in ("conv_igemm_kernel", "conv3x3_halo_kernel", "bottleneck_tail_kernel", "entry_reduce_kernel", "stem7_pool_kernel")) and row["Counter_Name"] == counter:
                tot += float(row["Counter_Value"]); n += 1
    return tot, n
fs, fn = conv_sum("pmc_fetch", "FETCH_SIZE")
ws, wn = conv_sum("pmc_write", "WRITE_SIZE")
if fn and wn:
    # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 FETCH_SIZE counts 64 B per 128-B request -> x2
    fetch = fs * 1024.0 * 2.0 / fn
    write = ws * 1024.0 / wn
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    d = {"kernel_source_sha256": bench.kernel_source_sha(),   # the sources these counters were measured on
         "conv_fetch_bytes_per_launch": fetch, "conv_write_bytes_per_launch": write,
         "conv_bytes_per_launch": fetch + write, "launches_profiled": fn,
         "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --steps 2 --warmup 1 --cpu-frames 0 --cpu-port-frames 0 --no-extra --no-profile`, "
                   "FETCH_SIZE x2 (gfx950 correction), " + os.path.basename(out.rstrip('/'))}
    json.dump(d, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
    print("pmc_traffic", d)
