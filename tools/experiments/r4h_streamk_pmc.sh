set -e
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
OUT=gpurun_out/r4h
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -- python3 tools/streamk_bench.py --frames 512 --iters 2 > $OUT/tcc.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/sq -- python3 tools/streamk_bench.py --frames 512 --iters 2 > $OUT/sq.log 2>&1
python3 - <<'PY' > gpurun_out/r4h_streamk_pmc.txt
import csv, glob, collections
for sub in ("tcc", "sq"):
    f = glob.glob(f"gpurun_out/r4h/{sub}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:48] + " grid " + row.get("Grid_Size", "?")
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in agg.items():
        print(k)
        for c, v in d.items():
            print("     %-28s per-dispatch avg %.4g (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
rm -rf $OUT
