# LDS / MFMA counters of one tail_bench shape (run on the GPU box through gpurun).
# usage: bash tools/pmc_tail.sh "<shape filter>" <tag>
set -o pipefail
FILTER="$1"; TAG=${2:-pmc_tail}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d $OUT/p1 -- python3 tools/tail_bench.py --only "$FILTER" --iters 2 > $OUT/p1.log 2>&1; echo "pass rc=$?"
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:90]
        if "tail" not in k and "conv" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
    for k in acc:
        a = {c: acc[k][c] / n[k][c] for c in acc[k]}
        print(k)
        print("   dispatches", max(n[k].values()), "  LDS conflict / active = %.4f" % (a.get("SQ_LDS_BANK_CONFLICT", 0) / max(a.get("SQ_LDS_IDX_ACTIVE", 1), 1)),
              "  MFMA busy / busy cycles(x4 SIMD) = %.3f" % (a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(4 * a.get("SQ_BUSY_CYCLES", 1), 1)), {c: round(v) for c, v in a.items()})
PY
find $OUT -name "*.csv" -size +5M -delete
