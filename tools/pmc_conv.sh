# PMC passes over one conv_bench shape (run on the GPU box through gpurun).
# usage: bash tools/pmc_conv.sh "<shape filter>" <tag> [ENV=VAL ...]
set -o pipefail
FILTER="$1"; TAG=${2:-pmc}; shift 2
for kv in "$@"; do export "$kv"; done
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
i=0
while read -r CNT; do
  i=$((i+1))
  rocprofv3 --pmc $CNT --output-format csv -d $OUT/p$i -- python3 tools/conv_bench.py --only "$FILTER" --iters 3 > $OUT/p$i.log 2>&1; echo "pass $i rc=$?";  true || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
done <<'CNTS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_TAG_STALL_sum
GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
CNTS
python3 - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if "conv_igemm" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in acc: print(f"{k:40s} {acc[k]/n[k]:18.1f}  (per dispatch, {n[k]} dispatches)")
PY
find $OUT -name "*.csv" -size +5M -delete
