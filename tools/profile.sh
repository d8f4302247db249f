# rocprofv3 profiles of the headline bench (run on the GPU box through gpurun).
# usage: bash tools/profile.sh <tag> [config]      config: mc30 (default: the headline) | single | ens5 | vit
set -o pipefail
TAG=${1:-r1}
CFG=${2:-mc30}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
BENCH="python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --cpu-port-frames 0 --no-extra --no-profile"
if [ "$CFG" != "mc30" ]; then BENCH="python3 bench.py --config $CFG --steps 5 --warmup 2 --no-profile"; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
echo "trace rc=$?"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1
echo "pmc_sq rc=$?"
if [ "$CFG" = "mc30" ]; then
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
echo "pmc_fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
echo "pmc_write rc=$?"
fi
find $OUT -name "*.csv" | head -20
du -sh $OUT
python3 tools/profile_summary.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
# keep only the summaries small enough to merge back
find $OUT -name "*counter_collection.csv" -size +20M -delete
find $OUT -name "*kernel_trace.csv" -size +20M -delete
