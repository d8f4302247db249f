# Vector-ALU / LDS issue counters of the headline bench, per kernel (run on the GPU box through gpurun).
# usage: bash tools/pmc_valu.sh <tag>
set -o pipefail
TAG=${1:-pmc_valu}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
BENCH="python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --cpu-port-frames 0 --no-extra --no-profile"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU --output-format csv -d $OUT/pmc_valu -- $BENCH > $OUT/pmc_valu.log 2>&1
echo "pmc_valu rc=$?"
python3 tools/profile_summary.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
find $OUT -name "*counter_collection.csv" -size +20M -delete
