# Same-box A/B of two library builds (failure_aware_vision_amd/lib/variants/a.so, b.so) by the per-op table of the headline step.
# usage: bash tools/ab_ops.sh <out file> [rounds]
set -e
OUT=${1:-gpurun_out/ab_ops.txt}; ROUNDS=${2:-2}
LIB=failure_aware_vision_amd/lib
cp $LIB/libfav_hip.so $LIB/variants/_keep.so
: > $OUT
for r in $(seq $ROUNDS); do
  for v in a b; do
    cp $LIB/variants/$v.so $LIB/libfav_hip.so
    echo "## $v" >> $OUT
    timeout -k 10 300 python tools/op_table.py --steps 3 2>/dev/null | awk '{print $1, $2, $(NF-3)}' >> $OUT
  done
done
cp $LIB/variants/_keep.so $LIB/libfav_hip.so
