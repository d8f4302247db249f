# Same-box A/B of several builds of the library: failure_aware_vision_amd/lib/variants/<name>.so (built by hand), swapped into place
# between alternating runs of the headline bench.  usage: bash tools/ab.sh <out file> [rounds] [extra bench args]
set -e
OUT=${1:-gpurun_out/ab.txt}; ROUNDS=${2:-2}; shift 2 || true
LIB=failure_aware_vision_amd/lib
cp $LIB/libfav_hip.so $LIB/_keep.so
: > $OUT
for r in $(seq $ROUNDS); do
  for f in $LIB/variants/*.so; do
    v=$(basename $f .so)
    cp $f $LIB/libfav_hip.so
    echo -n "$v " >> $OUT
    timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-extra --cpu-frames 0 --cpu-port-frames 0 --no-profile "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3))" >> $OUT || echo failed >> $OUT
  done
done
cp $LIB/_keep.so $LIB/libfav_hip.so; rm -f $LIB/_keep.so
