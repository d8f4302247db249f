# Same-box A/B of library builds on the ViT path: failure_aware_vision_amd/lib/variants/*.so through FAV_LIB_PATH, alternating.
# usage: bash tools/ab_vit.sh <out file> [rounds] [batches...]
OUT=${1:-gpurun_out/ab_vit.txt}; ROUNDS=${2:-2}; shift 2 || true
BATCHES=${@:-"64 512"}
: > $OUT
for r in $(seq $ROUNDS); do
  for f in failure_aware_vision_amd/lib/variants/*.so; do
    for b in $BATCHES; do
      echo -n "$(basename $f .so) " >> $OUT
      FAV_LIB_PATH=$f timeout -k 10 200 python tools/vit_bench.py --batch $b --steps 20 2>&1 | grep vit_b16 >> $OUT || echo failed >> $OUT
    done
  done
done
