# attention block shape (waves per block; two blocks per CU): experiments build (make EXPERIMENTS=1 OUT=../lib/variants/exp.so)
export FAV_LIB_PATH=failure_aware_vision_amd/lib/variants/exp.so
OUT=${1:-gpurun_out/r4_attn_waves.txt}; : > $OUT
for r in 1 2; do for w in 0 4 5 6 7 8; do
  echo -n "FAV_ATTN_WAVES=$w: " >> $OUT
  FAV_ATTN_WAVES=$w timeout -k 10 120 python tools/vit_bench.py --batch 512 --steps 10 2>&1 | grep vit_b16 >> $OUT || echo failed >> $OUT
done; done
