# L2 warm-up touches in the 256 x 256 ping-pong tile (FAV_CONV_TOUCH = K steps ahead, 0 = off): experiments build
export FAV_LIB_PATH=failure_aware_vision_amd/lib/variants/exp.so
OUT=${1:-gpurun_out/r4_touch.txt}; : > $OUT
for r in 1 2; do for t in 0 1 2 3 4; do
  echo "FAV_CONV_TOUCH=$t" >> $OUT
  for o in "L3c1" "T3c1"; do FAV_CONV_TOUCH=$t timeout -k 10 120 python tools/conv_bench.py --frames 7680 --only "$o" --iters 5 2>&1 | grep "TF/s" >> $OUT || exit 1; done
done; done
