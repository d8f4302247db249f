set -e
OUT=gpurun_out/r4d_l2nw4.txt
: > $OUT
for nw in 8 4 8 4; do
  echo "## FAV_TAIL_L2_NW=$nw" >> $OUT
  FAV_TAIL_L2_NW=$nw timeout -k 10 300 python tools/tail_bench.py --frames 7680 --iters 5 --only "L2 " >> $OUT 2>&1
done
echo "## phases nw=4" >> $OUT
FAV_TAIL_L2_NW=4 FAV_CONV_DBG=1 timeout -k 10 300 python tools/tail_bench.py --frames 7680 --iters 1 --only "L2 " 2>&1 | grep -E "tail dbg|L2" >> $OUT
echo "## tests nw=4 (default)" >> $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_tail.py tests/test_gpu_ops.py -x -q >> $OUT 2>&1
