set -e
OUT=gpurun_out/r4c_l3nw4.txt
: > $OUT
for nw in 8 4 8 4; do
  echo "## FAV_TAIL_L3_NW=$nw" >> $OUT
  FAV_TAIL_L3_NW=$nw timeout -k 10 300 python tools/tail_bench.py --frames 7680 --iters 5 --only "L3 3x3" >> $OUT 2>&1
done
echo "## phases nw=4" >> $OUT
rm -f gpurun_out/r4c_dump.bin
FAV_TAIL_L3_NW=4 FAV_CONV_DBG=1 FAV_CONV_DBG_DUMP=gpurun_out/r4c_dump.bin timeout -k 10 300 python tools/tail_bench.py --frames 7680 --iters 1 --only "L3 3x3" 2>&1 | grep -E "tail dbg|L3" >> $OUT
python tools/phase_overlap.py gpurun_out/r4c_dump.bin >> $OUT 2>&1
rm -f gpurun_out/r4c_dump.bin
echo "## tests nw=4" >> $OUT
FAV_TAIL_L3_NW=4 FAV_TAIL_MIN_ROWS=0 timeout -k 10 600 python -m pytest tests/test_gpu_tail.py -x -q >> $OUT 2>&1
