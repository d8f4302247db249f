set -e
OUT=gpurun_out/r4e_chunk_clocks.txt
: > $OUT
for nw in 8 4; do
echo "## FAV_TAIL_L2_NW=$nw" >> $OUT
FAV_TAIL_L2_NW=$nw FAV_CONV_DBG=1 timeout -k 10 300 python tools/tail_bench.py --frames 7680 --iters 1 2>&1 | grep -E "tail dbg" >> $OUT
done
