set -e
OUT=gpurun_out/r4b_x128.txt
: > $OUT
for x in 0 1 0 1; do
  echo "## FAV_CONV_X128=$x" >> $OUT
  FAV_CONV_X128=$x timeout -k 10 300 python tools/conv_bench.py --frames 7680 --iters 5 --only "L3c2" >> $OUT 2>&1
done
echo "## hash check (bit-identity of the two tiles)" >> $OUT
FAV_CONV_X128=0 timeout -k 10 300 python tools/conv_hash.py >> $OUT 2>&1 || true
FAV_CONV_X128=1 timeout -k 10 300 python tools/conv_hash.py >> $OUT 2>&1 || true
