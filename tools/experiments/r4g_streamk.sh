set -e
OUT=gpurun_out/r4g_streamk.txt
: > $OUT
timeout -k 10 400 python -m pytest tests/test_gpu_streamk.py -x -q >> $OUT 2>&1
timeout -k 10 200 python tools/streamk_bench.py --frames 64 >> $OUT 2>&1
timeout -k 10 200 python tools/streamk_bench.py --frames 512 --iters 5 >> $OUT 2>&1
for sk in 0 1 0 1; do
  echo "## FAV_STREAMK=$sk" >> $OUT
  FAV_STREAMK=$sk timeout -k 10 200 python tools/vit_bench.py >> $OUT 2>&1
done
timeout -k 10 900 python -m pytest tests/test_gpu_vit.py tests/test_gpu_golden.py -x -q -k "vit" >> $OUT 2>&1
