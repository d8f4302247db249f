mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/gpu_tests_6.log 2>&1; tail -n 6 gpurun_out/gpu_tests_6.log
for np in 1 2 4 8 16; do
  echo "== FAV_PIPE=$np"
  FAV_PIPE=$np timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-frames 0 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('achieved'))"
done
