set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/gpu_tests_3.log 2>&1; tail -n 6 gpurun_out/gpu_tests_3.log
for cfg in "52 416" "104 416" "128 1024" "256 1024" "512 2048"; do
  set -- $cfg
  echo "== chunk_a $1 chunk_b $2"
  timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cpu-frames 0 --chunk-a $1 --chunk-b $2 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done
