set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/gpu_tests_2.log 2>&1; tail -n 15 gpurun_out/gpu_tests_2.log
for cfg in "13 104" "26 104" "52 104" "104 104" "26 208" "26 416" "52 416"; do
  set -- $cfg
  echo "== chunk_a $1 chunk_b $2"
  timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cpu-frames 0 --chunk-a $1 --chunk-b $2 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done
