for cfg in "32 256" "64 512" "128 512" "256 1024" "512 2048" "1024 4096"; do
  set -- $cfg
  echo "== chunk_a $1 chunk_b $2"
  timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cpu-frames 0 --chunk-a $1 --chunk-b $2 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done
