for cfg in "1024 4096" "8192 8192"; do
  set -- $cfg
  for prof in "" "--no-profile"; do
  echo "== chunk_a $1 chunk_b $2 $prof"
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-frames 0 --chunk-a $1 --chunk-b $2 $prof | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('achieved'))"
  done
done
