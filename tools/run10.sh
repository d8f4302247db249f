mkdir -p gpurun_out
python bench.py > gpurun_out/bench_r1_final.json 2> gpurun_out/bench_r1_final.err; cat gpurun_out/bench_r1_final.json
for pol in none last_layer "layer4+fc"; do
python bench.py --policy $pol --cpu-frames 0 --steps 5 --warmup 2 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$pol', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline_mfma']['achieved'])"
done
