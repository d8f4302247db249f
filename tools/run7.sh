mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/gpu_tests_7.log 2>&1; tail -n 6 gpurun_out/gpu_tests_7.log
timeout -k 10 300 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_r1_c.json 2>gpurun_out/bench_r1_c.err; cat gpurun_out/bench_r1_c.json
