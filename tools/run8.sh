mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "not ten_thousand" > gpurun_out/gpu_tests_8.log 2>&1; tail -n 30 gpurun_out/gpu_tests_8.log
