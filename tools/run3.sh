set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/gpu_tests_4.log 2>&1; tail -n 6 gpurun_out/gpu_tests_4.log
timeout -k 10 300 python tools/op_table.py --out gpurun_out/op_table_mc3.json > gpurun_out/op_table_mc3.txt 2>&1; cat gpurun_out/op_table_mc3.txt
