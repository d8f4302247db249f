set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/gpu_tests_5.log 2>&1; tail -n 6 gpurun_out/gpu_tests_5.log
timeout -k 10 300 python tools/op_table.py > gpurun_out/op_table_bk32.txt 2>&1; cat gpurun_out/op_table_bk32.txt
FAV_CONV_BK=64 timeout -k 10 300 python tools/op_table.py > gpurun_out/op_table_bk64.txt 2>&1; head -3 gpurun_out/op_table_bk64.txt
