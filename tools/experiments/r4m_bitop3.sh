set -e
OUT=gpurun_out/r4m_bitop3.txt
: > $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_tail.py -x -q >> $OUT 2>&1
timeout -k 10 600 python bench.py --steps 10 --warmup 3 --no-extra --cpu-frames 0 --cpu-port-frames 0 >> $OUT 2>&1
timeout -k 10 300 python tools/op_table.py --steps 3 >> $OUT 2>&1
