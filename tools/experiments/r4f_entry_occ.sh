set -e
OUT=gpurun_out/r4f_entry_occ.txt
: > $OUT
for occ in 3 2 3 2; do
  echo "## FAV_ENTRY_OCC=$occ" >> $OUT
  FAV_ENTRY_OCC=$occ timeout -k 10 300 python tools/op_table.py --steps 3 2>&1 | grep -E "wall|drop\+red|^ *5 tail|^ *6 tail" >> $OUT
done
echo "## vit bench (attention NKT=13)" >> $OUT
timeout -k 10 300 python tools/vit_bench.py >> $OUT 2>&1 || true
timeout -k 10 900 python -m pytest tests/test_gpu_vit.py tests/test_gpu_golden.py -x -q -k "vit" >> $OUT 2>&1
