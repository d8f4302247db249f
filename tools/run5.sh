mkdir -p gpurun_out
for ns in 2 3 4; do for bk in 32 64; do
echo "=== NS=$ns BK=$bk"
FAV_CONV_NS=$ns FAV_CONV_BK=$bk timeout -k 10 120 python tools/conv_bench.py 2>&1 | grep -v "^X" | awk '{print $1,$2,$3,$4, $6, $8, $10, $14, $15}'
done; done
