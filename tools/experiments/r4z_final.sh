# round-4 final measurement set (normal build): rocprofv3 trace + PMC passes for the headline, traces for the other configs, op table,
# bench lines.  Steps are chained: a failed or timed-out GPU step ends the script.
set -e
timeout -k 10 900 bash tools/profile.sh r4z > gpurun_out/r4z_profile.log 2>&1
for c in single ens5 vit; do timeout -k 10 400 bash tools/profile.sh r4z_$c $c > gpurun_out/r4z_profile_$c.log 2>&1; done
timeout -k 10 400 bash tools/pmc_valu.sh r4z_valu > gpurun_out/r4z_valu.log 2>&1
timeout -k 10 300 python tools/op_table.py --steps 3 > gpurun_out/r4z_op_table.txt 2>&1
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4z_bench.json 2> gpurun_out/r4z_bench.err
for c in single ens5 vit; do timeout -k 10 400 python bench.py --config $c --steps 10 --warmup 3 > gpurun_out/r4z_bench_$c.json 2>> gpurun_out/r4z_bench.err; done
tail -c 300 gpurun_out/r4z_bench.json
