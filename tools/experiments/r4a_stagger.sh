set -e
mkdir -p gpurun_out
OUT=gpurun_out/r4a_stagger.txt
: > $OUT
rm -f gpurun_out/r4a_dump.bin
echo "## phase dump, no stagger" >> $OUT
FAV_CONV_DBG=1 FAV_CONV_DBG_DUMP=gpurun_out/r4a_dump.bin timeout -k 10 300 python tools/tail_bench.py --frames 7680 --iters 1 --only "L3 3x3" >> $OUT 2>&1
FAV_CONV_DBG=1 FAV_CONV_DBG_DUMP=gpurun_out/r4a_dump.bin timeout -k 10 300 python tools/tail_bench.py --frames 7680 --iters 1 --only "L2 tail 3x3" >> $OUT 2>&1
python tools/phase_overlap.py gpurun_out/r4a_dump.bin >> $OUT 2>&1
for us in 0 60 120 0 60 120 30 240; do
  echo "## FAV_TAIL_STAGGER_US=$us" >> $OUT
  FAV_TAIL_STAGGER_US=$us timeout -k 10 300 python tools/tail_bench.py --frames 7680 --iters 5 --only "L3 3x3" >> $OUT 2>&1
done
for us in 0 25 50 0 25 50 100; do
  echo "## FAV_TAIL_STAGGER_US=$us" >> $OUT
  FAV_TAIL_STAGGER_US=$us timeout -k 10 300 python tools/tail_bench.py --frames 7680 --iters 5 --only "L2 " >> $OUT 2>&1
done
rm -f gpurun_out/r4a_dump2.bin
echo "## phase dump, stagger 120 (L3)" >> $OUT
FAV_TAIL_STAGGER_US=120 FAV_CONV_DBG=1 FAV_CONV_DBG_DUMP=gpurun_out/r4a_dump2.bin timeout -k 10 300 python tools/tail_bench.py --frames 7680 --iters 1 --only "L3 3x3" >> $OUT 2>&1
python tools/phase_overlap.py gpurun_out/r4a_dump2.bin >> $OUT 2>&1
rm -f gpurun_out/r4a_dump.bin gpurun_out/r4a_dump2.bin
