for v in a_base b_lstore; do
  echo "== $v" >> gpurun_out/r4_ab_convbench.txt
  FAV_LIB_PATH=failure_aware_vision_amd/lib/variants/$v.so timeout -k 10 200 python tools/conv_bench.py --frames 512 --only "V " --iters 5 2>&1 | grep "TF/s" >> gpurun_out/r4_ab_convbench.txt
  for o in "L3c1" "L3c2" "L4c1" "L4c2"; do
    FAV_LIB_PATH=failure_aware_vision_amd/lib/variants/$v.so timeout -k 10 200 python tools/conv_bench.py --frames 7680 --only "$o" --iters 3 2>&1 | grep "TF/s" >> gpurun_out/r4_ab_convbench.txt
  done
done
