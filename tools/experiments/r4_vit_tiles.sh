# ViT at the per-GPU share (64 frames) and above: tile choice x streams, experiments build (make EXPERIMENTS=1 OUT=../lib/variants/exp.so)
# 256x256 tiles per GEMM at 64 frames / one stream: qkv 441, fc1 588, proj / fc2 147 (half of each on two streams)
export FAV_LIB_PATH=failure_aware_vision_amd/lib/variants/exp.so
export FAV_CONV_BIG=1 FAV_CONV_BIG_MINM=2048
OUT=${1:-gpurun_out/r4_vit_tiles.txt}; : > $OUT
for b in 64 128; do
  for st in 1 2; do
    for big in 100000 500 400 200 100 50; do
      echo -n "batch $b streams $st FAV_VIT_BIG_TILES=$big: " >> $OUT
      FAV_VIT_BIG_TILES=$big timeout -k 10 120 python tools/vit_bench.py --batch $b --steps 30 --streams $st 2>&1 | grep vit_b16 >> $OUT || echo failed >> $OUT
    done
  done
done
