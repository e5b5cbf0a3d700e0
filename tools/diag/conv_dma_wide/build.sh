#!/bin/bash
# Builds scratch/libymk_wide.so (the product library with the experiment's LDS-DMA kernel file: 256 x 256 "wide" form behind
# ymk_debug_option("conv_split_tile", 22)) and scratch/libymk_ablate_<n>.so (the same with -DYMK_ABLATE=n) for
# tools/jobs/r05_l.sh / r05_m.sh.  Nothing here touches yomitoku_amd/: the product sources are copied, patched and compiled
# under scratch/.  Usage: bash tools/diag/conv_dma_wide/build.sh [n ...]
set -e
ROOT=$(cd "$(dirname "$0")/../../.." && pwd)
CSRC=$ROOT/yomitoku_amd/csrc
W=$ROOT/scratch/wide_build
mkdir -p $W
(cd $CSRC && make >/dev/null)
cp $CSRC/ymk_conv_split.hip $W/ymk_conv_split.hip
(cd $W && patch -s -p3 ymk_conv_split.hip < $ROOT/tools/diag/conv_dma_wide/route_tile22.patch)
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result -I$CSRC -I$ROOT/include"
hipcc $FLAGS -c $W/ymk_conv_split.hip -o $W/split.o
OBJS=$(ls $ROOT/yomitoku_amd/lib/obj/*.o | grep -v "ymk_conv_dma\|ymk_conv_split")
for n in 0 "$@"; do
  hipcc $FLAGS -DYMK_ABLATE=$n -c $ROOT/tools/diag/conv_dma_wide/ymk_conv_dma_wide_experiment.hip -o $W/dma_$n.o
  out=$ROOT/scratch/libymk_ablate_$n.so
  [ "$n" = 0 ] && out=$ROOT/scratch/libymk_wide.so
  hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-z,defs -o $out $OBJS $W/split.o $W/dma_$n.o
  echo "built $out"
done
