#!/bin/bash
# dev: MinkUNet-14 iteration time for several builds of the library on ONE box (host speed varies box to box):
# tools/ab_unet.sh "<voxel counts>" <rounds> lib1.so lib2.so ...   ("-" = the shipped build)
SIZES=$1; ROUNDS=$2; shift 2
for v in $SIZES; do for r in $(seq $ROUNDS); do for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset WARPCONVNET_AMD_LIB; else export WARPCONVNET_AMD_LIB=$PWD/warpconvnet_amd/csrc/$lib; fi
  echo -n "$lib round $r: "; python tools/bench_minkunet.py --voxels $v --iters 20 2>/dev/null | tail -1
done; done; done
