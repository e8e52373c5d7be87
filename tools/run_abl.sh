#!/bin/bash
# dev: phase profile of the ablation builds (tools/build_abl.sh)
for name in "$@"; do
  echo "=== $name"
  WARPCONVNET_AMD_LIB=$PWD/warpconvnet_amd/csrc/libwcn_$name.so python tools/prof_phases.py 2>&1 | grep -v amdgpu.ids | tail -20
done
