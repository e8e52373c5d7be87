#!/bin/bash
# dev: variants of conv_mfma_cs.hip built with ablation flags -> warpconvnet_amd/csrc/libwcn_abl_<name>.so
cd "$(dirname "$0")/../warpconvnet_amd/csrc" || exit 1
make -j8 >/dev/null 2>&1
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $flags -c conv_mfma_cs.hip -o _build/cs_$name.o || exit 1
  objs=$(ls _build/*.o | grep -v "/cs_" | grep -v "conv_mfma_cs.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs _build/cs_$name.o -o libwcn_abl_$name.so || exit 1
  echo built libwcn_abl_$name.so
done
