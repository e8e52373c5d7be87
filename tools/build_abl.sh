#!/bin/bash
# dev: variants of ONE source file built with extra flags -> warpconvnet_amd/csrc/libwcn_abl_<name>.so
# usage: tools/build_abl.sh <file.hip> name:"-DFLAG ..." [name2:flags ...]     (load with WARPCONVNET_AMD_LIB=...)
# The gather-GEMM ablations of DESIGN.md section 4.2a (-DCS_ABL_LOCAL / -DCS_ABL_NOSTORE / -DCS_ABL_PLAINSTORE) are not in the
# shipped kernel: `git apply tools/cs_ablations.patch` first, `git apply -R` afterwards.
cd "$(dirname "$0")/../warpconvnet_amd/csrc" || exit 1
src=$1; shift
stem=${src%.hip}
make -j8 >/dev/null 2>&1
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $flags -c $src -o _build/abl_$name.o || exit 1
  objs=$(ls _build/*.o | grep -v "/abl_" | grep -v "/$stem.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs _build/abl_$name.o -o libwcn_abl_$name.so || exit 1
  echo built libwcn_abl_$name.so
done
