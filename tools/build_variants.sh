#!/bin/bash
# Build A/B variants of libspx_nnue.so into variants/ (git-ignored; travels to the GPU box).
#   usage: tools/build_variants.sh name1:"-DFOO=1 -DBAR=2" name2:"..."
set -e
cd "$(dirname "$0")/.."
mkdir -p variants
SRC=stormphrax_amd/csrc
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags \
      $SRC/spx_kernels.hip $SRC/spx_ftx.hip $SRC/spx_movegen.hip $SRC/spx_api.cpp $SRC/spx_chess.cpp $SRC/spx_luts.cpp $SRC/spx_synth.cpp $SRC/spx_selfplay.cpp $SRC/spx_group.cpp \
      -o variants/libspx_$name.so -ldl &
done
wait
ls -la variants/
