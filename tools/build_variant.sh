#!/bin/bash
# Build a measurement variant of the library next to the product (same ABI, -DJB_LAB + other -D flags; the product
# is never built with JB_LAB):
#   bash tools/build_variant.sh <name> [-DJB_STORE_AUX=18 ...]   ->  tools/ab/libjpegblk_<name>.so
# Only jb_kernels.hip is recompiled; the host objects of the product build are reused.
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
C=$R/jpeg_decoder_amd/csrc
mkdir -p $R/tools/ab
make -C $C >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w -DJB_LAB "$@" -c $C/jb_kernels.hip -o /tmp/jb_kernels_$NAME.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|Scratch" | paste - - - |
  sed -E 's/.*Name: ([^ ]*) .*VGPRs: ([0-9]+).*: ([0-9]+) .*/\1 vgpr=\2 scratch=\3/' | grep -E "Li1ELi1ELb0ELb0|Li2ELi2ELb0ELb0|Li1ELi1ELb0ELb1" || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -Wl,-z,defs -o $R/tools/ab/libjpegblk_$NAME.so /tmp/jb_kernels_$NAME.o $C/jb_huff.o $C/jb_api.o $C/jb_geometry.o $C/jb_frontend.o $C/jb_frontend_ext.o $C/jb_batch.o
echo "built tools/ab/libjpegblk_$NAME.so"
