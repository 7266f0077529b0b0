#!/bin/bash
# Experiment variant of the device entropy decoder next to the product (same ABI, other -D flags):
#   bash tools/build_huff_variant.sh <name> [-DJBH_NO_STORE ...]  ->  tools/ab/libjpegblk_h_<name>.so
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
C=$R/jpeg_decoder_amd/csrc
mkdir -p $R/tools/ab
make -C $C >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w "$@" -c $C/jb_huff.hip -o /tmp/jb_huff_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $R/tools/ab/libjpegblk_h_$NAME.so $C/jb_kernels.o /tmp/jb_huff_$NAME.o $C/jb_api.o $C/jb_geometry.o $C/jb_frontend.o $C/jb_frontend_ext.o $C/jb_batch.o
echo "built tools/ab/libjpegblk_h_$NAME.so"
