#!/bin/bash
# What costs the small-image regime its 22 % per tile?  Same kernel, 512 images per launch, sizes
# chosen to separate the effects (4:2:0, tile = 32 MCUs = 512 px):
#   1024x448  64 MCUs per row: row-bound tiling, no wrap, aligned rows          (clean small image)
#   640x448   40 MCUs per row: linear tiling, segments wrap, rows 1920 B (64-B aligned)
#   672x448   42 MCUs per row: wrap, rows 2016 B (32-B aligned)
#   680x451   wrap, rows 2040 B (8-B aligned), no 3-pixel tail, ragged bottom
#   679x451   wrap, rows 2037 B (odd), 3-pixel tail on every row          (the reference's img.jpg)
# and 679x451 with the row-bound tiling / byte stores forced.
set -u
L=jpeg_decoder_amd/libjpegblk.so
export JB_BENCH_EVENT_EVERY=4
bash tools/ab_bench.sh "$L" "1024x448-420:512 640x448-420:512 672x448-420:512 680x451-420:512 679x451-420:512" 1
echo "--- 679x451, row-bound tiling forced"
JPEGBLK_ROW_TILING=1 bash tools/ab_bench.sh "$L" "679x451-420:512" 1
echo "--- 679x451, byte stores forced"
JPEGBLK_BYTE_STORE=1 bash tools/ab_bench.sh "$L" "679x451-420:512" 1
echo "--- 4:4:4 counterparts"
bash tools/ab_bench.sh "$L" "1024x448-444:512 679x451-444:512 1920x1080-444:128" 1
