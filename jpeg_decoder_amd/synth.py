"""Seeded synthetic inputs for the block pipeline (generator A).

Block-level generator: produces what the host Huffman stage would hand to the device -- packed
int16 coefficient blocks in decode (MCU-interleaved) order plus natural-order quantisation
tables -- without any bitstream.  Statistics follow the bundled images of the reference
(64-90 % zero coefficients in luma): DC ~ U[-64,64], AC at zig-zag
position k is non-zero with p = 0.9*exp(-k/6), magnitude geometric, clipped so that
|coef*q| <= 1023.  Tables are the JPEG Annex K luminance/chrominance tables scaled to
quality 90 (libjpeg scaling), de-zigzagged the way reference types.hpp:86-92 stores them.
"""
import numpy as np

SEED_BASE = 0x4A504547  # "JPEG"

# zig-zag scan position -> natural (row-major) index, ITU-T T.81 Figure A.6
ZIGZAG = np.array([
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5,
    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63], dtype=np.int64)

# ITU-T T.81 Annex K, Tables K.1 / K.2, natural order
_K1_LUMA = np.array([
    16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55,
    14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
    18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99], dtype=np.int64)
_K2_CHROMA = np.array([
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
    24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99], dtype=np.int64)


def annex_k_qtabs(quality=90):
    """-> uint16 [4, 64] natural order; table 0 = luma, 1 = chroma, 2..3 = zeros (absent)."""
    scale = 5000 // quality if quality < 50 else 200 - 2 * quality
    q = np.zeros((4, 64), np.uint16)
    for t, base in enumerate((_K1_LUMA, _K2_CHROMA)):
        q[t] = np.clip((base * scale + 50) // 100, 1, 255)
    return q


def geometry(width, height, hs, vs):
    """(mcus_x, mcus_y, blocks_per_mcu, n_coded_blocks) as reference read_sof derives them
    (jpeg.cpp:77-80, 118-125)."""
    mcu_w, mcu_h = (width + 7) // 8, (height + 7) // 8
    mcu_w_real = mcu_w + (1 if hs == 2 and mcu_w % 2 else 0)
    mcu_h_real = mcu_h + (1 if vs == 2 and mcu_h % 2 else 0)
    mcus_x, mcus_y = mcu_w_real // hs, mcu_h_real // vs
    bpm = hs * vs + 2
    return mcus_x, mcus_y, bpm, mcus_x * mcus_y * bpm


def synth_blocks(width, height, hs, vs, image_index=0, qtabs=None, qtab_id=(0, 1, 1), dense=False):
    """-> (coef int16 [n_coded_blocks, 64] natural order / decode order, qtabs uint16 [4, 64]).

    dense=True makes every coefficient non-zero (worst case for any zero-shortcut)."""
    if qtabs is None:
        qtabs = annex_k_qtabs(90)
    mcus_x, mcus_y, bpm, n = geometry(width, height, hs, vs)
    rng = np.random.default_rng(SEED_BASE + image_index)
    k = np.arange(64)
    p = 0.9 * np.exp(-k / 6.0)
    if dense:
        p = np.ones(64)
    p[0] = 1.0
    coef_zz = np.zeros((n, 64), np.int32)
    chunk = 1 << 16
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        m = e - s
        nz = rng.random((m, 64)) < p
        mag = rng.geometric(0.35, size=(m, 64))
        sign = rng.integers(0, 2, size=(m, 64)) * 2 - 1
        c = np.where(nz, mag * sign, 0)
        c[:, 0] = rng.integers(-64, 65, size=m)
        coef_zz[s:e] = c
    # clip in natural order so |coef * q| <= 1023 for the table of the block's component
    coef = np.zeros((n, 64), np.int32)
    coef[:, ZIGZAG] = coef_zz
    slot = np.arange(n) % bpm
    comp = np.where(slot < bpm - 2, 0, slot - (bpm - 2) + 1)
    for c in range(3):
        lim = 1023 // np.maximum(qtabs[qtab_id[c]].astype(np.int64), 1)
        rows = comp == c
        coef[rows] = np.clip(coef[rows], -lim, lim)
    return coef.astype(np.int16), qtabs


def random_blocks(n_blocks, seed, lo=-32768, hi=32767):
    """Uniform full-range int16 blocks (adversarial: exercises every truncation path)."""
    rng = np.random.default_rng(seed)
    return rng.integers(lo, hi + 1, size=(n_blocks, 64), dtype=np.int64).astype(np.int16)
