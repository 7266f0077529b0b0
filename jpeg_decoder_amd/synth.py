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


# ---- stream-level generator (generator B) ------------------------------
# ITU-T T.81 Annex K.3 "typical" Huffman tables as (BITS[16], HUFFVAL) -- what a DHT segment holds.
# tests/test_abi.py checks them against the DHT segments of a libjpeg-written file.
_DC_VALS = list(range(12))
_AC_LUMA_VALS = [
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07,
    0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0,
    0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28,
    0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49,
    0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69,
    0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89,
    0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7,
    0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5,
    0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
    0xf9, 0xfa]
_AC_CHROMA_VALS = [
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71,
    0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0,
    0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26,
    0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
    0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68,
    0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87,
    0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5,
    0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
    0xf9, 0xfa]
# order: DC luminance, AC luminance, DC chrominance, AC chrominance (K.3.1, K.5, K.4, K.6)
ANNEX_K_HUFFMAN = (
    ([0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0], _DC_VALS),
    ([0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d], _AC_LUMA_VALS),
    ([0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0], _DC_VALS),
    ([0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77], _AC_CHROMA_VALS),
)

_writer = None


def _writer_lib():
    """tools/jpegwriter/libjpegwriter.so (test/bench infrastructure), built on first use."""
    global _writer
    if _writer is None:
        import ctypes
        import os
        import subprocess
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "jpegwriter")
        so = os.path.join(d, "libjpegwriter.so")
        if not os.path.exists(so):
            subprocess.run(["make", "-C", d], check=True, capture_output=True)
        L = ctypes.CDLL(so)
        vp = ctypes.c_void_p
        L.jw_encode_ex.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp,
                                   ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_long]
        L.jw_encode_ex.restype = ctypes.c_long
        _writer = L
    return _writer


def encode_jpeg(coef, width, height, hs, vs, qtabs, qtab_id=(0, 1, 1), restart_interval=0, dqt16=False,
                huffman=ANNEX_K_HUFFMAN, per_component_scans=False):
    """Coefficient blocks (int16 [n, 64], natural order, decode order -- the device seam's input)
    -> a baseline JFIF byte stream that entropy-decodes to exactly those blocks.  restart_interval
    in MCUs.  Requires |DC difference| <= 2047 and |AC| <= 1023 (the baseline symbol alphabet).
    per_component_scans: three non-interleaved scans instead of one interleaved scan (the
    reference cannot read those; blocks that only pad the frame to whole MCUs are then not coded)."""
    import ctypes
    coef = np.ascontiguousarray(coef, np.int16)
    n = geometry(width, height, hs, vs)[3]
    if coef.shape != (n, 64):
        raise ValueError(f"expected coef [{n}, 64], got {coef.shape}")
    q = np.ascontiguousarray(qtabs, np.uint16)
    ids = np.asarray(qtab_id, np.int32)
    dht = np.zeros((4, 272), np.uint8)
    for t, (bits, vals) in enumerate(huffman):
        if sum(bits) != len(vals):
            raise ValueError("BITS does not match HUFFVAL")
        dht[t, :16] = bits
        dht[t, 16:16 + len(vals)] = vals
    cap = 1024 + coef.size * 4  # worst case: 16-bit code + 10/11 value bits per coefficient, stuffed
    out = np.empty(cap, np.uint8)
    r = _writer_lib().jw_encode_ex(coef.ctypes.data, width, height, hs, vs, q.ctypes.data, ids.ctypes.data,
                                   dht.ctypes.data, restart_interval, int(bool(dqt16)), int(bool(per_component_scans)),
                                   out.ctypes.data, cap)
    if r < 0:
        raise ValueError({-1: "bad argument", -2: "output buffer too small", -3: "value not encodable in baseline"}.get(r, str(r)))
    return out[:r].tobytes()
