"""The device entropy decoder's KERNELS on the host (no GPU): tools/huff_emu compiles
jpeg_decoder_amd/csrc/jb_huff.hip -- the kernels' own text -- against a small SIMT shim (one OS thread per
lane, barriers and wave exchanges through std::barrier) and runs it on JPEG files; the coefficients must
equal the host decoder's (itself pinned to the reference's decodeHuffman() dumps, tests/test_abi.py).
This checks the pass structure, the checkpoints, the scans and the verification of jb_huff.hip wherever
the CPU suite runs; the same cases run on the MI355X in tests/test_gpu_huffman.py."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLD

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tools", "huff_emu")
EMU = os.path.join(EMU_DIR, "huff_emu")


@pytest.fixture(scope="module")
def emu():
    import jpeg_decoder_amd as jb
    jb.lib()  # libjpegblk.so must exist: the emulator links its host side (jb_huff_prepare_, jb_huff_pack_, jb_entropy_decode)
    r = subprocess.run(["make", "-C", EMU_DIR], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return EMU


def run(emu, paths, chunk=None, strict=True, launches=None):
    env = dict(os.environ)
    env.pop("JPEGBLK_CHUNK_BYTES", None)
    if chunk:
        env["JPEGBLK_CHUNK_BYTES"] = str(chunk)
    cmd = [emu, "--quiet"] + (["--strict"] if strict else []) + (["--launches", str(launches)] if launches else []) + list(paths)
    return subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)


def test_reference_images_through_the_emulated_kernels(emu):
    """The reference's six bundled baseline images (img4 has DRI = 100), all in ONE submission: status 0 and
    coefficients equal to the host decoder's."""
    paths = [os.path.join(GOLD, "images", n + ".jpg") for n in ("img", "img2", "img3", "img4", "img5", "img6")]
    for chunk in (128, 64):
        r = run(emu, paths, chunk)
        assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.parametrize("hs,vs", [(1, 1), (2, 1), (1, 2), (2, 2)])
def test_writer_streams_every_layout(emu, tmp_path, hs, vs):
    """Writer-made streams of every sampling layout: no restart intervals, intervals of one MCU (a chunk each:
    only the writing pass runs), of 7 MCUs and of two MCU rows (several chunks per interval), ragged sizes."""
    from jpeg_decoder_amd import synth
    paths = []
    for (w, h, ris) in [(333, 211, (0, 1, 7)), (97, 61, (0, 3)), (640, 360, (0, 2 * ((640 + 8 * hs - 1) // (8 * hs)),))]:
        coef, q = synth.synth_blocks(w, h, hs, vs, 11)
        for ri in ris:
            p = tmp_path / f"w{w}_{ri}.jpg"
            p.write_bytes(synth.encode_jpeg(coef, w, h, hs, vs, q, restart_interval=ri))
            paths.append(str(p))
    for chunk in (128, 64):
        r = run(emu, paths, chunk)
        assert r.returncode == 0, r.stdout + r.stderr


def three_table_variant(data):
    """The same stream with the Cr component on Huffman tables of its own (ids 2: copies of the Cb tables, ids 1):
    a frame may name three tables of each kind (reference jpeg.cpp:148-196 reads ids up to 3)."""
    d = bytearray(data)
    sos = d.index(b"\xff\xda")
    tabs, i = {}, 2
    while i < sos:  # the DHT segments in front of the scan
        assert d[i] == 0xff
        m, ln = d[i + 1], (d[i + 2] << 8) | d[i + 3]
        if m == 0xc4:
            j = i + 4
            while j < i + 2 + ln:
                n = sum(d[j + 1:j + 17])
                tabs[d[j]] = bytes(d[j + 1:j + 17 + n])
                j += 17 + n
        i += 2 + ln
    extra = b"".join(bytes([tc_th + 1]) + tabs[tc_th] for tc_th in (0x01, 0x11))
    assert d[sos + 4] == 3 and d[sos + 10] == 0x11   # Ns = 3; the third component's selectors
    d[sos + 10] = 0x22
    return bytes(d[:sos]) + b"\xff\xc4" + (2 + len(extra)).to_bytes(2, "big") + extra + bytes(d[sos:])


def test_grayscale_and_three_tables_per_kind(emu, tmp_path):
    """Single-component baseline frames (one block per MCU in the scan, delivered as the Y blocks of a 4:4:4 frame),
    with and without restart intervals, Annex-K and optimised tables; and a frame whose three components name three
    different tables of each kind."""
    pytest.importorskip("PIL")
    import io
    from PIL import Image
    from jpeg_decoder_amd import synth
    rng = np.random.default_rng(4)
    g = np.clip(np.cumsum(rng.normal(0, 5, (397, 531)), axis=1) + 128, 0, 255).astype(np.uint8)
    paths = []
    for name, kw in (("gray", {}), ("gray_dri", {"restart_marker_blocks": 5}), ("gray_opt", {"optimize": True})):
        b = io.BytesIO()
        Image.fromarray(g).save(b, "JPEG", quality=88, **kw)
        (tmp_path / (name + ".jpg")).write_bytes(b.getvalue())
        paths.append(str(tmp_path / (name + ".jpg")))
    for (hs, vs) in ((1, 1), (2, 2)):
        coef, q = synth.synth_blocks(333, 211, hs, vs, 5)
        p = tmp_path / f"three_{hs}{vs}.jpg"
        p.write_bytes(three_table_variant(synth.encode_jpeg(coef, 333, 211, hs, vs, q)))
        paths.append(str(p))
    for chunk in (128, 64):
        r = run(emu, paths, chunk)
        assert r.returncode == 0 and "not taken" not in r.stdout, r.stdout + r.stderr


def test_dense_random_data_needs_the_second_launch_or_flags(emu, tmp_path):
    """Uniform random coefficients in the full baseline alphabet (16-bit codes with 10 magnitude bits, hardly any EOB
    to fall into step at): the second-level tables run for most symbols, and lanes rarely meet their previous paths.
    Whatever the kernels accept must be exact; with enough launches they must accept it."""
    from jpeg_decoder_amd import synth
    rng = np.random.default_rng(7)
    w, h = 512, 384
    n = synth.geometry(w, h, 2, 2)[3]
    coef = rng.integers(-1023, 1024, size=(n, 64)).astype(np.int16)
    coef[:, 0] = rng.integers(-1000, 1001, size=n)
    coef[rng.random((n, 64)) < 0.3] = 0
    coef[::3, 20:60] = 0
    p = tmp_path / "dense.jpg"
    p.write_bytes(synth.encode_jpeg(coef, w, h, 2, 2, synth.annex_k_qtabs(75)))
    r = run(emu, [str(p)], strict=False)
    assert r.returncode == 0, r.stdout + r.stderr
    r = run(emu, [str(p)], strict=True, launches=8)
    assert r.returncode == 0, r.stdout + r.stderr


def test_mutants_never_accepted_wrongly(emu, tmp_path):
    """A short run of the mutation fuzzer (tools/huff_emu/fuzz_emu.py): damaged scans, tables, restart intervals,
    truncations -- what the kernels accept, the host decoder accepts with the same coefficients."""
    r = subprocess.run([sys.executable, os.path.join(EMU_DIR, "fuzz_emu.py"), "--n", "48", "--seed", "9"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
