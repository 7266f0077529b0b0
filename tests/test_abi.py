"""CPU tests of the C-ABI library (no compute calls: this container has no GPU): it loads, it
exports every symbol include/jpegblk.h declares, its pure-host entry points (geometry, table
resolution, the JFIF/Huffman front end, PPM sink) are correct, and the compute entry points
fail loudly -- there is no CPU fallback behind the ABI."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import BASELINE_IMAGES, GOLD, ROOT, load_golden


@pytest.fixture(scope="module")
def jb():
    import jpeg_decoder_amd as jb
    if not os.path.exists(jb.lib_path()):
        jb.build_library()
    return jb


def test_library_exports_every_declared_symbol(jb):
    header = open(os.path.join(ROOT, "include", "jpegblk.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = sorted(set(re.findall(r"\b(jb_[a-z_0-9]+)\s*\(", header)))
    assert len(declared) >= 20
    L = ctypes.CDLL(jb.lib_path())
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert L.jb_abi_version() == 1


def test_no_cpu_fallback(jb):
    """Without a HIP device the product refuses to compute (and says why)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(jb.JbError) as e:
        jb.Context(0, 1 << 20, 1 << 20, 2)
    assert e.value.status == -6 and "device" in str(e.value).lower()
    # and the product never links, loads or imports the oracle
    for root, _, files in os.walk(os.path.join(ROOT, "jpeg_decoder_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                text = open(os.path.join(root, f), errors="replace").read()
                assert "liboracle" not in text and "pyoracle" not in text and "jbo_" not in text, f


def test_geometry_matches_oracle(jb, oracle):
    from oracle.pyoracle import make_desc as odesc
    for hs in (1, 2):
        for vs in (1, 2):
            for (w, h) in [(1, 1), (8, 8), (9, 17), (679, 451), (1279, 885), (4096, 4096), (65535, 65535)]:
                a = jb.geometry_of(jb.make_desc(w, h, hs, vs))
                b = oracle.geometry(odesc(w, h, hs, vs))
                for f, _ in a._fields_:
                    assert getattr(a, f) == getattr(b, f), (f, w, h, hs, vs)
    for bad, status in [(jb.make_desc(0, 1, 1, 1), -2), (jb.make_desc(1, 65536, 1, 1), -2),
                        (jb.make_desc(8, 8, 4, 1), -3), (jb.make_desc(8, 8, 1, 1, (4, 0, 0)), -4)]:
        with pytest.raises(jb.JbError) as e:
            jb.geometry_of(bad)
        assert e.value.status == status


def test_resolve_qtabs(jb):
    q = np.arange(256, dtype=np.uint16).reshape(4, 64)
    out = jb.resolve_qtabs(jb.make_desc(8, 8, 1, 1, (2, 0, 3)), q)
    assert out.dtype == np.int32 and np.array_equal(out, q[[2, 0, 3]].astype(np.int32))


@pytest.mark.parametrize("name", BASELINE_IMAGES)
def test_front_end_matches_reference_coefficients(jb, manifest, name):
    """Host marker parser + Huffman decoder == the reference's decodeHuffman() output
    (golden coefficient dumps from the reference build), quant tables and frame geometry."""
    desc, coef, qtabs, _ = load_golden(name)
    data = open(os.path.join(GOLD, "images", name + ".jpg"), "rb").read()
    d2, q2, c2 = jb.entropy_decode(data)
    assert (d2.width, d2.height, d2.hs, d2.vs, list(d2.qtab_id)) == (desc.width, desc.height, desc.hs, desc.vs, list(desc.qtab_id))
    assert np.array_equal(q2, qtabs)
    assert c2.shape == coef.shape and np.array_equal(c2, coef)
    d3, q3, c3 = jb.entropy_decode(data, headers_only=True)
    assert c3 is None and d3.width == desc.width and np.array_equal(q3, qtabs)


def test_front_end_rejections(jb):
    """What the front end cannot decode comes back as a status (the reference exit(1)s,
    jpeg.cpp:69-87, 800-805): not-a-JPEG, truncation, and the frame types outside Huffman-coded
    8-bit 1- or 3-component DCT (12-bit, arithmetic coding, four components, chroma not 1x1)."""
    good = open(os.path.join(GOLD, "images", "img2.jpg"), "rb").read()
    cases = [(b"not a jpeg at all", -8), (b"", -8), (good[:200], -8), (good[:-2000], -8)]
    sof = good.index(b"\xff\xc0")
    four = bytearray(good)          # Nf = 4 (CMYK-style frame)
    four[sof + 9] = 4
    cases.append((bytes(four), -9))
    twelve = bytearray(good)        # 12-bit sample precision
    twelve[sof + 4] = 12
    cases.append((bytes(twelve), -9))
    arith = bytearray(good)         # SOF9: arithmetic-coded sequential
    arith[sof + 1] = 0xC9
    cases.append((bytes(arith), -9))
    bad = bytearray(good)           # chroma sampled 2x1
    bad[sof + 10 + 3 + 1] = 0x21
    cases.append((bytes(bad), -3))
    for data, status in cases:
        with pytest.raises(jb.JbError) as e:
            jb.entropy_decode(data)
        assert e.value.status == status, (status, str(e.value))


def test_front_end_capacity_and_corruption(jb):
    good = open(os.path.join(GOLD, "images", "img4.jpg"), "rb").read()
    buf = np.frombuffer(good, np.uint8)
    desc = jb.ImageDesc()
    q = np.zeros((4, 64), np.uint16)
    small = np.zeros(64, np.int16)
    rc = jb.lib().jb_entropy_decode(buf.ctypes.data, buf.size, ctypes.byref(desc), q.ctypes.data, small.ctypes.data, small.nbytes)
    assert rc == -5
    # flip bytes in the middle of the scan: must return (corrupt or merely different), never crash
    sos = good.index(b"\xff\xda")
    for k in range(20):
        dmg = bytearray(good)
        pos = sos + 20 + 997 * k
        dmg[pos] ^= 0x5A
        try:
            jb.entropy_decode(bytes(dmg))
        except jb.JbError as e:
            assert e.status == -8


def test_write_ppm(jb, tmp_path):
    rgb = np.arange(5 * 3 * 3, dtype=np.uint8).reshape(3, 15)
    p = tmp_path / "o.ppm"
    assert jb.lib().jb_write_ppm(str(p).encode(), rgb.ctypes.data, 5, 3, 15) == 0
    raw = p.read_bytes()
    assert raw.startswith(b"P6\n5 3\n255\n") and raw[len(b"P6\n5 3\n255\n"):] == rgb.tobytes()


@pytest.mark.parametrize("w,h", [(5, 3), (4, 4), (7, 2), (1, 1), (333, 17)])
def test_write_bmp_reads_back(jb, tmp_path, w, h):
    """saveToBMP replacement (reference jpeg.cpp:462-509 writes R,B,G behind an OS/2 header): the
    file must read back, in a standard reader, as exactly the RGB buffer -- with row padding and a
    source stride larger than the row."""
    from PIL import Image
    rng = np.random.default_rng(w * 100 + h)
    stride = 3 * w + 5
    buf = rng.integers(0, 256, (h, stride), dtype=np.uint8)
    p = tmp_path / "o.bmp"
    assert jb.lib().jb_write_bmp(str(p).encode(), buf.ctypes.data, w, h, stride) == 0
    assert p.stat().st_size == 54 + ((3 * w + 3) // 4 * 4) * h
    got = np.asarray(Image.open(p).convert("RGB"))
    assert np.array_equal(got, buf[:, :3 * w].reshape(h, w, 3))
    assert jb.lib().jb_write_bmp(str(p).encode(), buf.ctypes.data, w, h, 3 * w - 1) == -2
    assert jb.lib().jb_write_bmp(None, buf.ctypes.data, w, h, stride) == -1


def _pil_jpeg(img, **kw):
    import io
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(img).save(b, "JPEG", optimize=False, **kw)
    return b.getvalue()


def test_front_end_restart_intervals_any_length(jb):
    """Restart intervals are counted in MCUs (T.81), whatever their length: the same picture
    written with and without DRI (1, 2, 5, 11 MCUs per interval -- none a whole MCU row) decodes
    to identical coefficient blocks in 4:4:4, 4:2:2 and 4:2:0.  (The reference's own restart test,
    jpeg.cpp:414-419, only works for whole-row intervals such as its bundled img4.jpg, which
    test_front_end_matches_reference_coefficients pins.)"""
    pytest.importorskip("PIL")
    rng = np.random.default_rng(3)
    img = np.clip(np.cumsum(rng.normal(0, 6, (93, 157, 3)), axis=1) + 128, 0, 255).astype(np.uint8)
    for sub, hsvs in ((0, (1, 1)), (1, (2, 1)), (2, (2, 2))):
        d0, q0, c0 = jb.entropy_decode(_pil_jpeg(img, quality=85, subsampling=sub))
        assert (d0.hs, d0.vs) == hsvs and (d0.width, d0.height) == (157, 93)
        for blocks in (1, 2, 5, 11):
            data = _pil_jpeg(img, quality=85, subsampling=sub, restart_marker_blocks=blocks)
            assert b"\xff\xdd" in data
            d, q, c = jb.entropy_decode(data)
            assert np.array_equal(q, q0) and np.array_equal(c, c0), (sub, blocks)
        # a missing restart marker is an error, not a silent mis-decode
        data = bytearray(_pil_jpeg(img, quality=85, subsampling=sub, restart_marker_blocks=5))
        pos = data.index(b"\xff\xd0", data.index(b"\xff\xda"))
        data[pos + 1] = 0x00  # FF D0 -> FF 00 (a stuffed FF byte)
        with pytest.raises(jb.JbError):
            jb.entropy_decode(bytes(data))


def test_front_end_sixteen_bit_quant_table(jb):
    """Pq = 1 tables keep all 16 bits (the reference keeps the low byte, jpeg.cpp:216)."""
    pytest.importorskip("PIL")
    img = np.full((16, 16, 3), 128, np.uint8)
    data = bytearray(_pil_jpeg(img, quality=90, subsampling=0))
    dqt = data.index(b"\xff\xdb")
    seglen = (data[dqt + 2] << 8) | data[dqt + 3]
    assert data[dqt + 4] == 0x00  # Pq = 0, Tq = 0, 64 one-byte entries
    tab8 = bytes(data[dqt + 5:dqt + 5 + 64])
    tab16 = b"".join(bytes([1 if i == 5 else 0, v]) for i, v in enumerate(tab8))  # entry 5 += 256
    new_seg = b"\xff\xdb" + (2 + 1 + 128).to_bytes(2, "big") + b"\x10" + tab16
    rest = data[dqt + 2 + seglen:]
    if seglen > 2 + 65:  # a second table lives in the same segment: keep it as its own segment
        tail = bytes(data[dqt + 5 + 64:dqt + 2 + seglen])
        new_seg += b"\xff\xdb" + (2 + len(tail)).to_bytes(2, "big") + tail
    patched = bytes(data[:dqt]) + new_seg + bytes(rest)
    d, q, c = jb.entropy_decode(patched)
    d0, q0, c0 = jb.entropy_decode(bytes(data))
    want = q0.copy()
    zz5 = 2  # zig-zag position 5 is natural index 2
    want[0, zz5] += 256
    assert np.array_equal(q, want) and np.array_equal(c, c0)


def test_front_end_parallel_restart_intervals(jb):
    """jb_entropy_decode_mt: restart intervals of one image on several threads == serial decode
    (bundled img4.jpg: DRI = 100; PIL files: short intervals, ragged last interval); images
    without DRI and corrupt streams take the serial path and give the same answers/errors."""
    data = open(os.path.join(GOLD, "images", "img4.jpg"), "rb").read()
    _, coef, _, _ = load_golden("img4")
    for t in (2, 3, 8, 64):
        d, q, c = jb.entropy_decode(data, n_threads=t)
        assert np.array_equal(c, coef), t
    d, q, c = jb.entropy_decode(open(os.path.join(GOLD, "images", "img.jpg"), "rb").read(), n_threads=8)
    assert np.array_equal(c, load_golden("img")[1])  # no DRI: serial path
    pytest.importorskip("PIL")
    rng = np.random.default_rng(5)
    img = np.clip(np.cumsum(rng.normal(0, 4, (131, 259, 3)), axis=1) + 128, 0, 255).astype(np.uint8)
    for sub in (0, 2):
        for blocks in (1, 3, 16):
            f = _pil_jpeg(img, quality=80, subsampling=sub, restart_marker_blocks=blocks)
            _, _, c1 = jb.entropy_decode(f)
            for t in (2, 5):
                _, _, ct = jb.entropy_decode(f, n_threads=t)
                assert np.array_equal(ct, c1), (sub, blocks, t)
    # a damaged interval is still reported
    dmg = bytearray(_pil_jpeg(img, quality=80, subsampling=0, restart_marker_blocks=3))
    sos = dmg.index(b"\xff\xda")
    dmg[sos + 40:sos + 60] = b"\xff\xd9" * 10
    with pytest.raises(jb.JbError):
        jb.entropy_decode(bytes(dmg), n_threads=4)


def test_front_end_scan_without_eoi_and_alternating_tables(jb):
    """A buffer that ends right after the last entropy-coded byte (no EOI): the restart-interval
    threads and the serial path see the same last interval (the last byte used to be dropped from
    the parallel split).  And the per-thread Huffman table cache: files with different tables
    decoded alternately on one thread keep decoding to their own coefficients."""
    from jpeg_decoder_amd import synth
    w, h = 333, 211
    coef, q = synth.synth_blocks(w, h, 2, 2, 8)
    data = synth.encode_jpeg(coef, w, h, 2, 2, q, restart_interval=5)
    assert data[-2:] == b"\xff\xd9"
    cut = data[:-2]
    for t in (1, 4):
        _, _, c = jb.entropy_decode(cut, n_threads=t)
        assert np.array_equal(c, coef), t
    pytest.importorskip("PIL")
    rng = np.random.default_rng(9)
    img = np.clip(np.cumsum(rng.normal(0, 5, (97, 131, 3)), axis=1) + 128, 0, 255).astype(np.uint8)
    a = _pil_jpeg(img, quality=85, subsampling=0)                  # Annex K tables
    import io
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, "JPEG", quality=85, subsampling=0, optimize=True)   # this file's own tables
    b = buf.getvalue()
    assert a[a.index(b"\xff\xc4"):a.index(b"\xff\xda")] != b[b.index(b"\xff\xc4"):b.index(b"\xff\xda")]
    ca, cb = jb.entropy_decode(a)[2], jb.entropy_decode(b)[2]
    assert np.array_equal(ca, cb)                                   # same picture, same coefficients
    for _ in range(3):
        assert np.array_equal(jb.entropy_decode(a)[2], ca) and np.array_equal(jb.entropy_decode(b)[2], ca)
        assert np.array_equal(jb.entropy_decode(data)[2], coef)


# ---- stream-level round trips through the build's own baseline writer (tools/jpegwriter) ----

def test_annex_k_huffman_tables_match_libjpeg():
    """The writer's Huffman tables (synth.ANNEX_K_HUFFMAN, ITU-T T.81 Annex K.3) are the ones a
    libjpeg-written file carries in its DHT segments."""
    pytest.importorskip("PIL")
    from jpeg_decoder_amd import synth
    raw = _pil_jpeg(np.zeros((16, 16, 3), np.uint8), quality=90)
    tabs, i = {}, 2
    while raw[i + 1] != 0xDA:
        length = raw[i + 2] * 256 + raw[i + 3]
        if raw[i + 1] == 0xC4:
            seg, j = raw[i + 4:i + 2 + length], 0
            while j < len(seg):
                n = sum(seg[j + 1:j + 17])
                tabs[(seg[j] >> 4, seg[j] & 15)] = (list(seg[j + 1:j + 17]), list(seg[j + 17:j + 17 + n]))
                j += 17 + n
        i += 2 + length
    for key, (bits, vals) in zip([(0, 0), (1, 0), (0, 1), (1, 1)], synth.ANNEX_K_HUFFMAN):
        assert tabs[key] == (list(bits), list(vals)), key


@pytest.mark.parametrize("hs,vs", [(1, 1), (2, 1), (1, 2), (2, 2)])
def test_writer_round_trip_blocks_exact(jb, hs, vs):
    """blocks -> baseline stream -> jb_entropy_decode -> the same blocks, tables and geometry:
    ragged sizes, restart intervals of any length (1 MCU, not dividing the row, longer than the
    image), 8- and 16-bit DQT, separate chroma tables, serial and restart-parallel decoding."""
    from jpeg_decoder_amd import synth
    q = synth.annex_k_qtabs(75)
    q[2] = np.clip(q[1].astype(int) + 3, 1, 255)
    cases = [(1, 1, 0), (8, 8, 0), (17, 33, 1), (100, 60, 3), (333, 211, 7), (640, 360, 40), (515, 300, 100000 % 65536),
             (1920, 1080, 0)]
    for i, (w, h, ri) in enumerate(cases):
        qid = (0, 1, 2) if i % 2 else (0, 1, 1)
        coef, _ = synth.synth_blocks(w, h, hs, vs, i, qtabs=q, qtab_id=qid)
        for dqt16 in (False, True):
            data = synth.encode_jpeg(coef, w, h, hs, vs, q, qid, restart_interval=ri, dqt16=dqt16)
            assert data[:2] == b"\xff\xd8" and data[-2:] == b"\xff\xd9"
            for threads in (1, 4):
                desc, q2, c2 = jb.entropy_decode(data, n_threads=threads)
                assert (desc.width, desc.height, desc.hs, desc.vs, tuple(desc.qtab_id)) == (w, h, hs, vs, qid)
                assert np.array_equal(c2, coef), (w, h, ri, dqt16, threads)
                for t in set(qid):
                    assert np.array_equal(q2[t], q[t])


def test_writer_round_trip_extreme_values(jb):
    """Largest magnitudes the baseline alphabet can carry (DC difference +-2047, AC +-1023), long
    zero runs (ZRL), blocks that end exactly at coefficient 63 (no EOB), and 16-bit table entries."""
    from jpeg_decoder_amd import synth
    w, h = 64, 48
    n = synth.geometry(w, h, 1, 1)[3]
    rng = np.random.default_rng(5)
    coef = np.zeros((n, 64), np.int16)
    coef[:, 0] = np.where(np.arange(n) % 2, 1023, -1024)           # DC differences of +-2047
    coef[::3, 63] = rng.choice([-1023, 1023, 1, -1], size=len(coef[::3]))   # run of 62 zeros, no EOB
    coef[1::3, 17] = -1023
    coef[1::3, 40] = 512
    coef[2::3, 1:] = rng.integers(-1023, 1024, size=(len(coef[2::3]), 63))  # dense
    q = np.zeros((4, 64), np.uint16)
    q[0] = 1
    q[1] = np.arange(1, 65) * 1000 % 65535 + 1   # 16-bit entries
    data = synth.encode_jpeg(coef, w, h, 1, 1, q, (0, 1, 1), dqt16=True)
    desc, q2, c2 = jb.entropy_decode(data)
    assert np.array_equal(c2, coef) and np.array_equal(q2[:2], q[:2])
    with pytest.raises(ValueError):           # 8-bit DQT cannot hold those entries
        synth.encode_jpeg(coef, w, h, 1, 1, q, (0, 1, 1), dqt16=False)
    coef[5, 9] = 1024                          # outside the baseline AC alphabet
    with pytest.raises(ValueError):
        synth.encode_jpeg(coef, w, h, 1, 1, q, (0, 1, 1), dqt16=True)


def test_front_end_fuzz_under_sanitizers(tmp_path):
    """The host front end built for the CPU with AddressSanitizer + UBSan (tools/fuzz) must turn
    every mutated stream into a jb_status: a short run here (a 12-minute, 550k-mutant run was
    clean when this test was written); any sanitizer report fails the process."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    d = os.path.join(ROOT, "tools", "fuzz")
    b = subprocess.run(["make", "-C", d], capture_output=True, text=True)
    if b.returncode != 0 and ("cannot find -lasan" in b.stderr or "cannot find -lubsan" in b.stderr or "libasan" in b.stderr):
        pytest.skip("toolchain without sanitizer runtimes")
    assert b.returncode == 0, b.stderr[-2000:]
    seeds = sorted(os.path.join(GOLD, "images", f) for f in os.listdir(os.path.join(GOLD, "images")) if f.endswith(".jpg"))
    try:  # progressive and grayscale seeds for the general front end
        from PIL import Image
        img = _smooth_image(97, 61)
        for i, kw in enumerate([dict(progressive=True, subsampling=0), dict(progressive=True, subsampling=2, restart_marker_blocks=2)]):
            Image.fromarray(img).save(tmp_path / f"p{i}.jpg", "JPEG", quality=70, **kw)
            seeds.append(str(tmp_path / f"p{i}.jpg"))
        Image.fromarray(img).convert("L").save(tmp_path / "g.jpg", "JPEG", quality=70, progressive=True)
        seeds.append(str(tmp_path / "g.jpg"))
    except ImportError:
        pass
    r = subprocess.run([os.path.join(d, "fuzz_frontend"), "4", "12345"] + seeds, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    n = int(r.stdout.split()[0])
    assert n > 500, r.stdout


def test_unstuff_word_loop_equals_the_byte_loop():
    """csrc/jb_entropy.h unstuff(): the PEXT word loop and the memchr loop, switched per 4 KiB by the density of
    0xFF, against the one-byte-at-a-time rule (reference file.hpp:59-104) on random scans with restart markers,
    fill bytes, EOI in the middle and a lone 0xFF at the end; under ASan + UBSan with exact-length buffers."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    d = os.path.join(ROOT, "tools", "fuzz")
    b = subprocess.run(["make", "-C", d, "unstuff_check"], capture_output=True, text=True)
    if b.returncode != 0 and ("cannot find -lasan" in b.stderr or "cannot find -lubsan" in b.stderr or "libasan" in b.stderr):
        pytest.skip("toolchain without sanitizer runtimes")
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([os.path.join(d, "unstuff_check"), "4", "2024"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert int(r.stdout.split()[1]) > 1000, r.stdout


# ---- beyond the reference: progressive, grayscale (csrc/jb_frontend_ext.cpp) ----

def _smooth_image(w, h, seed=3):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    return np.clip(np.stack([xx * 0.6 + yy * 0.2, 200 - yy * 0.7, (xx + yy) * 0.35 + 30], -1)
                   + rng.normal(0, 6, (h, w, 3)), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("sub", [0, 1, 2])
def test_progressive_equals_baseline_coefficients(jb, sub):
    """The pin for the progressive decoder (the reference rejects SOF2, jpeg.cpp:69-73): libjpeg's
    encoder is deterministic, so the baseline and the progressive encoding of the same pixels
    hold the same quantised coefficients -- and the baseline file goes through the front end that
    is integer-exact against the reference.  All four scan types of libjpeg's default script (DC
    first/refine, AC first/refine with EOB runs), three subsamplings, ragged sizes, two qualities,
    with and without restart intervals."""
    pytest.importorskip("PIL")
    for (w, h) in [(333, 211), (64, 64), (17, 9), (640, 360), (8, 8)]:
        img = _smooth_image(w, h)
        for q in (92, 30):
            for restart in (0, 3):
                kw = {"restart_marker_blocks": restart} if restart else {}
                base = _pil_jpeg(img, quality=q, subsampling=sub, **kw)
                import io
                from PIL import Image
                b = io.BytesIO()
                Image.fromarray(img).save(b, "JPEG", quality=q, subsampling=sub, progressive=True, **kw)
                prog = b.getvalue()
                assert b"\xff\xc2" in prog
                d0, q0, c0 = jb.entropy_decode(base)
                d1, q1, c1 = jb.entropy_decode(prog)
                assert (d0.width, d0.height, d0.hs, d0.vs, tuple(d0.qtab_id)) == (d1.width, d1.height, d1.hs, d1.vs, tuple(d1.qtab_id))
                assert np.array_equal(q0, q1) and np.array_equal(c0, c1), (w, h, q, restart)
                dh, qh, _ = jb.entropy_decode(prog, headers_only=True)
                assert (dh.width, dh.height, dh.hs, dh.vs) == (w, h, d0.hs, d0.vs) and np.array_equal(qh, q0)


def test_grayscale_frames(jb, oracle):
    """One-component frames (rejected by the reference, jpeg.cpp:83-87) are delivered as 4:4:4
    with all-zero Cb and Cr blocks, so that the reference's colour formulas give R = G = B =
    Y + 128: baseline and progressive agree block for block, and the picture is libjpeg's picture
    up to the IDCT difference."""
    pytest.importorskip("PIL")
    import io
    from PIL import Image
    from oracle.pyoracle import make_desc
    gray = np.asarray(Image.fromarray(_smooth_image(203, 117)).convert("L"))
    outs = []
    for kw in ({}, {"progressive": True}):
        b = io.BytesIO()
        Image.fromarray(gray).save(b, "JPEG", quality=88, **kw)
        desc, q, coef = jb.entropy_decode(b.getvalue())
        assert (desc.width, desc.height, desc.hs, desc.vs) == (203, 117, 1, 1)
        assert not coef[1::3].any() and not coef[2::3].any() and coef[0::3].any()
        rgb = oracle.blocks_to_rgb(make_desc(203, 117, 1, 1, list(desc.qtab_id)), coef, q)
        assert np.array_equal(rgb[..., 0], rgb[..., 1]) and np.array_equal(rgb[..., 0], rgb[..., 2])
        ref = np.asarray(Image.open(io.BytesIO(b.getvalue())).convert("L")).astype(int)
        assert np.abs(rgb[..., 0].astype(int) - ref).max() <= 3
        outs.append(coef)
    assert np.array_equal(outs[0], outs[1])


def test_bundled_progressive_sample_decodes(jb, oracle):
    """The progressive file the reference ships but cannot read (images/prograssive-sample-2.jpg):
    decoded blocks through the oracle's pixel path stay close to libjpeg's decode of the file."""
    pytest.importorskip("PIL")
    from PIL import Image
    from oracle.pyoracle import make_desc
    path = os.path.join(GOLD, "images", "prograssive-sample-2.jpg")
    desc, q, coef = jb.entropy_decode(open(path, "rb").read())
    rgb = oracle.blocks_to_rgb(make_desc(desc.width, desc.height, desc.hs, desc.vs, list(desc.qtab_id)), coef, q, nthreads=4)
    ref = np.asarray(Image.open(path).convert("RGB")).astype(int)
    assert rgb.shape == ref.shape
    d = np.abs(rgb.astype(int) - ref)
    # different IDCT arithmetic and, for subsampled chroma, replication here against libjpeg's
    # "fancy" interpolation: close on average, bounded at edges
    assert d.mean() < 2.5, d.mean()


def test_progressive_truncation_and_corruption(jb):
    pytest.importorskip("PIL")
    import io
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(_smooth_image(160, 120)).save(b, "JPEG", quality=80, progressive=True)
    good = b.getvalue()
    jb.entropy_decode(good)
    sos = good.index(b"\xff\xda")
    with pytest.raises(jb.JbError) as e:
        jb.entropy_decode(good[:sos + 40])
    assert e.value.status == -8
    for k in range(40):
        dmg = bytearray(good)
        dmg[sos + 14 + 53 * k] ^= 0xA5
        try:
            jb.entropy_decode(bytes(dmg))
        except jb.JbError as err:
            assert err.status in (-8, -9, -3, -4, -2)


@pytest.mark.parametrize("hs,vs", [(1, 1), (2, 1), (1, 2), (2, 2)])
def test_sequential_frames_in_per_component_scans(jb, hs, vs):
    """A baseline frame coded as three non-interleaved scans (legal T.81, rejected by the reference,
    jpeg.cpp:255-264): every block inside a component's own grid round-trips exactly; blocks that
    only pad the frame to whole MCUs are not coded in such scans and come back as zeros; libjpeg
    reads the same file."""
    from jpeg_decoder_amd import synth
    for i, (w, h, ri) in enumerate([(64, 48, 0), (100, 60, 3), (37, 29, 0), (333, 211, 7), (8, 8, 0)]):
        coef, q = synth.synth_blocks(w, h, hs, vs, 40 + i)
        data = synth.encode_jpeg(coef, w, h, hs, vs, q, restart_interval=ri, per_component_scans=True)
        desc, q2, got = jb.entropy_decode(data)
        assert (desc.width, desc.height, desc.hs, desc.vs) == (w, h, hs, vs) and np.array_equal(q2[:2], q[:2])
        mx, my, bpm, _ = synth.geometry(w, h, hs, vs)
        want = coef.reshape(my, mx, bpm, 64).copy()
        bw, bh = (w + 7) // 8, (h + 7) // 8
        cbw, cbh = ((w + hs - 1) // hs + 7) // 8, ((h + vs - 1) // vs + 7) // 8
        for y in range(my):
            for x in range(mx):
                for s in range(hs * vs):
                    if y * vs + s // hs >= bh or x * hs + s % hs >= bw:
                        want[y, x, s] = 0
                if y >= cbh or x >= cbw:
                    want[y, x, hs * vs:] = 0
        assert np.array_equal(got, want.reshape(-1, 64)), (w, h, ri)
        try:
            import io
            from PIL import Image
            assert Image.open(io.BytesIO(data)).convert("RGB").size == (w, h)
        except ImportError:
            pass
