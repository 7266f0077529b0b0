"""GPU parity of the BATCH form of decode(path) at BASELINE.json's batch configurations (-m gpu):
jb_batch_decoder (single- and multi-device) over writer-made JPEG files, every decoded image
compared bit-exactly with the oracle on the blocks that were written.
  config 4: 1920x1080 4:4:4 files (>= 32), restart intervals on;
  config 5: 8192x8192 4:2:0 files (>= 4), restart intervals on;
  mixed:    one large 4:4:4 file next to many small 4:2:0 files (groups bounded by the pixel side).
Each in malloc and arena output modes, >= 8 host threads, grouped and one-image-per-submission."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def jb():
    import jpeg_decoder_amd as jb
    assert jb.lib().jb_device_count() >= 1, jb.lib().jb_last_error(None)
    return jb


def _write_files(tmp, tag, w, h, hs, vs, n_files, n_distinct, ri, oracle):
    """n_files distinct images: n_distinct seeded synth images, the rest MCU-rotations of them.
    -> (paths, expected RGB arrays)."""
    from jpeg_decoder_amd import synth
    from oracle.pyoracle import make_desc as odesc
    bpm = hs * vs + 2
    bases = [synth.synth_blocks(w, h, hs, vs, 300 + i) for i in range(n_distinct)]
    paths, want = [], []
    for i in range(n_files):
        coef, q = bases[i % n_distinct]
        if i >= n_distinct:
            coef = np.roll(coef, (i * 131) * bpm, axis=0)
        p = os.path.join(tmp, f"{tag}_{i}.jpg")
        with open(p, "wb") as f:
            f.write(synth.encode_jpeg(coef, w, h, hs, vs, q, restart_interval=ri))
        paths.append(p)
        want.append(oracle.blocks_to_rgb(odesc(w, h, hs, vs), coef, q, nthreads=16))
    return paths, want


def _check(imgs, st, tm, want):
    assert tm["rc"] == 0 and all(s == 0 for s in st), (tm["rc"], tm["error"], st)
    for i, (g, w) in enumerate(zip(imgs, want)):
        assert g is not None and g.shape == w.shape and np.array_equal(g, w), i


@pytest.fixture(scope="module")
def files_1080p(tmp_path_factory, oracle):
    d = str(tmp_path_factory.mktemp("c4"))
    return _write_files(d, "c4", 1920, 1080, 1, 1, 32, 4, 240, oracle)  # DRI = one MCU row


@pytest.fixture(scope="module")
def files_8192(tmp_path_factory, oracle):
    d = str(tmp_path_factory.mktemp("c5"))
    return _write_files(d, "c5", 8192, 8192, 2, 2, 4, 1, 512, oracle)  # DRI = one MCU row


@pytest.mark.parametrize("entropy", ["host", "device"])
@pytest.mark.parametrize("devices", [None, [0, 0]])
@pytest.mark.parametrize("arena", [False, True])
def test_batch_decoder_config4_1080p_444(jb, files_1080p, monkeypatch, arena, devices, entropy):
    """BASELINE config 4 (one GPU's share, as files): 32 x 1920x1080 4:4:4, 8 threads.  The files
    carry restart intervals (one per MCU row): with the entropy stage on the host threads
    (JPEGBLK_GPU_HUFFMAN=0, north_star's split) and on the device (the batch decoder's default)."""
    paths, want = files_1080p
    total = sum((w.size + 255) // 256 * 256 for w in want)
    if entropy == "host":  # north_star's split: Huffman on the host threads
        monkeypatch.setenv("JPEGBLK_GPU_HUFFMAN", "0")
    else:                  # the batch decoder's default
        monkeypatch.delenv("JPEGBLK_GPU_HUFFMAN", raising=False)
    for group_mb in (None, "0"):
        if group_mb is None:
            monkeypatch.delenv("JPEGBLK_GROUP_MB", raising=False)
        else:
            monkeypatch.setenv("JPEGBLK_GROUP_MB", group_mb)
        with jb.BatchDecoder(8, 0, arena_bytes=total if arena else 0, devices=devices) as dec:
            for _ in range(2):  # second run: buffers, ring and arena are reused
                imgs, st, tm = dec.run(paths)
                _check(imgs, st, tm, want)
            assert dec.device_entropy_images == (2 * len(paths) if entropy == "device" else 0)


@pytest.mark.parametrize("entropy", ["host", "device"])
@pytest.mark.parametrize("devices", [None, [0, 0]])
@pytest.mark.parametrize("arena", [False, True])
def test_batch_decoder_config5_8192_420(jb, files_8192, monkeypatch, arena, devices, entropy):
    """BASELINE config 5 shape: 8192x8192 4:2:0 files (201 MB of coefficients and of pixels per
    image), 8 threads.  Entropy stage on the host -- "host Huffman on all cores overlapped with
    device IDCT", restart-interval splitting on (more host threads than files) -- and on the device."""
    paths, want = files_8192
    total = sum((w.size + 255) // 256 * 256 for w in want)
    if entropy == "host":  # north_star's split: Huffman on the host threads
        monkeypatch.setenv("JPEGBLK_GPU_HUFFMAN", "0")
    else:                  # the batch decoder's default
        monkeypatch.delenv("JPEGBLK_GPU_HUFFMAN", raising=False)
    with jb.BatchDecoder(8, 0, arena_bytes=total if arena else 0, devices=devices) as dec:
        imgs, st, tm = dec.run(paths)
        _check(imgs, st, tm, want)
        assert dec.device_entropy_images == (len(paths) if entropy == "device" else 0)
    if not arena and devices is None:  # ring back-pressure: 2 threads, ring of 4 slots, 8 submissions
        with jb.BatchDecoder(2, 0) as dec:
            imgs, st, tm = dec.run(paths + paths)
            _check(imgs, st, tm, want + want)


@pytest.mark.parametrize("entropy", ["host", "device"])
@pytest.mark.parametrize("devices", [None, [0, 0]])
@pytest.mark.parametrize("threads", [1, 8])
@pytest.mark.parametrize("arena", [False, True])
def test_batch_decoder_mixed_sizes_and_samplings(jb, oracle, tmp_path, monkeypatch, arena, threads, devices, entropy):
    """One 2048x1536 4:4:4 file (18.9 MB of coefficients, 9.4 MB of pixels) among 44 files of
    679x451 4:2:0: the ring slots are sized by the large image, and a group of small 4:2:0 images
    that fits its coefficient capacity (19 images) would overflow its pixel capacity (10) -- groups
    are bounded by both."""
    if entropy == "device":  # (the default) no restart intervals in these files: the self-synchronising decoder
        monkeypatch.delenv("JPEGBLK_GPU_HUFFMAN", raising=False)
    else:
        monkeypatch.setenv("JPEGBLK_GPU_HUFFMAN", "0")
    big_p, big_w = _write_files(str(tmp_path), "big", 2048, 1536, 1, 1, 1, 1, 0, oracle)
    small_p, small_w = _write_files(str(tmp_path), "small", 679, 451, 2, 2, 44, 3, 0, oracle)
    paths = small_p[:5] + big_p + small_p[5:]
    want = small_w[:5] + big_w + small_w[5:]
    total = sum((w.size + 255) // 256 * 256 for w in want)
    with jb.BatchDecoder(threads, 0, arena_bytes=total if arena else 0, devices=devices) as dec:
        imgs, st, tm = dec.run(paths)
        _check(imgs, st, tm, want)
        assert dec.device_entropy_images == (len(paths) if entropy == "device" else 0)


def test_batch_decoder_device_entropy_on_the_reference_images(jb, monkeypatch):
    """The reference's six bundled baseline images (five without restart markers: the
    self-synchronising decoder; img4 with DRI = 100: the interval decoder), three times over, through
    the batch decoder with the entropy stage on the device: the golden pixels of the reference."""
    from conftest import BASELINE_IMAGES, GOLD, load_golden
    monkeypatch.delenv("JPEGBLK_GPU_HUFFMAN", raising=False)
    names = BASELINE_IMAGES * 3
    paths = [os.path.join(GOLD, "images", n + ".jpg") for n in names]
    with jb.BatchDecoder(4, 0) as dec:
        imgs, st, tm = dec.run(paths)
        assert tm["rc"] == 0 and all(x == 0 for x in st), (tm, st)
        assert dec.device_entropy_images == len(paths)
    for n, got in zip(names, imgs):
        assert np.array_equal(got, load_golden(n)[3]), n


def test_multi_device_decoder_rejects_and_reports(jb, tmp_path):
    """jb_batch_decoder_create_multi: a device that does not exist is reported at creation; a
    failing file is reported per file at its original index, the rest decode."""
    from conftest import BASELINE_IMAGES, GOLD, load_golden
    with pytest.raises(jb.JbError) as e:
        jb.BatchDecoder(4, devices=[0, 99])
    assert e.value.status == -6
    names = BASELINE_IMAGES + BASELINE_IMAGES[:3]
    paths = [os.path.join(GOLD, "images", n + ".jpg") for n in names]
    paths.insert(3, str(tmp_path / "missing.jpg"))
    with jb.BatchDecoder(4, devices=[0, 0, 0]) as dec:
        imgs, st, tm = dec.run(paths)
    assert tm["rc"] == -8 and st[3] == -8 and imgs[3] is None
    k = 0
    for i in range(len(paths)):
        if i == 3:
            continue
        assert st[i] == 0 and np.array_equal(imgs[i], load_golden(names[k])[3]), i
        k += 1


def test_context_sizes_itself_from_the_frame(jb):
    """A context created without staging (0,0) -- the Python and C++ default -- decodes files: the
    ring is built from the parsed frame and grows with larger frames (jb_ctx_reserve)."""
    from conftest import GOLD, load_golden
    with jb.Context(0) as ctx:
        assert ctx.device == 0
        for name in ("img2", "img5", "img2"):  # small, larger (re-size), small again
            got = ctx.decode_file(os.path.join(GOLD, "images", name + ".jpg"))
            assert np.array_equal(got, load_golden(name)[3]), name
    node = jb.lib().jb_device_numa_node(0)
    assert node >= 0 or node == -7  # known, or reported as unknown (JB_ERR_STATE)
    p = jb.lib().jb_pinned_alloc_on(0, 1 << 20)
    assert p
    jb.lib().jb_pinned_free(p)
    assert not jb.lib().jb_pinned_alloc_on(99, 1 << 20)


def test_batch_decoder_device_entropy_falls_back_per_image(jb, oracle, tmp_path, monkeypatch):
    """A batch of DRI files of one geometry with one damaged file in the middle: the group goes
    through the device entropy decoder, the damaged image is flagged by its status word and handed to
    the host decoder (the authority), which rejects it; every other image decodes."""
    monkeypatch.delenv("JPEGBLK_GPU_HUFFMAN", raising=False)
    paths, want = _write_files(str(tmp_path), "dri", 640, 360, 2, 2, 9, 3, 10, oracle)
    data = bytearray(open(paths[4], "rb").read())
    sos = data.index(b"\xff\xda")
    for k in range(300, 340):       # garbage inside the scan, marker structure intact
        if data[sos + k] != 0xff and data[sos + k - 1] != 0xff:
            data[sos + k] = (data[sos + k] * 7 + 13) % 255
    open(paths[4], "wb").write(bytes(data))
    try:
        jb.entropy_decode(bytes(data))
        host_accepts = True
    except jb.JbError:
        host_accepts = False
    with jb.BatchDecoder(2, 0) as dec:
        imgs, st, tm = dec.run(paths)
        assert dec.device_entropy_images == len(paths)
    for i in range(len(paths)):
        if i == 4:
            assert (st[i] == 0) == host_accepts
        else:
            assert st[i] == 0 and np.array_equal(imgs[i], want[i]), i


@pytest.mark.parametrize("entropy", ["host", "device"])
def test_batch_decoder_headers_beyond_the_head_read_in_pass_1(jb, tmp_path, monkeypatch, entropy):
    """Pass 1 of the batch decoder reads only the first 64 KB of a file to size its buffers; a file
    whose tables and frame header come later (here: 3 x 60 KB of APPn / COM segments in front of
    them, as a camera file with a thumbnail and a colour profile has), a file that ends inside that
    padding and a progressive file must come out exactly as from the single-image decode of the whole
    file."""
    from conftest import BASELINE_IMAGES, GOLD, load_golden
    if entropy == "host":
        monkeypatch.setenv("JPEGBLK_GPU_HUFFMAN", "0")
    else:
        monkeypatch.delenv("JPEGBLK_GPU_HUFFMAN", raising=False)
    paths, want = [], []
    pad = b"".join(bytes([0xff, m, 0xea, 0x62]) + bytes(60000) for m in (0xe1, 0xe2, 0xfe))  # length 0xea62 = 60002
    for n in BASELINE_IMAGES:
        raw = open(os.path.join(GOLD, "images", n + ".jpg"), "rb").read()
        assert raw[:2] == b"\xff\xd8"
        for tag, data in (("plain", raw), ("padded", raw[:2] + pad + raw[2:])):
            p = str(tmp_path / f"{n}_{tag}.jpg")
            open(p, "wb").write(data)
            paths.append(p)
            want.append(load_golden(n)[3])
    cut = str(tmp_path / "cut.jpg")
    open(cut, "wb").write(open(paths[1], "rb").read()[:70000])  # ends inside the padding
    prog = os.path.join(GOLD, "images", "prograssive-sample-2.jpg")
    extra = [cut] + ([prog] if os.path.exists(prog) else [])
    with jb.Context(0) as one:
        singles = []
        for p in extra:
            try:
                singles.append(one.decode_file(p))
            except jb.JbError as e:
                singles.append(e.status)
    with jb.BatchDecoder(4, 0) as dec:
        imgs, st, tm = dec.run(paths + extra)
    for i, w in enumerate(want):
        assert st[i] == 0 and np.array_equal(imgs[i], w), (i, paths[i], st[i])
    for j, s in enumerate(singles):
        i = len(paths) + j
        if isinstance(s, int):
            assert st[i] == s and imgs[i] is None, (extra[j], st[i], s)
        else:
            assert st[i] == 0 and np.array_equal(imgs[i], s), extra[j]


def test_batch_decoder_soak_short():
    """tools/batch_soak.py for a few seconds: random batches (sizes, samplings, restart intervals short /
    long / none, file-specific Huffman tables, damaged, truncated, progressive and grayscale files mixed
    in) through the batch decoder with the entropy stage on the device -- status for status and pixel
    for pixel the single-image decode with the entropy stage on the host (a 150-second run: 28,604
    images, 13.7 Gpixels, clean: profiles/r02b/batch_soak_150s.txt)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith("JPEGBLK_")}
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "batch_soak.py"), "--seconds", "10", "--seed", "3"],
                       capture_output=True, text=True, timeout=400, env=env)
    assert r.returncode == 0 and "batch soak ok" in r.stdout, (r.stdout + r.stderr)[-2000:]


@pytest.mark.parametrize("entropy", ["host", "device"])
def test_batch_decoder_device_resident_output(jb, oracle, tmp_path, monkeypatch, entropy):
    """jb_batch_decoder_set_device_output: the decoded images stay in the caller's device memory (a
    torch tensor here).  The reference's bundled images, writer files with and without restart
    intervals in groups, one damaged file (flagged by the device decoder, re-decoded through the host
    path into the same region -- or rejected) and a progressive one: every image copied back from the
    device equals the host-output decode; then host output again with the same decoder."""
    import torch
    from conftest import BASELINE_IMAGES, GOLD
    if entropy == "host":
        monkeypatch.setenv("JPEGBLK_GPU_HUFFMAN", "0")
    else:
        monkeypatch.delenv("JPEGBLK_GPU_HUFFMAN", raising=False)
    paths = [os.path.join(GOLD, "images", n + ".jpg") for n in BASELINE_IMAGES]
    more, _ = _write_files(str(tmp_path), "g", 640, 360, 2, 2, 9, 3, 10, oracle)
    plain, _ = _write_files(str(tmp_path), "p", 333, 211, 1, 1, 5, 2, 0, oracle)
    data = bytearray(open(more[4], "rb").read())
    sos = data.index(b"\xff\xda")
    for k in range(300, 340):
        if data[sos + k] != 0xff and data[sos + k - 1] != 0xff:
            data[sos + k] = (data[sos + k] * 7 + 13) % 255
    open(more[4], "wb").write(bytes(data))
    prog = os.path.join(GOLD, "images", "prograssive-sample-2.jpg")
    paths = paths + more + plain + ([prog] if os.path.exists(prog) else [])
    with jb.BatchDecoder(4, 0) as dec:
        want, st_want, _ = dec.run(paths)                       # host output: the expectation
        region = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda:0")
        dec.set_device_output(region.data_ptr(), region.numel())
        for _ in range(2):                                       # the region is recycled by every run
            ptrs, dims, st, tm = dec.run_to_device(paths)
            torch.cuda.synchronize()
            assert st == st_want, (st, st_want)
            for i, p in enumerate(paths):
                if st[i] != 0:
                    assert ptrs[i] == 0
                    continue
                w, h = dims[i]
                off = ptrs[i] - region.data_ptr()
                assert 0 <= off and off + w * h * 3 <= region.numel()
                got = region[off:off + w * h * 3].cpu().numpy().reshape(h, w, 3)
                assert np.array_equal(got, want[i]), p
        with pytest.raises(AssertionError):
            dec.run(paths)                                       # (the Python wrapper refuses to read device pointers as host memory)
        # what is not device memory of the decoder's device is refused when it is set, not when a kernel faults
        host = np.zeros(1 << 20, np.uint8)
        with pytest.raises(jb.JbError):
            dec.set_device_output(host.ctypes.data & ~255, 1 << 16)
        with pytest.raises(jb.JbError):
            dec.set_device_output(region.data_ptr(), region.numel() + (1 << 40))   # reaches beyond the allocation
        dec.set_device_output(region.data_ptr(), region.numel())
        # too small a region: the images that do not fit fail with JB_ERR_CAPACITY, the others are right
        small = torch.zeros(2 << 20, dtype=torch.uint8, device="cuda:0")
        dec.set_device_output(small.data_ptr(), small.numel())
        ptrs, dims, st, tm = dec.run_to_device(paths)
        assert -5 in st and 0 in st
        dec.set_device_output(0, 0)
        again, st2, _ = dec.run(paths)
        assert st2 == st_want and all((a is None and b is None) or np.array_equal(a, b) for a, b in zip(again, want))
    # a multi-device decoder (the one GPU listed twice): one region per listed device, file i in region i % 2
    with jb.BatchDecoder(4, 0, devices=[0, 0]) as multi:
        with pytest.raises(jb.JbError):
            multi.set_device_output(region.data_ptr(), region.numel())      # (that is the single-device form)
        with pytest.raises(jb.JbError):
            multi.set_device_outputs([(region.data_ptr(), region.numel())])  # one region for two devices
        second = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda:0")
        multi.set_device_outputs([(region.data_ptr(), region.numel()), (second.data_ptr(), second.numel())])
        for _ in range(2):
            ptrs, dims, st, tm = multi.run_to_device(paths)
            torch.cuda.synchronize()
            assert st == st_want
            for i, p in enumerate(paths):
                if st[i] != 0:
                    continue
                w, h = dims[i]
                reg = (region, second)[i % 2]
                off = ptrs[i] - reg.data_ptr()
                assert 0 <= off and off + w * h * 3 <= reg.numel(), (i, off)
                assert np.array_equal(reg[off:off + w * h * 3].cpu().numpy().reshape(h, w, 3), want[i]), p
        multi.set_device_outputs([])
        again, st2, _ = multi.run(paths)
        assert st2 == st_want and all((a is None and b is None) or np.array_equal(a, b) for a, b in zip(again, want))


@pytest.mark.parametrize("devices", [None, [0, 0]])
@pytest.mark.parametrize("output", ["malloc", "arena", "device"])
def test_batch_decoder_submit_collect_stream(jb, oracle, tmp_path, monkeypatch, output, devices):
    """jb_batch_decoder_submit / _collect: six batches of different files streamed through one decoder, two in
    flight, every image against the oracle on the blocks written; the state rules (a third submit, run or
    set_arena while batches are in flight, a ticket that does not exist) answer JB_ERR_STATE; then run() again
    on the same decoder (all of a device region belongs to it again)."""
    import torch
    monkeypatch.delenv("JPEGBLK_GPU_HUFFMAN", raising=False)
    pa, wa = _write_files(str(tmp_path), "a", 640, 360, 2, 2, 12, 3, 0, oracle)
    pb, wb = _write_files(str(tmp_path), "b", 333, 211, 1, 1, 10, 2, 5, oracle)
    pc, wc = _write_files(str(tmp_path), "c", 679, 451, 2, 1, 7, 2, 0, oracle)
    batches = [(pa, wa), (pb, wb), (pc + pa[:3], wc + wa[:3]), (pb[:1], wb[:1]), (pa + pb, wa + wb), ([], [])]
    per_batch = max(sum((w.size + 255) // 256 * 256 for w in want) for _, want in batches)
    n_regions = len(devices) if devices else 1
    regions = [torch.zeros(2 * per_batch + 512, dtype=torch.uint8, device="cuda:0") for _ in range(n_regions)]

    def pixels(result, want):
        if output != "device":
            imgs, st, tm = result
            _check(imgs, st, tm, want)
            return
        ptrs, dims, st, tm = result
        torch.cuda.synchronize()
        assert tm["rc"] == 0 and all(s == 0 for s in st), (tm, st)
        for i, w in enumerate(want):
            r = [g for g in regions if g.data_ptr() <= ptrs[i] < g.data_ptr() + g.numel()]
            assert len(r) == 1, i
            off = ptrs[i] - r[0].data_ptr()
            got = r[0][off:off + w.size].cpu().numpy().reshape(w.shape)
            assert dims[i] == (w.shape[1], w.shape[0]) and np.array_equal(got, w), i

    with jb.BatchDecoder(4, 0, arena_bytes=per_batch if output == "arena" else 0, devices=devices) as dec:
        if output == "device":
            if devices:
                dec.set_device_outputs([(g.data_ptr(), g.numel()) for g in regions])
            else:
                dec.set_device_output(regions[0].data_ptr(), regions[0].numel())
        t0 = dec.submit(batches[0][0])
        t1 = dec.submit(batches[1][0])
        with pytest.raises(jb.JbError) as e:
            dec.submit(batches[2][0])            # two in flight already
        assert e.value.status == -7
        assert jb.lib().jb_batch_decoder_set_arena(dec._h, 1 << 20) == -7
        bad = dict(t0)
        bad["id"] = type(t0["id"])(t0["id"].value + 6)
        assert jb.lib().jb_batch_decoder_collect(dec._h, bad["id"], None) == -7
        pixels(dec.collect(t0), batches[0][1])
        flight = [t1]
        for k in range(2, len(batches)):
            flight.append(dec.submit(batches[k][0]))      # batch k starts while batch k-1 runs
            pixels(dec.collect(flight.pop(0)), batches[k - 1][1])
        pixels(dec.collect(flight.pop(0)), batches[-1][1])
        assert jb.lib().jb_batch_decoder_collect(dec._h, t1["id"], None) == -7   # collected already
        assert dec.device_entropy_images > 0
        # the plain run on the same decoder afterwards
        if output == "device":
            pixels(dec.run_to_device(batches[4][0]), batches[4][1])
        else:
            pixels(dec.run(batches[4][0]), batches[4][1])
        # a batch still in flight when the decoder is destroyed finishes first
        dec.submit(batches[0][0])


@pytest.mark.parametrize("entropy", ["host", "device"])
@pytest.mark.parametrize("arena", [False, True])
def test_batch_decoder_later_runs_skip_pass_1_and_set_larger_images_aside(jb, oracle, tmp_path, monkeypatch, arena, entropy):
    """Once a decoder's buffers exist a run has no pass 1: files are parsed as their groups are formed, and an image
    larger than the buffers is decoded in a second, classic round that re-sizes them.  Run 1: small images (sizes the
    decoder); run 2: the same plus larger images of another sampling in between, one of them unreadable; run 3: all
    again (everything fits now); each against the oracle, and JPEGBLK_PASS1=1 gives the same."""
    if entropy == "device":
        monkeypatch.delenv("JPEGBLK_GPU_HUFFMAN", raising=False)
    else:
        monkeypatch.setenv("JPEGBLK_GPU_HUFFMAN", "0")
    small_p, small_w = _write_files(str(tmp_path), "s", 333, 211, 1, 1, 9, 3, 0, oracle)
    big_p, big_w = _write_files(str(tmp_path), "b", 1024, 768, 2, 2, 4, 2, 0, oracle)
    huge_p, huge_w = _write_files(str(tmp_path), "h", 2048, 1536, 1, 1, 1, 1, 16, oracle)
    missing = str(tmp_path / "does_not_exist.jpg")
    mixed_p = small_p[:2] + big_p[:1] + small_p[2:5] + [missing] + huge_p + big_p[1:] + small_p[5:]
    mixed_w = small_w[:2] + big_w[:1] + small_w[2:5] + [None] + huge_w + big_w[1:] + small_w[5:]
    total = sum((w.size + 255) // 256 * 256 for w in mixed_w + small_w if w is not None) + 4096   # the largest run's worth

    def check(result, want):
        imgs, st, tm = result
        for i, w in enumerate(want):
            if w is None:
                assert st[i] == -8 and imgs[i] is None, (i, st[i])
            else:
                assert st[i] == 0 and np.array_equal(imgs[i], w), (i, st[i])

    for pass1 in ("0", "1"):
        monkeypatch.setenv("JPEGBLK_PASS1", pass1)
        with jb.BatchDecoder(4, 0, arena_bytes=total if arena else 0) as dec:
            check(dec.run(small_p), small_w)
            check(dec.run(mixed_p), mixed_w)
            check(dec.run(mixed_p + small_p), mixed_w + small_w)
            if entropy == "device":
                assert dec.device_entropy_images > 0
