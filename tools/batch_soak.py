#!/usr/bin/env python3
"""Randomised soak of the batch decoder with the entropy stage on the device (its default), GPU box:
random batches of baseline files -- sizes 8..2600 each way, all four samplings, no restart
intervals / short ones / long ones (so all three device decoders and the host fallback take part),
Annex-K and file-specific Huffman tables (PIL optimize), a few damaged, truncated, progressive and
grayscale files mixed in -- through jb_batch_decoder with 1..16 host threads, malloc'ed outputs or
the pinned arena.  Every image must come out exactly as from the single-image decode with the
entropy stage on the HOST (JPEGBLK_GPU_HUFFMAN=0), status for status, pixel for pixel.
  python tools/batch_soak.py [--seconds 120] [--seed 1]
Test infrastructure; nothing here is on the product path."""
import argparse
import io
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jpeg_decoder_amd as jb  # noqa: E402
from jpeg_decoder_amd import synth  # noqa: E402


def one_file(rng):
    hs, vs = [(1, 1), (2, 1), (1, 2), (2, 2)][int(rng.integers(4))]
    w = int(rng.integers(8, 2600)) if rng.random() < 0.7 else int(rng.choice([16, 64, 679, 1920]))
    h = int(rng.integers(8, 900)) if rng.random() < 0.8 else int(rng.choice([8, 451, 1080]))
    kind = rng.random()
    mx = (w + 8 * hs - 1) // (8 * hs)
    ri = int(rng.choice([0, 0, 1, 3, mx, 2 * mx + 1, 1000000]))
    if kind < 0.65:
        coef, q = synth.synth_blocks(w, h, hs, vs, int(rng.integers(1 << 30)))
        return synth.encode_jpeg(coef, w, h, hs, vs, q, restart_interval=min(ri, 65535))
    try:
        from PIL import Image
    except ImportError:
        coef, q = synth.synth_blocks(w, h, hs, vs, int(rng.integers(1 << 30)))
        return synth.encode_jpeg(coef, w, h, hs, vs, q)
    img = np.clip(np.cumsum(rng.normal(0, 6, (h, w, 3)), axis=1) + 128 + rng.normal(0, 4, (h, w, 3)), 0, 255).astype(np.uint8)
    b = io.BytesIO()
    kw = {}
    if kind < 0.85:
        if ri:
            kw["restart_marker_blocks" if rng.random() < 0.5 else "restart_marker_rows"] = int(rng.integers(1, 9))
        Image.fromarray(img).save(b, "JPEG", quality=int(rng.integers(30, 99)), subsampling=int(rng.integers(0, 3)), optimize=bool(rng.integers(2)), **kw)
    elif kind < 0.93:
        Image.fromarray(img).save(b, "JPEG", quality=80, progressive=True)
    else:
        Image.fromarray(img[:, :, 0]).save(b, "JPEG", quality=80)
    return b.getvalue()


def damage(rng, data):
    d = bytearray(data)
    r = rng.random()
    if r < 0.5 and len(d) > 700:
        for _ in range(int(rng.integers(1, 6))):      # garbage inside the scan
            k = int(rng.integers(len(d) // 2, len(d) - 2))
            d[k] = int(rng.integers(0, 255))
    elif r < 0.8:
        d = d[: int(rng.integers(2, len(d)))]          # truncated
    else:
        k = int(rng.integers(2, min(len(d), 600)))     # a damaged header
        d[k] ^= 1 << int(rng.integers(8))
    return bytes(d)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    n_batches = n_images = n_bad = n_dev = n_streamed = n_veteran = 0
    veteran = None
    pixels = 0
    # the reference context keeps the entropy stage on the host (a context reads its knobs when it is created)
    os.environ["JPEGBLK_GPU_HUFFMAN"] = "0"
    one_ctx = jb.Context(0)
    os.environ.pop("JPEGBLK_GPU_HUFFMAN", None)
    with tempfile.TemporaryDirectory(dir="/tmp") as d, one_ctx as one:
        while time.time() - t0 < args.seconds:
            distinct = []
            for k in range(int(rng.integers(1, 7))):
                data = one_file(rng)
                if rng.random() < 0.12:
                    data = damage(rng, data)
                p = os.path.join(d, f"f{k}.jpg")
                with open(p, "wb") as f:
                    f.write(data)
                # what it must decode to: the single-image decode, entropy stage on the host
                try:
                    want = one.decode_memory(data)
                except jb.JbError as e:
                    want = e.status
                # the arena space the batch decoder may take for it (also for an image whose scan turns out corrupt)
                try:
                    dsc = jb.entropy_decode(data, headers_only=True)[0]
                    room = (dsc.width * dsc.height * 3 + 255) // 256 * 256
                except jb.JbError:
                    room = 0
                distinct.append((p, want, room))
            n = int(rng.integers(1, 80))
            order = [int(rng.integers(len(distinct))) for _ in range(n)]
            if rng.random() < 0.5:
                order.sort()                                # runs of one geometry: large device groups
            paths = [distinct[i][0] for i in order]
            threads = int(rng.choice([1, 2, 5, 16]))
            arena = 0
            if rng.random() < 0.5:
                arena = sum(distinct[i][2] for i in order) + 4096
            # one batch in three goes to a decoder that lives on across batches (malloc'ed outputs, 5 threads): its runs
            # have no pass 1, and whatever is larger than the buffers it has grown so far takes the second round
            if arena == 0 and rng.random() < 0.33:
                if veteran is None:
                    veteran = jb.BatchDecoder(5, 0)
                imgs, st, tm = veteran.run(paths)
                n_veteran += 1
                dec = None
            else:
                dec = jb.BatchDecoder(threads, 0, arena_bytes=arena)
            if dec is not None:
                with dec:
                    mode = rng.random()
                    if mode < 0.6:
                        imgs, st, tm = dec.run(paths)
                    else:
                        # the same batch through submit / collect: as two halves in flight at once (each side's arena holds
                        # the whole batch's worth), sometimes with a plain run on the same decoder in front
                        if mode < 0.7:
                            dec.run(paths[:3], keep_pixels=False)
                        cut = int(rng.integers(0, n + 1))
                        ta = dec.submit(paths[:cut])
                        tb = dec.submit(paths[cut:])
                        ia, sa, _ = dec.collect(ta)
                        ib, sb, _ = dec.collect(tb)
                        imgs, st = ia + ib, sa + sb
                        n_streamed += 1
                    n_dev += dec.device_entropy_images
            for j, i in enumerate(order):
                want = distinct[i][1]
                if isinstance(want, int):
                    assert st[j] == want and imgs[j] is None, (paths[j], st[j], want)
                    n_bad += 1
                else:
                    assert st[j] == 0 and np.array_equal(imgs[j], want), (paths[j], st[j], threads, bool(arena))
                    pixels += want.size // 3
            n_batches += 1
            n_images += n
    if veteran is not None:
        n_dev += veteran.device_entropy_images
        veteran.close()
    print(f"batch soak ok: {n_batches} batches ({n_streamed} of them as two halves through submit / collect, {n_veteran} on one long-lived decoder), {n_images} images ({n_bad} rejected as by the single-image host decode, "
          f"{n_dev} entropy-decoded on the device), {pixels / 1e9:.2f} Gpixels compared, {time.time() - t0:.0f} s, seed {args.seed}")


if __name__ == "__main__":
    main()
