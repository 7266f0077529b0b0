#!/usr/bin/env python3
"""Mutation fuzzer for the device-side entropy decoder (GPU box): damaged scans, damaged Huffman
tables, truncated files, wrong restart intervals -> jb_entropy_decode_device must answer with a
jb_status (never hang, never fault), and whenever it AND the host decoder accept a damaged stream
they must agree on every coefficient.  Usage: python tools/huff_fuzz.py --seconds 60 [--seed 1]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jpeg_decoder_amd as jb  # noqa: E402
from jpeg_decoder_amd import synth  # noqa: E402


def seeds():
    out = []
    for i, (w, h, hs, vs, ri) in enumerate([(200, 120, 2, 2, 3), (333, 211, 1, 1, 7), (160, 96, 2, 1, 1), (97, 131, 1, 2, 5), (640, 360, 2, 2, 40)]):
        coef, q = synth.synth_blocks(w, h, hs, vs, 60 + i)
        out.append(bytearray(synth.encode_jpeg(coef, w, h, hs, vs, q, restart_interval=ri)))
        if i % 2 == 0:  # the same without restart intervals: the self-synchronising decoder
            out.append(bytearray(synth.encode_jpeg(coef, w, h, hs, vs, q)))
    try:
        import io
        from PIL import Image
        rng = np.random.default_rng(3)
        img = np.clip(np.cumsum(rng.normal(0, 5, (151, 227, 3)), axis=1) + 128, 0, 255).astype(np.uint8)
        for kw in ({"restart_marker_blocks": 4}, {"restart_marker_rows": 1, "optimize": True}, {}, {"optimize": True}):
            b = io.BytesIO()
            Image.fromarray(img).save(b, "JPEG", quality=88, subsampling=2, **kw)
            out.append(bytearray(b.getvalue()))
    except ImportError:
        pass
    return out


def mutate(rng, d):
    d = bytearray(d)
    sos = d.index(b"\xff\xda")
    kind = rng.integers(0, 8)
    n = len(d)
    if kind <= 3:  # bytes inside the scan
        for _ in range(int(rng.integers(1, 6))):
            at = int(rng.integers(sos + 14, n - 2))
            d[at] = int(rng.integers(0, 256)) if kind == 3 else int(rng.integers(0, 255))
    elif kind == 4:  # truncate
        d = d[:int(rng.integers(sos + 14, n))]
    elif kind == 5:  # a byte of a Huffman table
        dht = d.index(b"\xff\xc4")
        at = int(rng.integers(dht + 4, sos))
        d[at] = int(rng.integers(0, 256))
    elif kind == 6:  # the restart interval
        dri = d.find(b"\xff\xdd")
        if dri > 0:
            d[dri + 5] = int(rng.integers(0, 256))
    else:  # delete a slice of the scan
        at = int(rng.integers(sos + 14, n - 20))
        del d[at:at + int(rng.integers(1, 16))]
    return bytes(d)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=30)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    base = seeds()
    counts, agree, lenient, t0 = {}, 0, 0, time.time()
    # both chunk sizes of the device decoder (a context reads JPEGBLK_CHUNK_BYTES when it is created)
    os.environ["JPEGBLK_CHUNK_BYTES"] = "64"
    ctx64 = jb.Context(0)
    os.environ["JPEGBLK_CHUNK_BYTES"] = "128"
    ctx128 = jb.Context(0)
    os.environ.pop("JPEGBLK_CHUNK_BYTES", None)
    with ctx64, ctx128:
        for s in base:  # the unmutated seeds decode and agree
            for ctx in (ctx64, ctx128):
                _, _, cd = ctx.entropy_decode_device(bytes(s))
                assert np.array_equal(cd, jb.entropy_decode(bytes(s))[2])
        while time.time() - t0 < args.seconds:
            ctx = (ctx64, ctx128)[int(rng.integers(0, 2))]
            data = mutate(rng, base[int(rng.integers(0, len(base)))])
            try:
                _, _, cd = ctx.entropy_decode_device(data)
                st = 0
            except jb.JbError as e:
                st, cd = e.status, None
            assert -9 <= st <= 0, st
            counts[st] = counts.get(st, 0) + 1
            if st == 0:
                try:
                    ch = jb.entropy_decode(data)[2]
                except jb.JbError:
                    ch = None
                if ch is not None:
                    assert np.array_equal(cd, ch), "device and host decoders disagree on a stream both accept"
                    agree += 1
                else:
                    # the host decoder is the authority: the device must not accept what it rejects
                    lenient += 1
                    if lenient <= 3:
                        out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", f"huff_fuzz_lenient_{lenient}.jpg")
                        try:
                            open(out, "wb").write(data)
                        except OSError:
                            pass
    total = sum(counts.values())
    print(f"huff fuzz ok (both chunk sizes of the device decoder at random): {total} mutants, statuses {dict(sorted(counts.items()))}, {agree} accepted by both decoders and equal, {lenient} accepted by the device decoder alone")
    assert lenient == 0, "the device decoder accepted streams the host decoder rejects (saved under gpurun_out/)"


if __name__ == "__main__":
    main()
