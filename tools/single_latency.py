#!/usr/bin/env python3
"""Latency of ONE decode(bytes) -- the reference's own use (one image per process,
jpeg.cpp:916-929) -- with the entropy stage on one host core (JPEGBLK_GPU_HUFFMAN=0), on the host
threads over restart intervals where the file has them, and on the device (=1), per image size.
Decides the single-image default of jb_decode_memory.  Every decode is compared with the host
path's pixels.  PIL files (quality 90, Annex-K tables) when PIL is importable, else the build's
own writer."""
import io
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jpeg_decoder_amd as jb  # noqa: E402
from jpeg_decoder_amd import synth  # noqa: E402


def photo(w, h, seed=1):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([(xx * 2 + yy) % 256, (yy * 3 + xx) % 256, (xx + yy * 2) // 3 % 256], -1)
    noise = rng.normal(0, 12, (h // 8 + 1, w // 8 + 1, 3)).repeat(8, 0).repeat(8, 1)[:h, :w]
    return np.clip(base * 0.6 + 60 + noise + rng.normal(0, 3, (h, w, 3)).astype(np.float32), 0, 255).astype(np.uint8)


def make(w, h, sub, dri_rows):
    try:
        from PIL import Image
        b = io.BytesIO()
        kw = {"restart_marker_rows": dri_rows} if dri_rows else {}
        Image.fromarray(photo(w, h)).save(b, "JPEG", quality=90, subsampling={"444": 0, "420": 2}[sub], optimize=False, **kw)
        return "pil", b.getvalue()
    except ImportError:
        hs, vs = (1, 1) if sub == "444" else (2, 2)
        coef, q = synth.synth_blocks(w, h, hs, vs, 1)
        mx = (w + 8 * hs - 1) // (8 * hs)
        return "writer", synth.encode_jpeg(coef, w, h, hs, vs, q, restart_interval=mx * dri_rows)


def timed(ctx, data, reps):
    """Best time of the C call alone (jb_decode_memory: the pixels arrive in a buffer the library
    allocates); the copy into a numpy array for the comparison is outside the clock."""
    import ctypes
    lib = jb.lib()
    buf = np.frombuffer(data, dtype=np.uint8)
    best, px = 1e9, None
    for r in range(reps + 2):
        p, w, h = ctypes.c_void_p(), ctypes.c_int32(), ctypes.c_int32()
        t0 = time.perf_counter()
        rc = lib.jb_decode_memory(ctx._h, buf.ctypes.data_as(ctypes.c_void_p), buf.size, ctypes.byref(p), ctypes.byref(w), ctypes.byref(h))
        dt = time.perf_counter() - t0
        assert rc == 0, lib.jb_last_error(ctx._h)
        if r == 0:
            px = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(w.value * h.value * 3,)).copy().reshape(h.value, w.value, 3)
        lib.jb_free(p)
        if r >= 2:
            best = min(best, dt)
    return best, px


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="", help="WxH: just this size (for a kernel trace)")
    ap.add_argument("--sub", default="", help="444 | 420: just this sampling")
    ap.add_argument("--dri", default="0,1", help="restart interval in MCU rows, comma separated (0 = none)")
    args = ap.parse_args()
    sizes = [(679, 451, "420"), (1024, 768, "420"), (1280, 720, "420"), (1280, 720, "444"), (1920, 1080, "444"), (1920, 1080, "420"), (4096, 4096, "420"), (4096, 4096, "444"), (8192, 8192, "420")]
    if args.only:
        sizes = [x for x in sizes if f"{x[0]}x{x[1]}" == args.only]
    if args.sub:
        sizes = [x for x in sizes if x[2] == args.sub]
    rows = []
    # (a context reads its knobs when it is created: one context per mode)
    os.environ["JPEGBLK_GPU_HUFFMAN"] = "0"
    ctx_host = jb.Context(0)
    os.environ["JPEGBLK_GPU_HUFFMAN"] = "2"
    ctx = jb.Context(0)
    os.environ.pop("JPEGBLK_GPU_HUFFMAN", None)
    with ctx_host, ctx:
        for w, h, sub in sizes:
            for dri in [int(v) for v in args.dri.split(",")]:
                src, data = make(w, h, sub, dri)
                reps = 8 if w * h < 3e7 else 4
                t_host, ref = timed(ctx_host, data, reps)
                before = ctx.device_entropy_images
                t_dev, px = timed(ctx, data, reps)
                took = ctx.device_entropy_images - before
                same = bool(np.array_equal(ref, px))
                row = {"size": f"{w}x{h}", "sub": sub, "dri_rows": dri, "source": src, "bytes": len(data),
                       "host_ms": round(t_host * 1e3, 3), "device_ms": round(t_dev * 1e3, 3),
                       "device_path_taken": took > 0, "pixels_equal": same}
                rows.append(row)
                print(json.dumps(row), flush=True)
                assert same
    print(json.dumps({"what": "one decode(bytes) end to end (parse, entropy stage, upload, kernel, download into malloc'ed pixels), best of N",
                      "rows": rows}))


if __name__ == "__main__":
    main()
