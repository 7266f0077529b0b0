#!/usr/bin/env python3
"""Per-block-round latency of the device entropy decoder (jb_huff.hip), in isolation: ONE image
(a 1080p 4:4:4 file with one restart interval per MCU row = 135 lanes, 720 blocks each) through
jb_entropy_decode_device, timed end to end (upload of the compressed scan + memset + kernel), so
time / 720 is what one wave needs for one block of each of its lanes when nothing contends.
JPEGBLK_LIB selects a build variant (tools/build_huff_variant.sh)."""
import io
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jpeg_decoder_amd as jb  # noqa: E402
from jpeg_decoder_amd import synth  # noqa: E402


def files():
    w, h = 1920, 1080
    coef, q = synth.synth_blocks(w, h, 1, 1, 1)
    yield "writer 1080p 4:4:4, DRI = 1 row", synth.encode_jpeg(coef, w, h, 1, 1, q, restart_interval=240), 720
    try:
        from PIL import Image
        rng = np.random.default_rng(1)
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([(xx * 2 + yy) % 256, (yy * 3 + xx) % 256, (xx + yy * 2) // 3 % 256], -1)
        noise = rng.normal(0, 12, (h // 8 + 1, w // 8 + 1, 3)).repeat(8, 0).repeat(8, 1)[:h, :w]
        img = np.clip(base * 0.6 + 60 + noise + rng.normal(0, 3, (h, w, 3)), 0, 255).astype(np.uint8)
        b = io.BytesIO()
        Image.fromarray(img).save(b, "JPEG", quality=90, subsampling=0, optimize=False, restart_marker_rows=1)
        yield "PIL q90 1080p 4:4:4, DRI = 1 row", b.getvalue(), 720
        b = io.BytesIO()
        Image.fromarray(img).save(b, "JPEG", quality=90, subsampling=0, optimize=False, restart_marker_blocks=8)
        yield "PIL q90 1080p 4:4:4, DRI = 8 MCUs", b.getvalue(), 24
    except ImportError:
        pass


def main():
    import ctypes
    import torch
    with jb.Context(0) as ctx:
        for name, data, rounds in files():
            buf = np.frombuffer(data, dtype=np.uint8)
            desc, q, _ = jb.entropy_decode(data, headers_only=True)
            g = jb.geometry_of(desc)
            t = torch.zeros((g.n_coded_blocks, 64), dtype=torch.int16, device="cuda:0")
            d2, q2 = jb.ImageDesc(), np.zeros((4, 64), np.uint16)
            args = (ctx._h, buf.ctypes.data_as(ctypes.c_void_p), buf.size, ctypes.byref(d2), q2.ctypes.data_as(ctypes.c_void_p), t.data_ptr(), t.numel() * 2)
            best = 1e9
            for rep in range(12):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                rc = jb.lib().jb_entropy_decode_device(*args)
                dt = time.perf_counter() - t0
                assert rc == 0, jb.lib().jb_last_error(ctx._h)
                if rep >= 2:
                    best = min(best, dt)
            t0 = time.perf_counter()
            for _ in range(5):
                jb.entropy_decode(data)
            host = (time.perf_counter() - t0) / 5
            print(f"{os.path.basename(jb.lib_path()):28s} {name:36s} device {best * 1e3:7.2f} ms = {best * 1e6 / rounds:6.2f} us per block round | host 1 thread {host * 1e3:6.2f} ms", flush=True)


if __name__ == "__main__":
    main()
