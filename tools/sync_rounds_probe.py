#!/usr/bin/env python3
"""How many synchronisation passes does the self-synchronising device decoder need?  For a set of
files without restart intervals: the time of one image through jb_entropy_decode_device when the
first attempt uses JPEGBLK_SYNC_ROUNDS = 1, 2, 3, 4, 8 passes (a first attempt that is not in step is
repeated with 64 passes, which shows as a jump in time).  One child process per setting (the knob is
read once)."""
import io
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def files():
    from jpeg_decoder_amd import synth
    out = {}
    for name, (w, h, hs, vs) in {"writer 1080p 444": (1920, 1080, 1, 1), "writer 4096 420": (4096, 4096, 2, 2), "writer 679x451 420": (679, 451, 2, 2)}.items():
        coef, q = synth.synth_blocks(w, h, hs, vs, 1)
        out[name] = synth.encode_jpeg(coef, w, h, hs, vs, q)
    try:
        from PIL import Image
        rng = np.random.default_rng(1)
        w, h = 1920, 1080
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([(xx * 2 + yy) % 256, (yy * 3 + xx) % 256, (xx + yy * 2) // 3 % 256], -1)
        noise = rng.normal(0, 12, (h // 8 + 1, w // 8 + 1, 3)).repeat(8, 0).repeat(8, 1)[:h, :w]
        img = np.clip(base * 0.6 + 60 + noise + rng.normal(0, 3, (h, w, 3)), 0, 255).astype(np.uint8)
        for q in (50, 90, 98):
            for sub in (0, 2):
                b = io.BytesIO()
                Image.fromarray(img).save(b, "JPEG", quality=q, subsampling=sub)
                out[f"PIL q{q} 1080p sub{sub}"] = b.getvalue()
    except ImportError:
        pass
    return out


def child(rounds):
    import jpeg_decoder_amd as jb
    with jb.Context(0) as ctx:
        for name, data in files().items():
            want = jb.entropy_decode(data)[2]
            best = 1e9
            for _ in range(4):
                t0 = time.perf_counter()
                got = ctx.entropy_decode_device(data)[2]
                best = min(best, time.perf_counter() - t0)
            assert np.array_equal(got, want), name
            t0 = time.perf_counter()
            jb.entropy_decode(data)
            host = time.perf_counter() - t0
            print(f"first attempt with {rounds:2d} passes: {name:24s} jb_entropy_decode_device {best * 1e3:7.2f} ms (retries with 64 passes when not in step)  host {host * 1e3:6.2f} ms", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(int(sys.argv[1]))
    else:
        for r in (1, 2, 3, 4, 8):
            env = dict(os.environ, JPEGBLK_SYNC_ROUNDS=str(r), JPEGBLK_GPU_HUFFMAN="1")
            subprocess.run([sys.executable, os.path.abspath(__file__), str(r)], env=env)
