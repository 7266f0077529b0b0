#!/usr/bin/env python3
"""How many synchronisation passes does the self-synchronising device decoder need?  For a set of
files without restart intervals: the smallest JPEGBLK_SYNC_ROUNDS with which the asynchronous path
(decode(bytes), no retry) takes the device path successfully, and the time of one image through
jb_entropy_decode_device.  Run each round count in a child process (the knob is read once)."""
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
    res = {}
    with jb.Context(0) as ctx:
        for name, data in files().items():
            ok = True
            try:
                best = 1e9
                for _ in range(4):
                    t0 = time.perf_counter()
                    got = ctx.decode_memory(data)
                    best = min(best, time.perf_counter() - t0)
                ok = ctx.device_entropy_images == 4 * (len(res) + 1)
            except jb.JbError:
                ok = False
            n_dev = ctx.device_entropy_images
            res[name] = (n_dev, best)
    prev = 0
    for name, (n_dev, best) in res.items():
        print(f"rounds={rounds:3d} {name:24s} device-path decodes {n_dev - prev}/4  decode(bytes) {best * 1e3:7.2f} ms")
        prev = n_dev


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(int(sys.argv[1]))
    else:
        for r in (1, 2, 3, 4, 8):
            env = dict(os.environ, JPEGBLK_SYNC_ROUNDS=str(r), JPEGBLK_GPU_HUFFMAN="1")
            subprocess.run([sys.executable, os.path.abspath(__file__), str(r)], env=env)
