#!/usr/bin/env python3
"""Why is a 256-file batch of 8192x8192 images, decoded as consecutive runs over one arena, slower per image than
one run of 32?  Times consecutive runs of N files (N = 16, 32, 42) and prints the library's own breakdown."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import jpeg_decoder_amd as jb
from jpeg_decoder_amd import synth
w = h = 8192
coef, q = synth.synth_blocks(w, h, 2, 2, image_index=7)
paths = []
for i in range(4):
    p = f"/tmp/probe_{i}.jpg"
    open(p, "wb").write(synth.encode_jpeg(np.roll(coef, i * 4099 * 6, axis=0), w, h, 2, 2, q))
    paths.append(p)
del coef
g = jb.geometry_of(jb.make_desc(w, h, 2, 2))
per = (g.rgb_bytes + 255) // 256 * 256
for n in (16, 32, 42):
    files = [paths[i % 4] for i in range(n)]
    with jb.BatchDecoder(16, 0, g.coef_bytes, g.rgb_bytes, arena_bytes=n * per) as dec:
        dec.run(files[:16], keep_pixels=False)
        for k in range(4):
            t0 = time.time()
            _, st, tm = dec.run(files, keep_pixels=False)
            print(n, "files run", k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in tm.items()}, "python wall", round(time.time() - t0, 4), "->", round(n / tm["wall_s"], 1), "images/s", flush=True)
