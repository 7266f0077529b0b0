#!/usr/bin/env python3
"""Does a batch that follows an idle gap run slower (clock / link wake-up), or is the slow first run of a fresh decoder
its ring and pinned buffers growing to the batch's group size?  128 x 1080p files: runs back to back, after sleeps,
and the first run of a second decoder created with the first batch's sizes already known."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import jpeg_decoder_amd as jb
from jpeg_decoder_amd import synth
w, h = 1920, 1080
coef, q = synth.synth_blocks(w, h, 1, 1, image_index=7)
paths = []
for i in range(8):
    p = f"/tmp/idle_{i}.jpg"
    open(p, "wb").write(synth.encode_jpeg(np.roll(coef, i * 4099 * 3, axis=0), w, h, 1, 1, q))
    paths.append(p)
g = jb.geometry_of(jb.make_desc(w, h, 1, 1))
per = (g.rgb_bytes + 255) // 256 * 256
files = [paths[i % 8] for i in range(128)]
for label in ("decoder A", "decoder B"):
    with jb.BatchDecoder(16, 0, g.coef_bytes, g.rgb_bytes, arena_bytes=128 * per) as dec:
        for what, gap in (("first run", 0), ("back to back", 0), ("back to back", 0), ("after 0.2 s idle", 0.2), ("back to back", 0), ("after 1 s idle", 1.0), ("back to back", 0), ("after 4 s idle", 4.0), ("back to back", 0)):
            time.sleep(gap)
            _, st, tm = dec.run(files, keep_pixels=False)
            print(f"{label}: {what:18s} wall {tm['wall_s'] * 1e3:7.1f} ms  read {tm['read_s'] * 1e3:6.1f}  entropy(prepare) {tm['entropy_s'] * 1e3:6.1f}  submit+wait {tm['device_s'] * 1e3:7.1f}", flush=True)
