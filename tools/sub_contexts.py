#!/usr/bin/env python3
"""One batch, one device, the host threads dealt over k contexts (a multi-device decoder with the same
device listed k times): do several rings decouple the host threads?  Host output into the pinned arena,
entropy stage on the device.  Test infrastructure."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import jpeg_decoder_amd as jb  # noqa: E402
from e2e_bench import make_jpegs, make_jpegs_writer  # noqa: E402


def main():
    for (w, h, sub, n, distinct_n, maker) in ((1920, 1080, "444", 1024, 8, make_jpegs), (8192, 8192, "420", 64, 4, make_jpegs_writer),
                                              (679, 451, "420", 8192, 16, make_jpegs_writer)):
        with tempfile.TemporaryDirectory(dir="/tmp") as d:
            distinct = maker(distinct_n, w, h, sub, d, 0)
            paths = [distinct[i % len(distinct)] for i in range(n)]
            g = jb.geometry_of(jb.entropy_decode(open(distinct[0], "rb").read(), headers_only=True)[0])
            per = (g.rgb_bytes + 255) // 256 * 256
            for k in (1, 2, 4):
                kw = {"devices": [0] * k} if k > 1 else {}
                with jb.BatchDecoder(16, 0, g.coef_bytes, g.rgb_bytes, arena_bytes=n * per, **kw) as dec:
                    walls = []
                    for rep in range(4):
                        _, st, tm = dec.run(paths, keep_pixels=False)
                        assert all(s == 0 for s in st)
                        walls.append(tm["wall_s"])
                print(f"{w}x{h} {sub} x{n}: {k} context(s): {n / min(walls[1:]):9.1f} images/s  walls {[round(x, 3) for x in walls]}", flush=True)


if __name__ == "__main__":
    main()
