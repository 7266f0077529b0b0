#!/usr/bin/env python3
"""Does overlapping batches close the gaps of one batch?  1,024 1080p files through ONE batch decoder
with 16 host threads, against TWO decoders with 8 threads each that decode 512 files each at the same
time from two caller threads (pinned arenas; the calls release the GIL).  Host output, entropy stage
on the device (the default).  Test infrastructure; nothing here is on the product path."""
import os
import sys
import tempfile
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import jpeg_decoder_amd as jb  # noqa: E402
from e2e_bench import make_jpegs  # noqa: E402


def main():
    n, w, h = 1024, 1920, 1080
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        distinct = make_jpegs(8, w, h, "444", d, 0)
        paths = [distinct[i % 8] for i in range(n)]
        g = jb.geometry_of(jb.entropy_decode(open(distinct[0], "rb").read(), headers_only=True)[0])
        per = (g.rgb_bytes + 255) // 256 * 256
        for label, parts, threads in (("one decoder, 16 threads, 1,024 files", 1, 16), ("two decoders, 8 threads each, 512 files each, at once", 2, 8),
                                      ("four decoders, 4 threads each, 256 files each, at once", 4, 4)):
            decs = [jb.BatchDecoder(threads, 0, g.coef_bytes, g.rgb_bytes, arena_bytes=(n // parts) * per) for _ in range(parts)]
            shares = [paths[k::parts] for k in range(parts)]
            best = 1e9
            for rep in range(4):
                res = [None] * parts

                def work(k):
                    res[k] = decs[k].run(shares[k], keep_pixels=False)

                th = [threading.Thread(target=work, args=(k,)) for k in range(parts)]
                t0 = time.perf_counter()
                for t in th:
                    t.start()
                for t in th:
                    t.join()
                dt = time.perf_counter() - t0
                assert all(all(s == 0 for s in r[1]) for r in res)
                if rep:
                    best = min(best, dt)
            for x in decs:
                x.close()
            print(f"{label:58s} {n / best:8.0f} images/s  ({best * 1e3:.0f} ms)", flush=True)


if __name__ == "__main__":
    main()
