#!/usr/bin/env python3
"""Experiment: run the fused kernel directly on pinned HOST memory (coefficients read and pixels
written over the link by the kernel itself, no staging copies) and compare with the staged
pipeline (jb_submit ring).  python tools/zero_copy_probe.py [--size 8192x8192] [--sub 420] [--n 32]"""
import argparse, ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jpeg_decoder_amd as jb
from jpeg_decoder_amd import synth
from jpeg_decoder_amd.api import DeviceBatch


def pinned(nbytes, dtype):
    p = jb.lib().jb_pinned_alloc(nbytes)
    return p, np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(nbytes,)).view(dtype)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="8192x8192")
    ap.add_argument("--sub", default="420")
    ap.add_argument("--n", type=int, default=32)
    a = ap.parse_args()
    import torch
    w, h = (int(v) for v in a.size.split("x"))
    hs, vs = {"444": (1, 1), "420": (2, 2)}[a.sub]
    coef, q = synth.synth_blocks(w, h, hs, vs, 1)
    desc = jb.make_desc(w, h, hs, vs)
    g = jb.geometry_of(desc)
    dev = torch.device("cuda:0")
    slots = 3
    bufs = []
    for _ in range(slots):
        pc, ac = pinned(g.coef_bytes, np.int16)
        ac[:] = coef.reshape(-1)
        pr, ar = pinned(g.rgb_bytes, np.uint8)
        bufs.append((pc, ac, pr, ar))
    q_t = torch.from_numpy(jb.resolve_qtabs(desc, q)).to(dev)
    streams = [torch.cuda.Stream(dev) for _ in range(2)]
    with jb.Context(0, g.coef_bytes, g.rgb_bytes, slots) as ctx:
        # staged reference result
        want = np.zeros(g.rgb_bytes, np.uint8)
        ctx.wait(ctx.submit(desc, bufs[0][1], q, want))
        for nstreams in (1, 2):
            for rep in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(a.n):
                    pc, _, pr, _ = bufs[i % slots]
                    b = DeviceBatch()
                    b.desc = desc
                    b.n_images = 1
                    b.d_coef = pc
                    b.coef_image_stride = g.coef_bytes
                    b.d_qtabs = q_t.data_ptr()
                    b.qtab_image_stride = 0
                    b.d_rgb = pr
                    b.rgb_row_stride = 3 * w
                    b.rgb_image_stride = g.rgb_bytes
                    ctx.blocks_to_rgb_device(b, streams[i % nstreams].cuda_stream)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            ok = np.array_equal(bufs[(a.n - 1) % slots][3], want)
            print(f"zero-copy, {nstreams} stream(s): {a.n / dt:9.1f} images/s  {a.n * w * h / dt / 1e9:6.2f} Gpixel/s  "
                  f"{a.n * (g.coef_bytes + g.rgb_bytes) / dt / 1e9:6.1f} GB/s over the link  exact={ok}")
        for rep in range(2):
            t0 = time.perf_counter()
            tickets = []
            for i in range(a.n):
                _, ac, _, ar = bufs[i % slots]
                tickets.append(ctx.submit(desc, ac, q, ar))
                if len(tickets) >= slots:
                    ctx.wait(tickets.pop(0))
            while tickets:
                ctx.wait(tickets.pop(0))
            dt = time.perf_counter() - t0
        print(f"staged ring (copy engines):  {a.n / dt:9.1f} images/s  {a.n * w * h / dt / 1e9:6.2f} Gpixel/s  "
              f"{a.n * (g.coef_bytes + g.rgb_bytes) / dt / 1e9:6.1f} GB/s over the link")


if __name__ == "__main__":
    main()
