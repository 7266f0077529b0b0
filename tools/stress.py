#!/usr/bin/env python3
"""Randomised soak of the device seam against the oracle (test infrastructure, GPU box):
random sizes (1..2200 each way, biased to tile edges), all four layouts, random table ids and
tables, sparse / dense / full-range coefficients, batches of 1-4 images, random row strides and
byte offsets of the output.  Every launch is compared byte for byte with oracle/ and the bytes
around the image must stay untouched.
  python tools/stress.py [--seconds 120] [--seed 1]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import torch
    import jpeg_decoder_amd as jb
    from jpeg_decoder_amd import synth
    from jpeg_decoder_amd.api import torch_batch
    from oracle.pyoracle import Oracle, make_desc as odesc
    ora = Oracle()
    rng = np.random.default_rng(args.seed)
    dev = torch.device("cuda:0")
    ts = torch.cuda.Stream(dev)
    edges = [1, 7, 8, 9, 15, 16, 17, 255, 256, 257, 511, 512, 513, 767, 768, 1023, 1024, 1025, 1535, 1536, 1537, 1920, 2047, 2048]
    t0, n, pixels = time.time(), 0, 0
    # two contexts (knobs are read when a context is made): the default -- 4:4:4 launches of this size take the
    # one-wave-per-16-MCUs kernel -- and one that keeps every launch on the 192-lane kernel
    os.environ["JPEGBLK_SMALL_GRID"] = "0"
    big_only = jb.Context(0)
    os.environ.pop("JPEGBLK_SMALL_GRID")
    with torch.cuda.stream(ts), jb.Context(0) as default_ctx, big_only:
        while time.time() - t0 < args.seconds:
            ctx = default_ctx if rng.random() < 0.5 else big_only
            hs, vs = [(1, 1), (2, 1), (1, 2), (2, 2)][rng.integers(4)]
            w = int(rng.choice(edges) + rng.integers(-3, 4)) if rng.random() < 0.5 else int(rng.integers(1, 2200))
            h = int(rng.choice(edges[:16]) + rng.integers(-3, 4)) if rng.random() < 0.3 else int(rng.integers(1, 700))
            w, h = max(w, 1), max(h, 1)
            qid = tuple(int(v) for v in rng.integers(0, 4, 3))
            q = rng.integers(1, 256, (4, 64)).astype(np.uint16)
            nimg = int(rng.integers(1, 5))
            desc = jb.make_desc(w, h, hs, vs, qid)
            g = jb.geometry_of(desc)
            kind = rng.integers(3)
            coefs = []
            for i in range(nimg):
                if kind == 0:
                    c = synth.synth_blocks(w, h, hs, vs, int(rng.integers(1 << 30)), qtabs=q, qtab_id=qid)[0]
                elif kind == 1:
                    c = synth.random_blocks(g.n_coded_blocks, int(rng.integers(1 << 30)), -300, 300)
                else:
                    c = synth.random_blocks(g.n_coded_blocks, int(rng.integers(1 << 30)))
                coefs.append(c)
            pad, off = int(rng.integers(0, 9)), int(rng.integers(0, 5))
            stride = 3 * w + pad
            img_stride = h * stride + int(rng.integers(0, 3)) * 4
            coef_t = torch.from_numpy(np.stack(coefs)).to(dev)
            q_t = torch.from_numpy(jb.resolve_qtabs(desc, q)).to(dev)
            raw = torch.full((off + nimg * img_stride + 16,), 0x5A, dtype=torch.uint8, device=dev)
            b = torch_batch(desc, nimg, coef_t, q_t, raw[off:off + nimg * img_stride].view(nimg, img_stride), rgb_row_stride=stride)
            b.rgb_image_stride = img_stride
            ctx.blocks_to_rgb_device(b, ts.cuda_stream)
            torch.cuda.synchronize()
            host = raw.cpu().numpy()
            assert (host[:off] == 0x5A).all() and (host[off + nimg * img_stride:] == 0x5A).all(), "wrote outside the batch"
            for i in range(nimg):
                img = host[off + i * img_stride: off + i * img_stride + h * stride].reshape(h, stride)
                want = ora.blocks_to_rgb(odesc(w, h, hs, vs, qid), coefs[i], q, nthreads=8)
                if not np.array_equal(img[:, :3 * w].reshape(h, w, 3), want):
                    bad = np.argwhere(img[:, :3 * w].reshape(h, w, 3) != want)
                    raise SystemExit(f"MISMATCH w={w} h={h} hs={hs} vs={vs} qid={qid} nimg={nimg} kind={kind} pad={pad} off={off} "
                                     f"image {i}: {len(bad)} bytes, first at {bad[0]}")
                assert (img[:, 3 * w:] == 0x5A).all(), "wrote into the row padding"
                tail = host[off + i * img_stride + h * stride: off + (i + 1) * img_stride]
                assert (tail == 0x5A).all(), "wrote between images"
            n += 1
            pixels += nimg * w * h
    print(f"stress ok: {n} launches, {pixels / 1e6:.1f} Mpixels compared, {time.time() - t0:.0f} s, seed {args.seed}")


if __name__ == "__main__":
    main()
