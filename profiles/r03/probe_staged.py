#!/usr/bin/env python3
"""the line-aligned ("staged") store stage of the linear tiling (JPEGBLK_STAGED_STORE=1) against the
12-byte-per-lane stores (=0) on tightly packed odd rows -- the reference's bundled sizes -- and on aligned rows as a
control.  HIP events around every launch, the two contexts interleaved round by round on one box; and every byte the
staged stage writes (pixels, row padding) compared with the product path's.  The staged stage only exists in -DJB_LAB
builds of the kernels:  bash tools/build_variant.sh lab && JPEGBLK_LIB=tools/ab/libjpegblk_lab.so python profiles/r03/probe_staged.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    import jpeg_decoder_amd as jb
    assert "lab" in os.environ.get("JPEGBLK_LIB", ""), "run with JPEGBLK_LIB=tools/ab/libjpegblk_lab.so (bash tools/build_variant.sh lab)"
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream(dev)
    ctxs = {}
    os.environ["JPEGBLK_SMALL_GRID"] = "0"
    for knob in ("0", "1"):
        os.environ["JPEGBLK_STAGED_STORE"] = knob
        ctxs[knob] = jb.Context(0)
    os.environ.pop("JPEGBLK_STAGED_STORE")

    def timed(ctx, res, launches):
        nb = len(res.batches)
        for k in range(30):
            ctx.blocks_to_rgb_device(res.batches[k % nb], stream.cuda_stream)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
        for k in range(launches):
            evs[k][0].record(stream)
            ctx.blocks_to_rgb_device(res.batches[k % nb], stream.cuda_stream)
            evs[k][1].record(stream)
        torch.cuda.synchronize()
        return [a.elapsed_time(b) * 1e3 for a, b in evs]

    out = {}
    with torch.cuda.stream(stream):
        for wl, n, sets in [("679x451-420", 512, 2), ("679x451-444", 512, 2), ("679x451-420@2048", 512, 2), ("1279x853-420", 256, 2),
                            ("427x640-420", 512, 2), ("1921x1081-444", 64, 2), ("1920x1080-444", 64, 2), ("640x448-420", 512, 2)]:
            res = bench.Resident(jb, torch, dev, wl, n, sets, seed=1)
            outs = {}
            for knob in ("0", "1"):
                res.tensors[0][1].fill_(0xC5)
                ctxs[knob].blocks_to_rgb_device(res.batches[0], stream.cuda_stream)
                torch.cuda.synchronize()
                outs[knob] = res.tensors[0][1].clone()
            assert torch.equal(outs["0"], outs["1"]), f"{wl}: the staged stage and the 12-byte stores wrote different bytes"
            del outs
            us = {"0": [], "1": []}
            for rnd in range(3):
                for knob in ("0", "1"):
                    us[knob] += timed(ctxs[knob], res, 60)
            row = {"algorithmic_bytes": res.alg_bytes, "bytes_equal": True}
            for knob, name in (("0", "12-byte stores"), ("1", "staged")):
                med = float(np.median(us[knob]))
                row[name] = {"us_median": round(med, 2), "us_min": round(float(np.min(us[knob])), 2), "GB_s": round(res.alg_bytes / med / 1e3, 1),
                             "frac_of_8TB_s": round(res.alg_bytes / med / 1e3 / 8000, 4)}
            out[f"{wl} x{n}"] = row
            print(f"{wl} x{n}", json.dumps(row), flush=True)
            del res
            torch.cuda.empty_cache()
    for c in ctxs.values():
        c.close()
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "probe_staged.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
