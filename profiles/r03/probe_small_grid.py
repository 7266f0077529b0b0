#!/usr/bin/env python3
"""the REAL kernel on a small grid.  jb_small_kernel_444 / _420 (one wave per 16 / 8 MCUs;
JPEGBLK_SMALL_GRID=1) against jb_tile_kernel<1,1> / <2,2> (192 lanes per 64 / 32 MCUs; =0) on launches that do not fill the device: HIP events around
every launch, cold buffer sets (> 512 MiB rotating), the two contexts interleaved round by round on one box."""
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
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream(dev)
    ctxs = {}
    for knob in ("0", "1"):
        os.environ["JPEGBLK_SMALL_GRID"] = knob
        ctxs[knob] = jb.Context(0)
    os.environ.pop("JPEGBLK_SMALL_GRID")

    def timed(ctx, res, launches):
        nb = len(res.batches)
        for k in range(40):
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
        import os as _os
        big = [("4096x4096-422", 8, 1), ("4096x4096-422", 16, 1), ("8192x8192-422", 4, 1), ("4096x4096-440", 8, 1), ("1920x1080-422", 128, 1), ("679x451-422", 512, 2)]
        for wl, n, sets in (big if _os.environ.get("PROBE_BIG") else [("1920x1080-444", 1, 32), ("1920x1080-444", 2, 16), ("1920x1080-444", 4, 8), ("1280x720-444", 1, 64),
                            ("640x360-444", 1, 64), ("4096x4096-444", 1, 4),
                            ("640x360-420", 1, 64), ("1920x1080-420", 1, 64), ("1920x1080-420", 4, 16), ("1920x1080-420", 8, 8),
                            ("4096x4096-420", 1, 8), ("4096x4096-420", 2, 4), ("4096x4096-420", 4, 2), ("4096x4096-420", 8, 1),
                            ("1920x1080-420", 32, 2), ("679x451-420", 512, 2), ("679x451-444", 512, 2), ("427x640-420", 512, 2),
                            ("640x360-422", 1, 64), ("1920x1080-422", 1, 32), ("1920x1080-422", 4, 8), ("4096x4096-422", 1, 4), ("4096x4096-422", 4, 1),
                            ("640x360-440", 1, 64), ("1920x1080-440", 1, 32), ("1920x1080-440", 4, 8), ("4096x4096-440", 1, 4), ("4096x4096-440", 4, 1)]):
            res = bench.Resident(jb, torch, dev, wl, n, sets, seed=1)
            us = {"0": [], "1": []}
            for rnd in range(4):
                for knob in ("0", "1"):
                    us[knob] += timed(ctxs[knob], res, 200)
            g = res.g
            per_default, per_small = (32, 8) if wl.endswith("420") else (64, 16)
            tiles_default = n * ((g.mcus_x * g.mcus_y + per_default - 1) // per_default)
            tiles_small = n * g.mcus_y * ((g.mcus_x + per_small - 1) // per_small)
            row = {"workgroups_default": tiles_default, "workgroups_small": tiles_small, "algorithmic_bytes": res.alg_bytes}
            for knob, name in (("0", "jb_tile_kernel"), ("1", "jb_small_kernel")):
                med = float(np.median(us[knob]))
                row[name] = {"us_median": round(med, 2), "us_min": round(float(np.min(us[knob])), 2), "us_mean": round(float(np.mean(us[knob])), 2),
                             "GB_s": round(res.alg_bytes / med / 1e3, 1)}
            out[f"{wl} x{n}"] = row
            print(f"{wl} x{n}", json.dumps(row), flush=True)
            del res
            torch.cuda.empty_cache()
    for c in ctxs.values():
        c.close()
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "probe_small_grid.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
