#!/usr/bin/env python3
"""End-to-end regimes of SURVEY 8(d) beside the kernel-only number of bench.py:
  (1) full decode(path): JPEG files -> T host threads -> pinned H2D -> kernels -> D2H
      (jb_batch_decoder), on synthetic baseline JPEGs written with PIL (quality 90, no Huffman
      optimisation -- the kind of file the reference's front end accepts).  The entropy stage runs
      on the device by default (the threads parse, de-stuff and pack); JPEGBLK_GPU_HUFFMAN=0 puts it
      on the host threads (north_star's split);
  (2) PCIe-inclusive block pipeline: pre-decoded coefficient blocks in pinned host memory ->
      jb_submit/jb_wait ring -> pixels in pinned host memory (no Huffman).
No rate is reported for unchecked pixels: before timing, every distinct file is decoded once through
the single-image decode(path) (jb_decode_file -- the path tests/ pin bit-exact against the oracle),
and EVERY image of the first timed pass of every configuration must equal that decode byte for byte
(`pixels_checked` in the output).
Every decode(path) line also carries the device-busy fraction (kernel time of the images decoded /
wall time; the kernel time is measured on a resident copy of one image), i.e. how idle the GPU is
while the host Huffman stage is the bottleneck.
Usage: python tools/e2e_bench.py [--size 1920x1080] [--sub 444|420] [--n 256] [--threads 1,8,16,64]
                                 [--source pil|writer]
  --source writer: files from the build's own baseline writer (tools/jpegwriter) on synthetic blocks
Several GPUs (host-fed scaling): launch one rank per GPU,
  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/e2e_bench.py --images <total> ...
every rank decodes its shard of the batch (image i -> rank i % N) on its own GPU with its own host
threads; rank 0 reports the whole-job rate = all images / slowest rank's wall time (gloo for the
bookkeeping; no collective on the data path).  E2E_SINGLE_DEVICE=1 rehearses that on one GPU.
"""
import argparse
import ctypes
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jpeg_decoder_amd as jb  # noqa: E402


def make_jpegs(n_distinct, w, h, sub, out_dir, dri_rows=0):
    from PIL import Image
    paths = []
    rng = np.random.default_rng(1)
    yy, xx = np.mgrid[0:h, 0:w]
    for i in range(n_distinct):
        base = np.stack([(xx * (2 + i) + yy) % 256, (yy * 3 + xx * (1 + i)) % 256, (xx + yy * 2) // 3 % 256], -1)
        noise = rng.normal(0, 12, (h // 8 + 1, w // 8 + 1, 3)).repeat(8, 0).repeat(8, 1)[:h, :w]
        img = np.clip(base * 0.6 + 60 + noise + rng.normal(0, 3, (h, w, 3)), 0, 255).astype(np.uint8)
        p = os.path.join(out_dir, f"synth_{w}x{h}_{sub}_{i}.jpg")
        kw = {"restart_marker_rows": dri_rows} if dri_rows else {}
        Image.fromarray(img).save(p, "JPEG", quality=90, subsampling={"444": 0, "420": 2}[sub], optimize=False, **kw)
        paths.append(p)
    return paths


def make_jpegs_writer(n_distinct, w, h, sub, out_dir, dri_rows=0):
    from jpeg_decoder_amd import synth
    hs, vs = {"444": (1, 1), "420": (2, 2), "422": (2, 1), "440": (1, 2)}[sub]
    ri = dri_rows * ((w + 8 * hs - 1) // (8 * hs))
    paths = []
    for i in range(n_distinct):
        coef, q = synth.synth_blocks(w, h, hs, vs, i)
        p = os.path.join(out_dir, f"writer_{w}x{h}_{sub}_{i}.jpg")
        with open(p, "wb") as f:
            f.write(synth.encode_jpeg(coef, w, h, hs, vs, q, restart_interval=ri))
        paths.append(p)
    return paths


def kernel_ms_per_image(desc, q, coef, device=0):
    """Device time of one image's launch with the blocks resident in HBM (torch = plumbing)."""
    import torch
    from jpeg_decoder_amd.api import torch_batch
    dev = torch.device(f"cuda:{device}")
    ts = torch.cuda.Stream(dev)
    with torch.cuda.stream(ts), jb.Context(device) as ctx:
        coef_t = torch.from_numpy(coef).to(dev).view(1, -1, 64)
        q_t = torch.from_numpy(jb.resolve_qtabs(desc, q)).to(dev)
        rgb_t = torch.empty((1, desc.height, 3 * desc.width), dtype=torch.uint8, device=dev)
        b = torch_batch(desc, 1, coef_t, q_t, rgb_t)
        for _ in range(50):
            ctx.blocks_to_rgb_device(b, ts.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(ts)
        for _ in range(50):
            ctx.blocks_to_rgb_device(b, ts.cuda_stream)
        e1.record(ts)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 50


def pinned_array(nbytes, dtype):
    p = jb.lib().jb_pinned_alloc(nbytes)
    if not p:
        raise MemoryError("jb_pinned_alloc")
    a = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(nbytes,)).view(dtype)
    return p, a


def device_output_run(jb, paths, want, threads, device, g0, args, w, h, world):
    """The batch with the decoded images left in device memory: no download at all."""
    import torch
    n = len(paths)
    per = (g0.rgb_bytes + 255) // 256 * 256
    region = torch.empty(n * per, dtype=torch.uint8, device=f"cuda:{device}")
    with jb.BatchDecoder(threads, device, g0.coef_bytes, g0.rgb_bytes) as dec:
        dec.set_device_output(region.data_ptr(), region.numel())
        dec.run_to_device(paths[:threads])
        runs = []
        for k in range(args.repeat):
            ptrs, dims, st, tm = dec.run_to_device(paths)
            assert all(x == 0 for x in st), st[:8]
            if k == 0:   # every image, copied back after the clock has stopped
                for i, p in enumerate(paths):
                    off = ptrs[i] - region.data_ptr()
                    got = region[off:off + g0.rgb_bytes].cpu().numpy().reshape(want[p].shape)
                    assert np.array_equal(got, want[p]), f"device output: image {i} differs from the single-image decode"
            runs.append(tm)
        on_device = dec.device_entropy_images
    tm = min(runs, key=lambda x: x["wall_s"])
    return {"output": "device memory (nothing downloaded)", "threads": threads, "images_per_s": round(args.n / tm["wall_s"], 1),
            "mpix_per_s": round(args.n * w * h / tm["wall_s"] / 1e6, 1), "entropy_cpu_s": round(tm["entropy_s"], 3),
            "submit_wait_s": round(tm["device_s"], 3), "wall_s": round(tm["wall_s"], 3), "walls": [round(x["wall_s"], 3) for x in runs],
            "n_gpus": world, "pixels_checked": n, "entropy_on_device": bool(on_device)}


def stream_run(jb, paths, want, threads, device, g0, args, arena):
    """The same files as batches of args.stream: through run() one batch after the other, and through
    submit / collect with two batches in flight (the start-up of one under the tail of the other)."""
    import time
    B = args.stream
    batches = [paths[i:i + B] for i in range(0, len(paths), B)]
    per = (g0.rgb_bytes + 255) // 256 * 256
    bad = []
    with jb.BatchDecoder(threads, device, g0.coef_bytes, g0.rgb_bytes, arena_bytes=B * per if arena else 0) as dec:
        def check_for(b):
            def check(i, view):
                if not np.array_equal(view, want[b[i]]):
                    bad.append(b[i])
            return check
        dec.run(paths[:threads], keep_pixels=False)
        t = dec.submit(paths[:threads])            # builds the second side
        dec.collect(t, keep_pixels=False)
        t = dec.submit(paths[:threads])
        t2 = dec.submit(paths[:threads])           # both sides once
        dec.collect(t, keep_pixels=False)
        dec.collect(t2, keep_pixels=False)
        seq, stm = [], []
        for k in range(args.repeat + 1):
            time.sleep(0.1)
            t0 = time.perf_counter()
            for b in batches:
                _, st, tm = dec.run(b, keep_pixels=False)
                assert tm["rc"] == 0, tm
            seq.append(time.perf_counter() - t0)
            time.sleep(0.1)   # (an idle gap: tools/timeline.py separates the passes by it)
            t0 = time.perf_counter()
            flight = []
            for b in batches:
                flight.append((dec.submit(b), b))
                if len(flight) == 2:
                    tk, bb = flight.pop(0)
                    _, st, tm = dec.collect(tk, keep_pixels=False, on_image=check_for(bb) if k == 0 else None)
                    assert tm["rc"] == 0, tm
            for tk, bb in flight:
                _, st, tm = dec.collect(tk, keep_pixels=False, on_image=check_for(bb) if k == 0 else None)
                assert tm["rc"] == 0, tm
            stm.append(time.perf_counter() - t0)
        assert not bad, f"stream: {len(bad)} images differ from the single-image decode"
    n = len(paths)
    # pass 0 is the checked one (the comparison runs inside the timed loop): not counted
    return {"batch": B, "batches": len(batches), "one_after_the_other_images_per_s": round(n / min(seq[1:]), 1),
            "two_in_flight_images_per_s": round(n / min(stm[1:]), 1), "walls_one_after_the_other": [round(x, 4) for x in seq],
            "walls_two_in_flight": [round(x, 4) for x in stm], "pixels_checked": n}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--sub", default="444")
    ap.add_argument("--n", "--images", dest="n", type=int, default=256)  # --images under torch.distributed.run (--n is ambiguous there)
    ap.add_argument("--threads", default="1,8,16,32,64")
    ap.add_argument("--source", default="pil", choices=["pil", "writer"])
    ap.add_argument("--distinct", type=int, default=8)
    ap.add_argument("--modes", default="malloc,arena", help="malloc | arena (pinned host arena) | device (pixels stay in HBM), comma separated")
    ap.add_argument("--dri", type=int, default=0, help="restart interval of the generated files in MCU rows (0 = none); files with "
                    "the batch decoder decodes the entropy stage on the device by default, JPEGBLK_GPU_HUFFMAN=0 on the host threads")
    ap.add_argument("--repeat", type=int, default=2, help="timed runs per configuration (the best is reported, all walls listed)")
    ap.add_argument("--no-pcie", action="store_true", help="skip part (2), so that the last device activity of the run is the last "
                    "timed batch (tools/timeline.py reads that burst out of a rocprofv3 trace)")
    ap.add_argument("--stream", type=int, default=0, help="also stream the files as batches of this many through "
                    "jb_batch_decoder_submit / _collect (two in flight), next to the same batches through run() one after the other")
    args = ap.parse_args()
    w, h = (int(v) for v in args.size.split("x"))
    from jpeg_decoder_amd.shard import rank_from_env
    world, rank, device = rank_from_env(os.environ, os.environ.get("E2E_SINGLE_DEVICE") == "1")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    out = {"size": args.size, "sampling": args.sub, "n_images": args.n, "host_cpus": os.cpu_count(),
           "cpu_affinity": len(os.sched_getaffinity(0)), "source": args.source}
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        distinct = (make_jpegs if args.source == "pil" else make_jpegs_writer)(args.distinct, w, h, args.sub, d, args.dri)
        out["restart_interval_rows"] = args.dri
        out["JPEGBLK_GPU_HUFFMAN"] = os.environ.get("JPEGBLK_GPU_HUFFMAN", "(unset: the batch decoder's default = entropy stage on the device)")
        from jpeg_decoder_amd.shard import shard_images
        mine = shard_images(args.n, rank, world)           # image i -> rank i % world
        paths = [distinct[i % len(distinct)] for i in mine]
        n_mine = len(paths)
        out["file_kbytes_mean"] = round(float(np.mean([os.path.getsize(p) for p in distinct])) / 1024, 1)
        # warm-up (file cache, HIP init)
        jb.decode_batch(paths[:8], n_threads=4, device=device, keep_pixels=False)
        res = []
        # what every image of a batch must decode to: the single-image decode(path) of its file
        with jb.Context(device) as one:
            want = {p: one.decode_file(p) for p in distinct}
        d0, q0, c0 = jb.entropy_decode(open(distinct[0], "rb").read())
        g0 = jb.geometry_of(d0)
        k_ms = kernel_ms_per_image(d0, q0, c0, device)
        out["kernel_ms_per_image"] = round(k_ms, 4)
        del c0
        # two output modes: "malloc" = the default ABI (pixels copied from pinned staging into
        # malloc'ed per-image buffers), "arena" = a pinned output arena owned by the decoder
        for mode, t in [(m, int(x)) for m in args.modes.split(",") for x in args.threads.split(",")]:
            arena = (n_mine * ((g0.rgb_bytes + 255) // 256 * 256)) if mode == "arena" else 0
            if mode == "device":   # device-resident output: the pixels stay in HBM (jb_batch_decoder_set_device_output)
                res.append(device_output_run(jb, paths, want, t, device, g0, args, w, h, world))
                continue
            with jb.BatchDecoder(t, device, g0.coef_bytes, g0.rgb_bytes, arena_bytes=arena) as dec:
                dec.run(paths[:t], keep_pixels=False)          # touch every lane once
                bad = []

                def check(i, view):
                    if not np.array_equal(view, want[paths[i]]):
                        bad.append(i)

                # timed: contexts and pinned buffers exist; the first pass hands every image to `check`
                # (after the decoder's clock has stopped)
                runs = [dec.run(paths, keep_pixels=False, on_image=check if k == 0 else None) for k in range(args.repeat)]
                on_device = dec.device_entropy_images
                assert not bad, f"{mode}/{t} threads: {len(bad)} of {n_mine} images differ from the single-image decode: {bad[:8]}"
                _, st, tm = min(runs, key=lambda x: x[2]["wall_s"])
                walls = [round(x[2]["wall_s"], 3) for x in runs]
            assert all(s == 0 for s in st), st[:8]
            stream = None
            if args.stream > 0:
                stream = stream_run(jb, paths, want, t, device, g0, args, mode == "arena")
            if dist is not None:   # whole job: all images / the slowest rank
                import torch
                tw = torch.tensor([tm["wall_s"]], dtype=torch.float64)
                dist.all_reduce(tw, op=dist.ReduceOp.MAX)
                tm = dict(tm, wall_s=float(tw.item()))
            res.append({"output": mode, "threads": t, "images_per_s": round(args.n / tm["wall_s"], 1),
                        "mpix_per_s": round(args.n * w * h / tm["wall_s"] / 1e6, 1),
                        "entropy_cpu_s": round(tm["entropy_s"], 3), "submit_wait_s": round(tm["device_s"], 3),
                        "wall_s": round(tm["wall_s"], 3), "walls": walls,
                        "device_busy_fraction": round(n_mine * k_ms * 1e-3 / tm["wall_s"], 4), "n_gpus": world,
                        "pixels_checked": n_mine, "entropy_on_device": bool(on_device)})
            if stream:
                res[-1]["stream"] = stream
        out["decode_path"] = res
        if world > 1:
            if rank == 0:
                print(json.dumps(out))
            dist.destroy_process_group()
            return
        if args.no_pcie:
            print(json.dumps(out))
            return
        # (2) PCIe-inclusive block pipeline from pre-decoded coefficients
        desc, q, coef = jb.entropy_decode(open(distinct[0], "rb").read())
        g = jb.geometry_of(desc)
        slots = 3
        bufs = []
        for _ in range(slots):
            pc, ac = pinned_array(g.coef_bytes, np.int16)
            ac[:] = coef.reshape(-1)
            pr, ar = pinned_array(g.rgb_bytes, np.uint8)
            bufs.append((pc, ac, pr, ar))
        with jb.Context(device, g.coef_bytes, g.rgb_bytes, slots) as ctx:
            n = max(args.n, 64)
            tickets = []
            for warm in (True, False):
                t0 = time.perf_counter()
                for i in range(n):
                    _, ac, _, ar = bufs[i % slots]
                    tickets.append(ctx.submit(desc, ac, q, ar))
                    if len(tickets) >= slots:
                        ctx.wait(tickets.pop(0))
                while tickets:
                    ctx.wait(tickets.pop(0))
                dt = time.perf_counter() - t0
            out["pcie_pipeline"] = {"images_per_s": round(n / dt, 1), "mpix_per_s": round(n * w * h / dt / 1e6, 1),
                                    "GBps_h2d_plus_d2h": round(n * (g.coef_bytes + g.rgb_bytes) / dt / 1e9, 2)}
        for pc, _, pr, _ in bufs:
            jb.lib().jb_pinned_free(pc)
            jb.lib().jb_pinned_free(pr)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
