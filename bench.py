#!/usr/bin/env python3
"""bench.py -- headline benchmark of the block pipeline (dequantize -> IDCT -> YCbCr->RGB).

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by the driver as  python -m torch.distributed.run --nproc-per-node N ...
  (one rank per GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env).

Workload (BASELINE.json north_star / SURVEY 8d "roofline target"): a stream of synthetic
4096x4096 baseline 4:4:4 images, already Huffman-decoded into packed int16 coefficient blocks
resident in HBM.  One STEP = one pass of the hot path over one batch of IMAGES_PER_STEP such
images (one kernel launch; 4.8 GB of distinct input+output per step, so nothing is served from
the 256 MiB Infinity Cache).  Every rank owns its own batch (images shard by index, no
collective on the data path): weak scaling.

Prints ONE JSON line on rank 0:  metric = Mpixels/s decoded (whole job), plus
  roofline     achieved = algorithmic bytes per launch / mean kernel time (HIP events on the
               launch stream around every 8th timed step), against the 8 TB/s HBM3E peak;
  cpu_baseline the reference CPU path (oracle/_ref, the genuine reference compiled in place,
               kind "reference") or, if that build is absent, the C restatement (kind "port"),
               timed on this box's host cores on a bounded sample -- rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH, HEIGHT, HS, VS = 4096, 4096, 1, 1
IMAGES_PER_STEP = 32  # 4.8 GB per launch: the drain of one launch is 1/4 of what it is with 8 images (+2 % kernel rate)
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
SAMPLING_NAME = {(1, 1): "4:4:4", (2, 2): "4:2:0", (2, 1): "4:2:2", (1, 2): "4:4:0"}


def cpu_baseline(coef, qtabs, budget_s=12.0):
    """Reference CPU path on one host core (the reference is single-threaded), bounded sample."""
    import numpy as np
    from oracle.pyoracle import Oracle, Ref, make_desc
    desc = make_desc(WIDTH, HEIGHT, HS, VS)
    mpix = WIDTH * HEIGHT / 1e6
    out = {}
    if Ref.available():
        ref = Ref()
        ms = [0.0, 0.0, 0.0]
        t_hot, reps, t0 = 0.0, 0, time.time()
        while time.time() - t0 < budget_s:
            ref.blocks_to_rgb(desc, coef, qtabs, stage_ms=ms)
            t_hot += sum(ms) / 1e3
            reps += 1
        out = {"value": round(mpix * reps / t_hot, 2), "unit": "Mpixels/s", "cores": 1, "kind": "reference",
               "sample": f"{reps} x one {WIDTH}x{HEIGHT} 4:4:4 image through the reference's own "
                         f"dequantize+inverseDCT+YCbCrToRGB (oracle/_ref, g++ -O2), {t_hot:.1f} s"}
    ora = Oracle()
    reps = 3
    s1 = ora.time_blocks_to_rgb(desc, coef, qtabs, 1, reps)
    port1 = mpix * reps / s1
    ncores = os.cpu_count() or 1
    nthr = min(ncores, 64)
    reps_mt = 8
    smt = ora.time_blocks_to_rgb(desc, coef, qtabs, nthr, reps_mt)
    if not out:
        out = {"value": round(port1, 2), "unit": "Mpixels/s", "cores": 1, "kind": "port",
               "sample": f"{reps} x one {WIDTH}x{HEIGHT} 4:4:4 image through oracle/jpegblk_oracle.c (gcc -O2), {s1:.1f} s"}
    try:
        model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        model = "unknown"
    out["cpu_model"] = model
    out["flags"] = "reference harness: g++ -O2 -fno-access-control -U_FORTIFY_SOURCE (oracle/Makefile); port: gcc -O2 -ffp-contract=off; no -march=native, no -ffast-math"
    out["port_1core_mpix_s"] = round(port1, 2)
    out["port_allcores_mpix_s"] = round(mpix * reps_mt / smt, 2)
    out["port_allcores_threads"] = nthr
    out["host_cpus"] = ncores
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--precondition", type=int, default=400,
                    help="untimed launches before the warm-up steps (lets clocks/power settle)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--images-per-step", type=int, default=IMAGES_PER_STEP)
    ap.add_argument("--workload", default="4096x4096-444",
                    help="WxH-444|420|422|440 (default = the headline workload; others are the "
                         "remaining BASELINE.json configs, for DESIGN.md tables)")
    args = ap.parse_args()
    global WIDTH, HEIGHT, HS, VS
    dims, samp = args.workload.split("-")
    WIDTH, HEIGHT = (int(v) for v in dims.split("x"))
    HS, VS = {"444": (1, 1), "420": (2, 2), "422": (2, 1), "440": (1, 2)}[samp]

    import numpy as np
    import torch
    import jpeg_decoder_amd as jb
    from jpeg_decoder_amd import synth
    from jpeg_decoder_amd.api import torch_batch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # rehearsal knobs (never set by the driver): run N ranks on a one-GPU box
    backend = os.environ.get("JB_BENCH_BACKEND", "nccl")  # "nccl" = RCCL on ROCm
    if os.environ.get("JB_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    if world > 1 or os.environ.get("JB_BENCH_FORCE_DIST") == "1":  # the latter: rehearse the RCCL path with one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend)
    n_gpus = max(world, 1)
    if jb.lib().jb_device_count() < 1:
        raise RuntimeError("bench.py needs a HIP device: " + jb.lib().jb_last_error(None).decode())
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")

    # all device work of this benchmark runs on one explicit torch stream, and the kernel is
    # launched on that same stream through the C ABI, so torch.cuda.Event times the kernel
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    nimg = args.images_per_step
    desc = jb.make_desc(WIDTH, HEIGHT, HS, VS)
    g = jb.geometry_of(desc)
    # one seeded image per rank on the host; the other images of the batch are MCU-rotations of
    # it made on the device (distinct bytes, same statistics)
    coef, qtabs = synth.synth_blocks(WIDTH, HEIGHT, HS, VS, image_index=rank)
    base = torch.from_numpy(coef).to(dev)
    coef_t = torch.empty((nimg, g.n_coded_blocks, 64), dtype=torch.int16, device=dev)
    for i in range(nimg):
        coef_t[i] = torch.roll(base, shifts=i * 7919 * g.blocks_per_mcu, dims=0)
    del base
    q_t = torch.from_numpy(jb.resolve_qtabs(desc, qtabs)).to(dev)
    rgb_t = torch.zeros((nimg, HEIGHT, 3 * WIDTH), dtype=torch.uint8, device=dev)
    batch = torch_batch(desc, nimg, coef_t, q_t, rgb_t)
    ctx = jb.Context(local_rank)

    def step():
        ctx.blocks_to_rgb_device(batch, stream.cuda_stream)

    def barrier():
        if dist is not None:
            dist.barrier()

    # Pre-conditioning (untimed, before the W warm-up steps): MI355X power management needs a
    # few hundred back-to-back launches of this kernel to settle (launch times go 240 -> 350 ->
    # 240 us over the first ~100 launches, tools/steps.py); a stream workload runs in the settled
    # state, so that is the state the K timed steps are taken in.
    for _ in range(args.precondition):
        step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    # HIP events bracket the kernel of every 8th timed step (25 samples of the default 200 steps).
    # Event packets between back-to-back launches are measurement overhead that the whole-job clock
    # sees: with a pair around EVERY step ms_per_step was 3.7 % higher (0.2110 vs 0.2035 ms) while the
    # bracketed kernel time was the same (JB_BENCH_EVENT_EVERY=1 restores that)
    every = max(1, int(os.environ.get("JB_BENCH_EVENT_EVERY", "8")))
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(0, args.steps, every)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        if k % every == 0:
            evs[k // every][0].record(stream)
        step()
        if k % every == 0:
            evs[k // every][1].record(stream)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_ms = [a.elapsed_time(b) for a, b in evs]
    pixels_per_step = nimg * WIDTH * HEIGHT
    from jpeg_decoder_amd.shard import job_throughput
    total_pixels, elapsed = job_throughput(dist, dev if backend == "nccl" else torch.device("cpu"),
                                           pixels_per_step * args.steps, elapsed)

    alg_bytes = nimg * (g.n_coded_blocks * 128 + WIDTH * HEIGHT * 3)  # 128 B/block in + 3 B/pixel out
    mean_ms = float(np.mean(kern_ms))
    achieved = alg_bytes / (mean_ms * 1e-3) / 1e9
    value = total_pixels / elapsed / 1e6

    if rank == 0:
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                # the PMC passes profile the default command: only report them for that workload
                if rec.get("algorithmic_bytes_per_launch") in (None, alg_bytes):
                    traffic = rec.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "Mpixels/s decoded (IDCT+colour)", "value": round(value, 1), "unit": "Mpixels/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"stream of {WIDTH}x{HEIGHT} baseline {SAMPLING_NAME[(HS, VS)]} images, {nimg} images per step "
                                   f"(one launch) per GPU, coefficient blocks resident in HBM",
                       "images_per_step_per_gpu": nimg, "sampling": SAMPLING_NAME[(HS, VS)], "parallelism": f"images sharded x{n_gpus}, no collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "kernel": jb.lib().jb_kernel_name(desc).decode(), "algorithmic_bytes_per_launch": alg_bytes,
                         "kernel_samples": len(kern_ms), "event_every": every,
                         "kernel_ms_mean": round(mean_ms, 4), "kernel_ms_median": round(float(np.median(kern_ms)), 4), "kernel_ms_min": round(float(np.min(kern_ms)), 4),
                         "kernel_gpix_s": round(pixels_per_step / (mean_ms * 1e-3) / 1e9, 2)},
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(coef, qtabs)
        print(json.dumps(line), flush=True)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
