"""One rank of the two-rank GPU test (tests/test_gpu_dist.py), started by torch.distributed.run.
Every rank is a fresh process driving the REAL HIP path through libjpegblk.so: it takes its shard
of the golden images (image i -> rank i % world, jpeg_decoder_amd/shard.py), decodes them with
decode(path) on its device -- once through a jb_ctx, once through a jb_batch_decoder whose host
threads allocate their pinned staging against that device -- and reports SHA-256 digests; the
bookkeeping (SUM of pixels, MAX of elapsed) goes through torch.distributed (gloo), the data path
uses no collective.  JB_SINGLE_DEVICE=1: all ranks share device 0 (a one-GPU box)."""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out_path = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    device = 0 if os.environ.get("JB_SINGLE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    import jpeg_decoder_amd as jb
    from conftest import BASELINE_IMAGES, GOLD
    from jpeg_decoder_amd.shard import job_throughput, shard_images
    dist.init_process_group("gloo")
    assert jb.lib().jb_device_count() > device, jb.lib().jb_last_error(None)
    mine = shard_images(len(BASELINE_IMAGES), rank, world)
    paths = [os.path.join(GOLD, "images", BASELINE_IMAGES[i] + ".jpg") for i in mine]
    dist.barrier()
    t0 = time.perf_counter()
    digests, digests_batch, pixels = {}, {}, 0
    with jb.Context(device) as ctx:
        assert ctx.device == device
        for i, p in zip(mine, paths):
            rgb = ctx.decode_file(p)
            digests[BASELINE_IMAGES[i]] = hashlib.sha256(rgb.tobytes()).hexdigest()
            pixels += rgb.shape[0] * rgb.shape[1]
    with jb.BatchDecoder(2, device) as dec:
        imgs, st, tm = dec.run(paths)
        assert tm["rc"] == 0 and all(s == 0 for s in st), (tm, st)
        for i, rgb in zip(mine, imgs):
            digests_batch[BASELINE_IMAGES[i]] = hashlib.sha256(rgb.tobytes()).hexdigest()
    elapsed_real = time.perf_counter() - t0
    dist.barrier()
    elapsed = 1.0 + rank  # synthetic per-rank time: the job time must be the MAX over ranks
    total_pixels, job_s = job_throughput(dist, torch.device("cpu"), pixels, elapsed)
    gathered = [None] * world
    dist.all_gather_object(gathered, {"rank": rank, "pid": os.getpid(), "device": device, "mine": mine, "digests": digests,
                                      "digests_batch": digests_batch, "pixels": pixels, "elapsed_real_s": elapsed_real,
                                      "lib": jb.lib_path()})
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump({"total_pixels": total_pixels, "job_s": job_s, "ranks": gathered}, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
