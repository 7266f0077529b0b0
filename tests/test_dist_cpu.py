"""CPU test of the N>1 path (gloo, world_size 2): images shard by index with no data-path
collective; the benchmark bookkeeping (barrier, MAX of elapsed, SUM of pixels) is what uses
torch.distributed.  The device seam is stood in for by the oracle here (this container has no GPU);
on the GPU box the same sharding feeds libjpegblk.so."""
import hashlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import BASELINE_IMAGES, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import load_golden
    from jpeg_decoder_amd.shard import job_throughput, shard_images
    from oracle.pyoracle import Oracle
    ora = Oracle()
    mine = shard_images(len(BASELINE_IMAGES), rank, world)
    dist.barrier()
    pixels, digests = 0, {}
    for i in mine:
        desc, coef, q, _ = load_golden(BASELINE_IMAGES[i])
        rgb = ora.blocks_to_rgb(desc, coef, q)
        digests[BASELINE_IMAGES[i]] = hashlib.sha256(rgb.tobytes()).hexdigest()
        pixels += desc.width * desc.height
    dist.barrier()
    elapsed = 1.0 + rank  # synthetic per-rank time: the job time must be the MAX
    total_pixels, job_s = job_throughput(dist, torch.device("cpu"), pixels, elapsed)
    gathered = [None] * world
    dist.all_gather_object(gathered, (mine, digests))
    if rank == 0:
        np.save(os.path.join(out_dir, "result.npy"),
                np.array([total_pixels, job_s], dtype=np.float64))
        import json
        with open(os.path.join(out_dir, "gathered.json"), "w") as f:
            json.dump(gathered, f)
    dist.destroy_process_group()


def test_two_rank_image_sharding(tmp_path, manifest):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    import json
    total_pixels, job_s = np.load(tmp_path / "result.npy")
    gathered = json.load(open(tmp_path / "gathered.json"))
    owned = sorted(i for mine, _ in gathered for i in mine)
    assert owned == list(range(len(BASELINE_IMAGES)))                 # every image exactly once
    assert gathered[0][0] == [0, 2, 4] and gathered[1][0] == [1, 3, 5]  # by index, round-robin
    for _, digests in gathered:
        for name, h in digests.items():
            assert h == manifest["images"][name]["rgb_sha256"]
    want_pixels = sum(manifest["images"][n]["width"] * manifest["images"][n]["height"] for n in BASELINE_IMAGES)
    assert total_pixels == want_pixels and job_s == 2.0               # SUM of pixels, MAX of time


def test_shard_edge_cases():
    from jpeg_decoder_amd.shard import shard_images
    assert shard_images(0, 0, 4) == []
    assert shard_images(3, 3, 4) == []
    assert sorted(sum((shard_images(1024, r, 8) for r in range(8)), [])) == list(range(1024))
    assert all(len(shard_images(1024, r, 8)) == 128 for r in range(8))
    with pytest.raises(ValueError):
        shard_images(4, 4, 4)


def test_bench_refuses_a_world_size_that_differs_from_gpus():
    """bench.py never prints n_gpus != --gpus: under a launcher whose WORLD_SIZE differs from --gpus
    it exits 2 before touching the GPU (so this runs on the CPU)."""
    import subprocess
    for world, gpus in (("1", "8"), ("2", "1"), ("4", "2")):
        env = dict(os.environ, WORLD_SIZE=world, RANK="0", LOCAL_RANK="0")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", gpus, "--steps", "1"],
                           capture_output=True, text=True, env=env, timeout=120)
        assert r.returncode == 2, (world, gpus, r.stdout, r.stderr)
        assert "refusing" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_launcher_environment_to_device_index():
    """One process per GPU: rank r of the node drives device LOCAL_RANK (bench.py, tools/e2e_bench.py); the
    rehearsal knobs put every rank on device 0; an inconsistent environment is an error, not device 0."""
    from jpeg_decoder_amd.shard import rank_from_env, shard_images
    assert rank_from_env({}) == (1, 0, 0)
    for world in (2, 4, 8):
        seen = set()
        for r in range(world):
            w, rank, dev = rank_from_env({"WORLD_SIZE": str(world), "RANK": str(r), "LOCAL_RANK": str(r)})
            assert (w, rank, dev) == (world, r, r)
            seen.update(shard_images(1024, rank, w))
            assert rank_from_env({"WORLD_SIZE": str(world), "RANK": str(r), "LOCAL_RANK": str(r)}, single_device=True) == (world, r, 0)
        assert seen == set(range(1024))          # the strong-scaled batch: every file exactly once
    for bad in ({"WORLD_SIZE": "2", "RANK": "2", "LOCAL_RANK": "0"}, {"WORLD_SIZE": "0"}, {"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "-1"}):
        with pytest.raises(ValueError):
            rank_from_env(bad)


def test_bench_and_e2e_bench_take_the_device_from_the_same_helper():
    """Both drivers map LOCAL_RANK -> device through jpeg_decoder_amd.shard.rank_from_env (no copy of the rule that
    could drift), and hand that index to jb.Context / jb.BatchDecoder."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for rel in ("bench.py", os.path.join("tools", "e2e_bench.py")):
        text = open(os.path.join(root, rel)).read()
        assert "rank_from_env(os.environ" in text and 'os.environ.get("LOCAL_RANK"' not in text, rel
