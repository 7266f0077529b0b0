"""Multi-GPU sharding of a batch of independent images.

The block pipeline has no exchange step: images (and blocks) are independent (reference
jpeg.cpp:574-589 touches each block on its own), so a batch shards by image index across one
process per GPU and the data path needs NO collective.  The only communication is the benchmark's
own bookkeeping: a barrier around the timed region and a reduction of elapsed time / pixel counts.
"""


def shard_images(n_images, rank, world):
    """Image indices owned by `rank`: i with i % world == rank (round-robin keeps every GPU's
    pinned staging ring equally loaded when image sizes vary slowly along the batch)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return list(range(rank, n_images, world))


def job_throughput(dist, device, pixels_local, elapsed_local):
    """Whole-job (pixels, seconds): SUM of pixels over ranks, MAX of elapsed over ranks.
    `dist` is torch.distributed (initialised) or None for a single process."""
    if dist is None:
        return float(pixels_local), float(elapsed_local)
    import torch
    t = torch.tensor([float(elapsed_local)], dtype=torch.float64, device=device)
    p = torch.tensor([float(pixels_local)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(p, op=dist.ReduceOp.SUM)
    return float(p.item()), float(t.item())


def rank_from_env(env, single_device=False):
    """(world, rank, device index) of this process from the launcher's environment
    (torch.distributed.run sets WORLD_SIZE / RANK / LOCAL_RANK): one process per GPU, rank r of a
    node drives device LOCAL_RANK.  single_device: every rank on device 0 (rehearsing N ranks on a
    one-GPU box).  Used by bench.py and tools/e2e_bench.py; unit-tested on the CPU."""
    world = int(env.get("WORLD_SIZE", "1"))
    rank = int(env.get("RANK", "0"))
    local = int(env.get("LOCAL_RANK", "0"))
    if world < 1 or not (0 <= rank < world) or local < 0:
        raise ValueError(f"inconsistent launcher environment: WORLD_SIZE={world} RANK={rank} LOCAL_RANK={local}")
    return world, rank, (0 if single_device else local)
