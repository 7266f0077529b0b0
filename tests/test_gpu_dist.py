"""Two real ranks of the HIP path (-m gpu).  The driver's multi-GPU run needs an 8-GPU node; what a
one-GPU box can prove is everything but the device index: two fresh child processes (started by
torch.distributed.run, as the driver starts them) each drive libjpegblk.so on the GPU, shard the
golden images by index, and must reproduce every golden digest; the whole-job figures are the SUM
of pixels and the MAX of elapsed.  And bench.py must honour --gpus: run plainly with --gpus 2 it
starts its two ranks itself and prints n_gpus 2; with a world size that differs from --gpus it
refuses (covered on the CPU in tests/test_dist_cpu.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import BASELINE_IMAGES, ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_child_ranks_real_hip_path(tmp_path, manifest):
    out = tmp_path / "ranks.json"
    env = dict(os.environ, JB_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(ROOT, "tests", "dist_gpu_worker.py"), str(out)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    res = json.load(open(out))
    ranks = sorted(res["ranks"], key=lambda x: x["rank"])
    assert [x["mine"] for x in ranks] == [[0, 2, 4], [1, 3, 5]]            # by index, round-robin
    assert ranks[0]["pid"] != ranks[1]["pid"] != os.getpid()                # fresh processes
    seen = {}
    for x in ranks:
        assert x["lib"].endswith("libjpegblk.so")
        for name, h in x["digests"].items():
            assert h == manifest["images"][name]["rgb_sha256"], name
            assert x["digests_batch"][name] == h, name
            seen[name] = seen.get(name, 0) + 1
    assert seen == {n: 1 for n in BASELINE_IMAGES}                          # every image exactly once
    want_pixels = sum(manifest["images"][n]["width"] * manifest["images"][n]["height"] for n in BASELINE_IMAGES)
    assert res["total_pixels"] == want_pixels and res["job_s"] == 2.0       # SUM of pixels, MAX of time


def test_bench_gpus_2_run_plainly_reports_two_ranks():
    """`python bench.py --gpus 2` with no launcher: it must start two ranks itself (before any GPU
    call) and print n_gpus 2 -- here both on the one GPU of the box, bookkeeping over gloo."""
    env = dict(os.environ, JB_BENCH_SINGLE_DEVICE="1", JB_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
                        "--precondition", "10", "--images-per-step", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak"
    assert d["config"]["launched_by"] == "bench.py" and "x2" in d["config"]["parallelism"]
    assert "cpu_baseline" not in d and "configs" not in d                   # N = 1 only
    assert d["value"] > 0 and 0 < d["roofline"]["frac"] < 1
    # configs 4 and 5 end to end over files: every rank decodes its share, the line counts all of them
    runs = d["end_to_end"]["runs"]
    assert runs["config4_files_1080p_444"]["weak"]["entropy_on_device"]["images"] == 2 * 128
    assert runs["config5_files_8192_420"]["weak"]["entropy_on_device"]["images"] == 2 * 32
    # strong-scaled: the batch as BASELINE.json states it, dealt over the two ranks
    assert runs["config4_files_1080p_444"]["strong"]["entropy_on_device"]["images"] == 1024
    assert runs["config5_files_8192_420"]["strong"]["entropy_on_device"]["images"] == 256
    for r in runs.values():
        for form in ("weak", "strong"):
            assert r[form]["entropy_on_device"]["entropy_stage_ran_on_device"] and r[form]["entropy_on_device"]["images_per_s"] > 0


def test_bench_one_rank_over_rccl():
    """The N > 1 code path with ONE rank: process group over RCCL (backend "nccl"), barriers and the reductions of
    job_throughput on the GPU -- what a one-GPU box can run of what the driver's 8-GPU node runs."""
    env = dict(os.environ, JB_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1",
                        "--precondition", "10", "--images-per-step", "2", "--no-cpu-baseline", "--no-configs", "--no-e2e"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["value"] > 0 and 0 < d["roofline"]["frac"] < 1
